"""`bench.py --gpus 2 --config 4` end to end on CPU (world_size 2, gloo, numpy stand-in engine): everything of the multi-GPU bench
path except the kernels - argument parsing, the self-spawn through torch.distributed.run (127.0.0.1 rendezvous), the longest-first
shard, the per-rank early return, the record gather, the all-reduce that counts the ranks, and the shape of the JSON line.  The
8-GPU run is the driver's; this removes the ways it could fail on first contact."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ, MIVOS_DIST_BACKEND="gloo", **(env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_config4_two_ranks_self_spawn_and_line_shape():
    two = _run(["--gpus", "2", "--config", "4", "--clips", "40", "--stub-engine"])
    one = _run(["--gpus", "1", "--config", "4", "--clips", "40", "--stub-engine"])
    for d, n in ((two, 2), (one, 1)):
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["unit"] == "frames/s" and d["higher_is_better"] is True
        assert d["stub_engine"] is True and d["config"]["baseline_config"] == 4 and d["config"]["clips"] == 40
        assert d["steps"] == sum(r["frames"] for r in d["per_rank"]) and [r["rank"] for r in d["per_rank"]] == list(range(n))
        assert sum(r["clips"] for r in d["per_rank"]) == 40 and d["ms_per_step"] > 0 and d["value"] > 0
    assert two["config"]["suite_checksum"] == one["config"]["suite_checksum"]           # every clip once, same results
    assert two["dist_backend"] == "gloo" and two["gloo_ranks"] == 2 and two["rccl_ranks"] is None and one["dist_backend"] is None
    loads = [r["frames"] for r in two["per_rank"]]
    assert max(loads) / (sum(loads) / 2) < 1.1                                            # longest-first balance


def test_default_config4_is_the_whole_suite():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--clips", type=int, default=474' in src and spec is not None
