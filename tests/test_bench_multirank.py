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


def test_config3_two_ranks_stub_engine_weak_scaling_line():
    """`bench.py --gpus 2 --config 3 --stub-engine` (the headline configuration's multi-rank path: one clip per rank, weak scaling): self-spawn, the
    window placed across the plain / fused boundary of the session, max-over-ranks clock, record gather, rank count - with a numpy stand-in engine."""
    two = _run(["--gpus", "2", "--config", "3", "--stub-engine", "--steps", "20", "--warmup", "5"])
    one = _run(["--gpus", "1", "--config", "3", "--stub-engine", "--steps", "20", "--warmup", "5"])
    for d, n in ((two, 2), (one, 1)):
        assert d["n_gpus"] == n and d["scaling"] == "weak" and d["steps"] == 20 and d["warmup"] == 5 and d["stub_engine"] is True
        assert d["config"]["baseline_config"] == 3 and d["config"]["clips_in_flight_per_gpu"] == 1
        assert d["config"]["untimed_steps_before_warmup"] == 137 + 54 and d["config"]["plain_steps"] == 10 and d["config"]["fused_steps"] == 10
        assert [r["rank"] for r in d["per_rank"]] == list(range(n)) and all(r["steps"] == 20 for r in d["per_rank"])
        assert d["full_session"]["steps"] == 137 and d["full_session"]["plain"] == 69 and d["full_session"]["fused"] == 68
        assert d["sustained"]["steps"] == 8 * 137 and d["sustained"]["sessions_in_flight"] == 1
    assert two["dist_backend"] == "gloo" and two["gloo_ranks"] == 2 and one["dist_backend"] is None


def test_window_phase_places_short_windows_across_the_plain_fused_boundary():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    c3, c2 = bench.CONFIGS[3], bench.CONFIGS[2]
    assert bench.window_phase(c3, 70, 5, 20) == (137 + 54, 10, 10)           # the driver's flags: one whole session, then steps 59..79 of the next = 10 plain + 10 fused
    assert bench.window_phase(c3, 70, 137, 8 * 137) == (0, 8 * 69, 8 * 68)   # whole sessions start at the session's start
    assert bench.window_phase(c3, 70, 0, 137) == (0, 69, 68)
    assert bench.window_phase(c2, 70, 5, 20) == (0, 20, 0)                   # one interaction: nothing to straddle
    pre, plain, fused = bench.window_phase(c3, 70, 5, 20, lanes=2)           # two sessions in lockstep reach their boundaries at combined step 138
    assert (pre + 5 + plain) % (2 * 137) == 2 * 69 and plain == 10 and fused == 10
    for w, k in ((0, 1), (3, 7), (50, 60), (100, 30), (136, 136)):
        pre, plain, fused = bench.window_phase(c3, 70, w, k)
        assert plain + fused == k and 137 <= pre < 2 * 137 and abs(plain - k * 69 / 137) <= 1
