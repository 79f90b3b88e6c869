"""Suite runner (mivos_amd/eval_suite.py) on CPU: synthetic config-4 suite, world_size-2 gloo run with a stub engine —
every clip is processed exactly once, the aggregate equals the single-process run."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from mivos_amd import eval_suite as ES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent("""
    import numpy as np

    class StubCore:
        '''Deterministic stand-in for InferenceCore: "propagates" by rolling the first mask.'''
        def __init__(self, spec):
            self.spec = spec
        def interact(self, mask, idx):
            s = self.spec
            out = np.stack([np.roll(mask, t, axis=1) for t in range(s.frames)], 0).astype(np.uint8)
            return out
        def interact_steps(self, mask, idx):
            for t in range(self.spec.frames - 1):
                TRACE.append((self.spec.clip_id, t))
                yield t
            return self.interact(mask, idx)

    TRACE = []

    def factory(spec):
        r = np.random.RandomState(spec.seed)
        first = (r.rand(12, 20) * (spec.objects + 1)).astype(np.uint8)
        return StubCore(spec), first
""")


def test_suite_spec_and_resize_rule():
    assert ES.yv_480p_size(720, 1280) == (480, 853)        # yv_test_dataset.py:102-109: landscape
    assert ES.yv_480p_size(1280, 720) == (853, 480)        # portrait: h*480//w
    assert ES.yv_480p_size(480, 854) == (480, 854)
    specs = ES.synthetic_suite(474)
    assert len(specs) == 474 and specs == ES.synthetic_suite(474)
    assert all(s.frames % 5 == 0 and 20 <= s.frames <= 180 and 1 <= s.objects <= 5 for s in specs)
    assert {s.objects for s in specs} == {1, 2, 3, 4, 5} and (specs[0].height, specs[0].width) == (480, 853)


def test_single_process_suite_and_summary():
    ns = {}
    exec(STUB, ns)
    specs = ES.synthetic_suite(9)
    seen = []
    recs = ES.run_suite(specs, ns["factory"], on_clip=lambda s, m: seen.append((s.clip_id, m.shape[0])))
    assert sorted(c for c, _ in seen) == list(range(9)) and all(t == specs[c].frames for c, t in seen)
    s = ES.summarize(recs, 9)
    assert s["clips"] == 9 and s["frames"] == sum(sp.frames - 1 for sp in specs)
    dup = recs + recs[:1]
    try:
        ES.summarize(dup, 9)
        raise AssertionError("duplicate clip not detected")
    except RuntimeError:
        pass


def test_two_lanes_interleave_clips_and_reproduce_the_sequential_suite():
    """run_suite(lanes=2): two clips in flight, advanced in turn one frame at a time, every lane inside its own context; the records
    (checksums, frames) equal the one-clip-at-a-time run, each clip is processed once, and the clips' seconds add up to the wall clock."""
    import contextlib
    ns = {}
    exec(STUB, ns)
    specs = ES.synthetic_suite(7)
    ref = ES.run_suite(specs, ns["factory"])
    entered = []

    @contextlib.contextmanager
    def lane_ctx(lane):
        entered.append(lane)
        yield

    seen = []
    recs = ES.run_suite(specs, ns["factory"], lanes=2, lane_ctx=lane_ctx, on_clip=lambda s, m: seen.append(s.clip_id))
    assert sorted(seen) == list(range(7)) and set(entered) == {0, 1}
    assert [(r["clip"], r["frames"], r["checksum"]) for r in recs] == [(r["clip"], r["frames"], r["checksum"]) for r in ref]
    assert ES.summarize(recs, 7)["checksum"] == ES.summarize(ref, 7)["checksum"]
    trace = ns["TRACE"]
    # two different clips alternate frame by frame while both are in flight
    switches = sum(1 for a, b in zip(trace, trace[1:]) if a[0] != b[0])
    assert switches > len(trace) // 2, (switches, len(trace))
    for cid in range(7):                                  # and every clip's frames come in order
        ts = [t for c, t in trace if c == cid]
        assert ts == list(range(specs[cid].frames - 1))
    assert all(r["seconds"] > 0 and r["lanes"] == 2 for r in recs)


def test_two_rank_gloo_suite(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from mivos_amd import shard, eval_suite as ES
    """) + STUB + textwrap.dedent("""
        rank, world, local = shard.init_distributed(backend="gloo")
        specs = ES.synthetic_suite(23)
        recs = ES.run_suite(specs, factory, rank, world)
        shard.barrier()
        allrecs = shard.gather_records(recs)
        if rank == 0:
            s = ES.summarize(allrecs, 23)
            ref = ES.summarize(ES.run_suite(specs, factory), 23)
            assert s["checksum"] == ref["checksum"] and s["frames"] == ref["frames"], (s, ref)
            loads = {}
            for r in allrecs:
                loads[r["rank"]] = loads.get(r["rank"], 0) + shard.clip_cost(r["frames"], r["objects"])
            assert set(loads) == {0, 1} and max(loads.values()) <= 1.2 * min(loads.values()), loads
            print("SUITE_OK", s["clips"], s["frames"])
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert "SUITE_OK 23" in out.stdout, out.stdout + out.stderr


def test_generator_suite_with_a_stub_generator():
    """run_generator_suite (generate_fusion.py:68-120 over a sharded suite): reference frames every `separation`, usable-object
    filter (> 100 pixels, at most 5), two-sided propagation over the whole clip; both ranks' shares together = the single run."""
    import numpy as np

    class StubGen:
        def __init__(self, spec):
            self.spec, self.calls = spec, []
        def reset(self, k):
            self.k = k
        def interact_mask(self, mask, idx, left, right):
            self.calls.append((idx, left, right, self.k))
            return np.full((self.k + 1, self.spec.frames, 6, 8), idx / 255.0, dtype=np.float32)

    gens = {}
    def factory(spec):
        gt = np.zeros((spec.frames, spec.objects, 1, 20, 20), dtype=np.float32)
        gt[:, 0] = 1.0                                     # object 0 fills the frame, the others are empty (not usable)
        if spec.objects > 1:
            gt[:, 1, :, :5, :5] = 1.0                      # 25 pixels: below the 10 x 10 threshold
        gens[spec.clip_id] = StubGen(spec)
        return gens[spec.clip_id], gt

    specs = ES.synthetic_suite(7)
    recs = ES.run_generator_suite(specs, factory, separation=5)
    assert sorted(r["clip"] for r in recs) == list(range(7))
    for r in recs:
        s = specs[r["clip"]]
        n_ref = len(range(0, s.frames, 5))
        assert r["reference_frames"] == n_ref and r["frames"] == n_ref * (s.frames - 1)
        assert gens[s.clip_id].calls == [(f, 0, s.frames - 1, 1) for f in range(0, s.frames, 5)]
    two = ES.run_generator_suite(specs, factory, 0, 2) + ES.run_generator_suite(specs, factory, 1, 2)
    assert ES.summarize(two, 7)["checksum"] == ES.summarize(recs, 7)["checksum"]
