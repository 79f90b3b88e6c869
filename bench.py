#!/usr/bin/env python
"""Benchmark of the MI355X propagation + fusion engine (BASELINE.json metric: propagated frames/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

One "step" = one propagated frame = one iteration of the reference's do_pass loop (`inference_core.py:165`): query
features (cached per frame), per-object top-k memory read, mask decoder, aggregation, memorize into the bank and - between
two interacted frames - difference-aware fusion.

Workloads (fixed, `--steps/--warmup` only choose WHICH steps are timed):
  --config 3 (default; BASELINE config 3, the one the metric is quoted on): 480x854 clip of 70 frames, K=5 objects,
             top_k=50, mem_freq=5; a session = interact(0) [69 plain frames] + interact(69) [68 fused frames] = 137 steps,
             repeated with a fresh InferenceCore while more steps are requested.
  --config 2: 480x854, 70 frames, K=1, top_k=20: session = interact(0) = 69 steps.
  --config 5: 1080x1920, K=3, `--frames` (default 1000) frames, unbounded bank (mem_freq=5 -> T grows to 200): one session.
  --config 4: synthetic YouTube-VOS-like suite (`--clips` of the 474 clips; lengths 5*U{4..36}, K~U{1..5}) sharded over
             the ranks by mivos_amd.eval_suite; strong scaling over the fixed suite (default: all 474 clips, ~3.5 min on one GPU).
Timed region: exactly `--steps` steps after `--warmup` untimed ones, bracketed by barrier + torch.cuda.synchronize(); at
its start every query feature that was encoded ahead of its frame's turn is dropped (`prepaid_frames: 0`).  Inputs are
resident in HBM before the clock starts.  `value` is measured with ONE session (clip) in flight per GPU - the reference's
interactive workload; a window shorter than a session is placed across the plain / fused boundary of the session
(window_phase).  Extra fields: `several_clips_in_flight` (--lanes sessions on as many HIP streams: suite throughput),
`full_session`, `sustained` (>= 5 s with sampled shader clock and package power).  Multi-GPU: one process per GPU; `python bench.py --gpus N` spawns its own ranks
through torch.distributed.run when it was not launched by it.  Configs 2/3/5: one clip per rank (weak scaling); config 4:
the suite is split (strong scaling).  value = steps of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
# fp16 MFMA dense peak (2.5 PFLOP/s) / 3 MFMA products per algorithmic multiply-add of the error-compensated
# f16x3 convolution = the roofline of that kernel in ALGORITHMIC (fp32-equivalent) FLOP/s
F16X3_PEAK_TFLOPS = 2500.0 / 3
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s (6.3 achievable)
VARIANT_NAMES = {0: "conv_igemm_kernel<128,128,2,2>", 1: "conv_igemm_kernel<64,64,2,2>",
                 2: "conv_igemm_kernel<128,32,4,1>", 3: "conv_igemm_kernel<128,64,2,2>", 4: "conv_cout1_kernel",
                 10: "conv_f16x3_kernel<128,128,2,2>", 11: "conv_f16x3_kernel<64,64,2,2>", 12: "conv_f16x3_kernel<128,32,4,1>",
                 13: "conv_f16x3_kernel<128,64,2,2>", 14: "conv_cout1_kernel", 15: "conv_f16x3_pipe_kernel<256,256,2,4>",
                 16: "conv_f16x3_pipe_kernel<128,256,2,4>", 17: "conv_f16x3_pipe_kernel<128,128,4,2>",
                 18: "conv_f16x3_pipe_kernel<64,256,2,4>", 19: "conv3x3_n32_direct_kernel",
                 20: "conv_f16x3_pp_kernel<128,128,2,4,0>", 21: "conv_f16x3_pp_kernel<128,256,2,4,0>",
                 22: "conv_f16x3_pp_kernel<128,64,4,2,0>", 23: "conv_f16x3_pp_kernel<256,256,2,4,0>", 24: "conv_f16x3_pp_kernel<64,128,2,4,0>",
                 30: "fusion_net_forward (conv1 + 2 x fusion_resblock_kernel + fusion_head_kernel)", 90: "memread_select_kernel", 91: "memread_finalize_kernel"}
# clips in flight per GPU (see --lanes): a 480p frame of 1-5 objects leaves most of the 256 CUs idle at the 1/16-resolution layers, a second
# clip on a second stream fills them (profiles/r05a_suite_lanes_ab.txt: config 4 199.7 -> 233.3 -> 241.4 frames/s at 1 / 2 / 3 lanes, identical masks)
GATE_FACTOR = 1.5        # tests/test_gpu_engine.py::ARBITRATION_FACTOR
# profiles/r05c_lanes_ab.txt: config 3 (5 objects) 206 -> 236 -> 221 frames/s at 1 / 2 / 3 lanes, config 2 (1 object) 404 -> 511 -> 593 -> 489 at 1..4,
# config 4 (1-5 objects) 200 -> 237 -> 247 -> 242
DEFAULT_LANES = {2: 3, 3: 2, 4: 3}
if os.environ.get("MIVOS_BENCH_LANES"):
    DEFAULT_LANES = {c: int(os.environ["MIVOS_BENCH_LANES"]) for c in (2, 3, 4)}
CONFIGS = {
    2: dict(name="davis480p_single_object", height=480, width=854, frames=70, objects=1, top_k=20, interactions=(0,)),
    3: dict(name="davis480p_multiobject_fusion", height=480, width=854, frames=70, objects=5, top_k=50, interactions=(0, -1)),
    5: dict(name="hd1080p_long_clip_unbounded_bank", height=1080, width=1920, frames=1000, objects=3, top_k=50, interactions=(0,)),
}


def gpu_sync(torch):
    """torch.cuda.synchronize() where there is a GPU (the --stub-engine plumbing runs have none)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class StubCore:
    """PLUMBING TEST ONLY (--stub-engine, tests/test_bench_multirank.py): the call surface of InferenceCore that this file uses (interact /
    interact_steps with step_cb, drop_lookahead, propagated_frames) on numpy, so that the multi-rank path of configs 2 / 3 / 5 - self-spawn,
    one clip per rank, window placement, max-over-ranks clock, record gather, rank count, line shape - runs with world size 2 without a GPU."""

    def __init__(self, frames, interacted=None):
        self.t, self.propagated_frames, self.interacted = frames, 0, set()

    def drop_lookahead(self):
        return 0

    def interact_steps(self, mask, idx, step_cb=None):
        import numpy as np
        self.interacted.add(idx)
        lo = max([i for i in self.interacted if i < idx] + [-1])
        hi = min([i for i in self.interacted if i > idx] + [self.t])
        for _ in range(hi - lo - 2):
            time.sleep(2e-4)
            self.propagated_frames += 1
            if step_cb is not None:
                step_cb()
            yield
        return np.zeros((self.t, 4, 4), dtype=np.uint8)

    def interact(self, mask, idx, step_cb=None):
        g = self.interact_steps(mask, idx, step_cb=step_cb)
        while True:
            try:
                next(g)
            except StopIteration as e:
                return e.value


LANE_CLIPS = {}           # lane -> (images, gt) for lanes >= 1 of the several-clips-in-flight measurements (lane 0 and the headline use the main clip)
CORE_FACTORY = None       # --stub-engine: callable(images, objects) -> StubCore; None = mivos_amd.inference_core.InferenceCore on the GPU


def make_core(prop, fuse, images, objects, mem_freq, dev):
    if CORE_FACTORY is not None:
        return CORE_FACTORY(images, objects)
    from mivos_amd.inference_core import InferenceCore
    return InferenceCore(prop, fuse, images, objects, mem_profile=0, mem_freq=mem_freq, device=dev)


class StepClock:
    """step_cb hook: after `warmup` steps drops the look-ahead of the running core, synchronises and stamps t0; stamps
    t1 after exactly `steps` more."""

    def __init__(self, warmup, steps, profile_every, ops, shard, torch):
        self.warmup, self.steps, self.every, self.ops, self.shard, self.torch = warmup, steps, profile_every, ops, shard, torch
        self.n, self.t0, self.t1, self.samples, self.cores, self.dropped = 0, None, None, [], [], 0

    def _stamp(self):
        gpu_sync(self.torch)
        self.shard.barrier()
        gpu_sync(self.torch)
        return time.perf_counter()

    def arm(self):
        if self.n == self.warmup and self.t0 is None:
            self.dropped = sum(c.drop_lookahead() for c in self.cores if c is not None)
            self.t0 = self._stamp()
        timed = self.t0 is not None and self.t1 is None
        # sample every `every`-th timed step with HIP events around each conv / memory-read launch
        self.ops.PROFILE = self.samples if (timed and self.every and (self.n - self.warmup) % self.every == 0) else None

    def __call__(self):
        self.n += 1
        if self.n == self.warmup + self.steps and self.t1 is None:
            self.t1 = self._stamp()
        self.arm()

    @property
    def done(self):
        return self.t1 is not None


PROFILES_DIR = os.path.join(ROOT, "profiles")


def _csrc_fingerprint():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from csrc_fingerprint import csrc_fingerprint
        return csrc_fingerprint(ROOT)
    finally:
        sys.path.pop(0)


def _fresh(table, path):
    """A committed PMC record is used only for the kernels it was read from: its `_meta.csrc_fingerprint` (scripts/csrc_fingerprint.py,
    written by the summary scripts) must equal this tree's.  Otherwise the lookup answers with a `stale` marker instead of numbers."""
    meta = table.get("_meta") or {}
    if meta.get("csrc_fingerprint") == _csrc_fingerprint():
        return None
    return dict(stale=True, source=os.path.basename(path), recorded_for=meta.get("csrc_fingerprint"), tree=_csrc_fingerprint(),
                note="mivos_amd/csrc changed after this PMC pass (or the record predates fingerprints): numbers withheld, re-run scripts/profile_bench.sh")


def pmc_traffic(config, kernel_name):
    """HBM-side bytes per launch of the kernel instantiation `kernel_name` in bench config `config`, from the committed
    rocprofv3 PMC passes of THAT config (profiles/*config<N>*pmc_traffic.json, newest first: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same bench command, FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md §HBM).
    The instantiation must match exactly (all template arguments).  None when no committed pass matches: a traffic figure
    of another config or another kernel variant would be worse than none."""
    import glob
    import re
    key = re.sub(r"[ ,]", "", kernel_name)
    for path in sorted(glob.glob(os.path.join(PROFILES_DIR, f"*config{config}*pmc_traffic.json")), reverse=True):
        try:
            table = json.load(open(path))
        except Exception:
            continue
        for name, rec in table.items():
            if name != "_meta" and re.sub(r"[ ,]", "", name.replace("mivos::", "").replace(", false>", ">")) == key:     # (", false": the FOLD flag of round 6's instantiations)
                if _fresh(table, path) is not None:
                    return _fresh(table, path)
                out = dict(bytes_per_launch=int(rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"]),
                           read=int(rec["read_bytes_per_launch"]), write=int(rec["write_bytes_per_launch"]),
                           launches_profiled=int(rec.get("launches", 0)), source=os.path.basename(path))
                m = re.search(r"memread\d*_T(\d+)_", os.path.basename(path))
                if m:       # the profiler does not survive the 1000-frame command: one read of this config at a fixed bank depth instead
                    out["note"] = (f"PMC pass of scripts/memread_case.py (one read of this config's shape at a {m.group(1)}-frame bank = the "
                                   "session's mean bank depth), not of the whole bench command")
                return out
    return None


def pmc_mfma_util(config, kernel_name):
    """Matrix-pipe busy share of the kernel instantiation `kernel_name` in bench config `config` from the committed
    `rocprofv3 --pmc MfmaUtil` pass of that config (profiles/*config<N>*mfma_util.json, newest first; scripts/pmc_mfma_util.py):
    sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE x SIMDs), mean over the dispatches.  None when no pass matches."""
    import glob
    import re
    key = re.sub(r"[ ,]", "", kernel_name)
    for path in sorted(glob.glob(os.path.join(PROFILES_DIR, f"*config{config}*mfma_util.json")), reverse=True):
        try:
            table = json.load(open(path))
        except Exception:
            continue
        for name, rec in table.items():
            base = re.sub(r"^void ", "", name.replace("mivos::", "")).split("(")[0].replace(", false>", ">")
            if name != "_meta" and re.sub(r"[ ,]", "", base) == key:
                if _fresh(table, path) is not None:
                    return _fresh(table, path)
                return dict(mean_pct=rec["mfma_util_mean_pct"], min_pct=rec["min_pct"], max_pct=rec["max_pct"], dispatches=rec["dispatches"],
                            source=os.path.basename(path))
    return None


def bench_states(synthetic):
    """The seeded synthetic weights with the closed-loop conditioning of round 5 (synthetic.CLOSED_LOOP_CONDITIONING: two post-hoc gains; same
    shapes, same arithmetic, same speed).  On the unconditioned weights the reference's own fp32 and fp64 runs of the mini session disagree at
    IoU 0.998 (aggregate_wbg's fp32 logit round trip at saturated pixels), which made the line's parity block measure the fixture, not the engine."""
    return (synthetic.condition_state(synthetic.make_prop_state(0), **synthetic.CLOSED_LOOP_CONDITIONING), synthetic.make_fuse_state(0))


_LANE_STREAMS = {}


def lane_streams(torch, dev, lanes):
    """One HIP stream per lane, created once per process: the allocator pools and scratch buffers keyed by them stay warm between the measurements."""
    key = (str(dev), lanes)
    if key not in _LANE_STREAMS:
        _LANE_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
    return _LANE_STREAMS[key]


class ClockSampler:
    """Background samples of the GPU's shader clock and package power while a measurement runs (rocm-smi --showclocks --showpower --json every
    ~0.3 s from a thread; the tool is part of the ROCm image and readable by an ordinary user).  `summary()` is None-safe: a box without the tool
    yields {"available": false, ...} instead of numbers."""

    def __init__(self, device_index=0, period=0.3, enabled=True):
        import threading
        self.idx, self.period, self.samples, self.error, self.enabled = device_index, period, [], None, enabled
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        while not self._stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "-d", str(self.idx), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                card = next(iter(json.loads(txt).values()))
                sclk = re.search(r"(\d+)", card.get("sclk clock speed:", ""))
                power = next((v for k, v in card.items() if "Power" in k), None)
                self.samples.append((time.perf_counter(), int(sclk.group(1)) if sclk else None, float(power) if power is not None else None))
            except Exception as e:                   # no tool / no permission / unexpected format: the measurement itself must not suffer
                self.error = repr(e)[:120]
                return
            self._stop.wait(self.period)

    def __enter__(self):
        if self.enabled:                 # (rank 0 only: eight ranks need not run eight pollers)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.enabled:
            self._thread.join(timeout=6)

    def summary(self, t0, t1):
        """Statistics of the samples taken inside [t0, t1] (perf_counter stamps of the timed region)."""
        inside = [(c, p) for t, c, p in self.samples if t0 <= t <= t1]
        clk = sorted(c for c, _ in inside if c)
        pw = sorted(p for _, p in inside if p is not None)
        if not clk:
            return dict(available=False, reason=self.error or "no rocm-smi sample fell inside the timed region")
        return dict(available=True, samples=len(inside), sclk_mhz_median=clk[len(clk) // 2], sclk_mhz_min=clk[0], sclk_mhz_max=clk[-1],
                    package_power_w_median=pw[len(pw) // 2] if pw else None, package_power_w_max=pw[-1] if pw else None,
                    source="rocm-smi --showclocks --showpower, sampled from a host thread beside the timed region")


def window_phase(cfg, T, warmup, steps, lanes=1):
    """Where in the repeated session the timed window sits.  A config-3 session is 69 plain steps (interact(0)) followed by 68 fused steps
    (interact(69)); a window shorter than a session that started at step `warmup` of a session - the driver's --steps 20 --warmup 5 - would hold
    plain frames only (and a 2-6 frame bank).  `preroll` extra untimed steps are therefore run first so that the window straddles the plain /
    fused boundary in the session's own proportion (69 : 68) and contains what happens between the two interactions (argmax + D2H of the first,
    difference maps and memorize of the second); the preroll starts with one whole session (first-use costs of the fused half).  Windows of a whole session or more, and single-interaction configs, start at step `warmup` as
    before.  Returns (preroll, plain steps in the window, fused steps in the window)."""
    inter = len(cfg["interactions"])
    n_plain = lanes * (T - 1)                      # `lanes` sessions advance in lockstep, a step is a frame of any of them
    session = n_plain + (lanes * (T - 2) if inter > 1 else 0)
    if inter < 2 or steps >= session:
        preroll = 0
    else:
        # one whole untimed session first: everything the fused half uses for the first time in a process (FusionNet's plan and workspaces, the
        # attention kernels, the page-locked result buffer of the first finished interaction) must not be charged to a 20-step window
        preroll = session + (n_plain - warmup - int(round(steps * n_plain / session))) % session
    plain = sum(1 for s in range(preroll + warmup, preroll + warmup + steps) if (s % session) < n_plain)
    return preroll, plain, steps - plain


def event_pair_overhead(torch):
    """Seconds a HIP-event pair measures around NOTHING on a busy stream (timestamp writes + command-processor gaps):
    subtracted from every per-launch sample so that short launches (50 us) are not inflated by 10-15 %."""
    x = torch.zeros(1 << 20, device="cuda")
    pairs = []
    for _ in range(64):
        x.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in pairs)
    return t[len(t) // 2] * 1e-3


def ops_precision():
    from mivos_amd import ops
    return ops.CONV_PRECISION


def kernel_rooflines(samples, overhead=0.0, config=3, select_kernel=None):
    """Aggregate the HIP-event samples per kernel instantiation.  Returns (dominant conv kernel's roofline record, the
    memory-read affinity record, per-kernel table)."""
    agg = {}
    bound = {}            # per conv instantiation: launches split by the roofline that bounds their SHAPE (arithmetic intensity against the ridge point)
    ridge = F16X3_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    for variant, flops, e0, e1, shape in samples:
        a = agg.setdefault(variant, [0.0, 0.0, 0, 0.0])
        secs = max(e0.elapsed_time(e1) * 1e-3 - overhead, 1e-7)
        a[0] += flops
        a[1] += secs
        a[2] += 1
        if variant == 30:
            a[3] += shape[-1]
        elif variant < 90:
            m, cin, cout, k, stride, has_res = shape
            # one read of the input, the weights and the residual, one write of the output (4 B per element)
            nbytes = 4.0 * m * stride * stride * cin + 4.0 * cout * k * k * cin + 4.0 * m * cout * (2 if has_res else 1)
            a[3] += nbytes
            if variant >= 10:
                b = bound.setdefault(variant, {"mfma": [0.0, 0.0, 0.0, 0, 0.0], "hbm": [0.0, 0.0, 0.0, 0, 0.0]})["hbm" if flops / nbytes < ridge else "mfma"]
                b[0] += flops; b[1] += nbytes; b[2] += secs; b[3] += 1
                b[4] += max(flops / (F16X3_PEAK_TFLOPS * 1e12), nbytes / (HBM_PEAK_GBS * 1e9))      # the time the roofline allows this launch
        else:
            a[3] += shape[-1]
    if not agg:
        return None, None, {}
    total = sum(x[1] for x in agg.values())
    table = {VARIANT_NAMES[v]: dict(launches=a[2], avg_us=round(a[1] / a[2] * 1e6, 2), tflops=round(a[0] / a[1] / 1e12, 2),
                                    time_share=round(a[1] / total, 3)) for v, a in sorted(agg.items())}
    roof = None
    convs = {v: a for v, a in agg.items() if v < 90}
    if convs:
        v, (flops, secs, n, abytes) = max(convs.items(), key=lambda kv: kv[1][1])
        ach = flops / secs / 1e12
        peak = F16X3_PEAK_TFLOPS if v >= 10 else MFMA_F32_PEAK_TFLOPS
        roof = dict(bound="mfma", kernel=VARIANT_NAMES[v], achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s",
                    frac=round(ach / peak, 4), traffic=pmc_traffic(config, VARIANT_NAMES[v]), mfma_util_pmc=pmc_mfma_util(config, VARIANT_NAMES[v]), launches_sampled=n,
                    peak_note=("algorithmic (fp32-equivalent) FLOP/s; kernel issues 3 fp16 MFMA products per term: 2500/3"
                               if v >= 10 else "fp32 MFMA dense peak"),
                    avg_launch_us=round(secs / n * 1e6, 2), algorithmic_gflop_per_launch=round(flops / n / 1e9, 3),
                    algorithmic_bytes_per_launch=int(abytes / n))
        if v in bound:
            # The instantiation serves ~30 layer shapes, some MFMA-bound, some HBM-bound at 4 bytes per activation (the 1x1 expansions with a residual:
            # 129 600 x 64 -> 256 moves 300 MB for 4 GFLOP).  `frac` above prices every launch against the MFMA peak; this block prices each launch against
            # the roofline that bounds its SHAPE (arithmetic intensity vs the ridge at peak_flops / peak_bytes) and sums the times the roofline allows.
            grp, allowed = {}, 0.0
            for name, (f_, b_, t_, n_, allow_) in bound[v].items():
                allowed += allow_
                if n_:
                    grp[name] = (dict(launches=n_, time_share=round(t_ / secs, 3), achieved=round(f_ / t_ / 1e12, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(f_ / t_ / 1e12 / peak, 4))
                                 if name == "mfma" else
                                 dict(launches=n_, time_share=round(t_ / secs, 3), achieved=round(b_ / t_ / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(b_ / t_ / 1e9 / HBM_PEAK_GBS, 4),
                                      note="algorithmic bytes: input, weights, residual and output once, 4 B per element"))
            roof["by_bounding_roofline"] = dict(ridge_flop_per_byte=round(ridge, 1), mfma_bound_shapes=grp.get("mfma"), hbm_bound_shapes=grp.get("hbm"),
                                                roofline_time_over_measured_time=round(allowed / secs, 4))
    aff = None
    if 90 in agg:
        flops, secs, n, abytes = agg[90]
        ach = flops / secs / 1e12
        f16 = ops_precision() == "f16x3"
        peak = F16X3_PEAK_TFLOPS if f16 else MFMA_F32_PEAK_TFLOPS
        aff = dict(bound="mfma", kernel="memread_select_kernel<F16> / memread_select256_kernel from 200 k memory positions (error-compensated fp16 MFMA affinity on pre-split keys + streaming top-k)" if f16
                   else "memread_select_kernel (exact fp32 MFMA affinity + streaming top-k)",
                   achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(ach / peak, 4),
                   peak_note=("algorithmic (fp32-equivalent) FLOP/s against 2500/3 = the roofline of the pipe this kernel runs on (3 fp16 MFMA products "
                              "per term)" if f16 else "fp32 MFMA dense peak"),
                   traffic=pmc_traffic(config, select_kernel) if select_kernel else None, traffic_kernel=select_kernel,
                   mfma_util_pmc=pmc_mfma_util(config, select_kernel) if select_kernel else None, launches_sampled=n, avg_launch_us=round(secs / n * 1e6, 2),
                   algorithmic_gflop_per_launch=round(flops / n / 1e9, 3), algorithmic_bytes_per_launch=int(abytes / n),
                   hbm_gbs_algorithmic=round(abytes / secs / 1e9, 1),
                   note="FLOP = 2*K*n_mem*n_q*128 of the affinity matmul only; bytes = keys + queries read once")
        if 91 in agg:
            f2, s2, n2, b2 = agg[91]
            aff["finalize"] = dict(kernel="memread_finalize_kernel (merge + softmax + sparse value gather)", avg_launch_us=round(s2 / n2 * 1e6, 2),
                                   gathered_bytes_per_launch=int(b2 / n2), gather_gbs=round(b2 / s2 / 1e9, 1),
                                   note="k value rows of 2 KB per (object, query); rows are shared by neighbouring queries and the value bank fits the "
                                        "256 MB Infinity Cache at 480p, so this rate is cache-served, not an HBM figure")
    return roof, aff, table


def mean_iou(a, b, k):
    from mivos_amd.util.tensor_util import compute_np_iou
    return round(float(sum(compute_np_iou(a == j, b == j) for j in range(1, k + 1)) / k), 6)


def cpu_baseline(torch, cfg, images, gt, mem_freq, prop, fuse, dev, n_frames, with_fp64):
    """The CPU oracle (oracle/stm_oracle.py — bit-identical to the unmodified reference on this torch build, see
    tests/test_oracle_golden.py and oracle/make_golden.py; /root/reference itself does not exist on the GPU box) on a
    bounded sample of the same workload, and the engine on the same sample for parity.  Sample: a mini session on the first
    `n_frames` frames of the clip with the workload's interaction pattern (so config 3 times fused frames too)."""
    from oracle import stm_oracle as O
    from mivos_amd.inference_core import InferenceCore
    from mivos_amd.util import synthetic
    k, top_k = cfg["objects"], cfg["top_k"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # oneDNN/OpenMP scale poorly past a few dozen threads here
    sd, fsd = bench_states(synthetic)
    sub, sgt = images[:, :n_frames].cpu(), gt[:n_frames].cpu()
    order = [0] + ([n_frames - 1] if len(cfg["interactions"]) > 1 else [])
    core = O.OracleCore(sd, fsd, sub, k, mem_freq=mem_freq, top_k=top_k)
    refs, dt = [], 0.0
    for idx in order:
        t0 = time.perf_counter()
        refs.append((core.interact(sgt[idx], idx).copy(), core.prob.clone()))
        dt += time.perf_counter() - t0
    eng = InferenceCore(prop, fuse, sub, k, mem_freq=mem_freq, device=dev)
    outs = []
    for idx in order:
        outs.append((eng.interact(sgt[idx], idx).copy(), eng.prob.cpu()))
    # every propagated frame of every interaction is compared: after interact(0) frames 1..n-1, after interact(n-1) frames 1..n-2 (fused)
    done, spans = set(), []
    for idx in order:
        done.add(idx)
        spans.append([t for t in range(n_frames) if t not in done])
    pairs = [(n, t) for n, span in enumerate(spans) for t in span]
    ious = [mean_iou(outs[n][0][t:t + 1], refs[n][0][t:t + 1], k) for n, t in pairs]
    mism = sum(int((outs[n][0][t] != refs[n][0][t]).sum()) for n, t in pairs)
    parity = dict(frames_compared=len(pairs), compared="every propagated frame after every interaction of the mini session",
                  mean_iou_engine_vs_ref_fp32=round(float(sum(ious) / len(ious)), 6), min_iou_engine_vs_ref_fp32=round(float(min(ious)), 6),
                  mismatching_pixel_fraction=round(mism / (len(pairs) * outs[0][0][0].size), 6),
                  max_abs_dprob=round(max(float((outs[n][1][:, t] - refs[n][1][:, t]).abs().max()) for n, t in pairs), 6))
    if with_fp64:
        # fp64 run of the same algorithm = the arbitration truth: the engine has to stay as close to it as the reference's own
        # fp32 arithmetic does (tests/test_gpu_engine.py::fp64_gate: per frame e <= ARBITRATION_FACTOR r + 2.5e-4)
        c64 = O.OracleCore(sd, fsd, sub, k, mem_freq=mem_freq, top_k=top_k, dtype=torch.float64)
        r64s = []
        for idx in order:
            r64s.append((c64.interact(sgt[idx], idx).copy(), c64.prob.clone()))
        e = torch.stack([(outs[n][1][:, t].double() - r64s[n][1][:, t]).abs().max() for n, t in pairs])
        r = torch.stack([(refs[n][1][:, t].double() - r64s[n][1][:, t]).abs().max() for n, t in pairs])
        live = r > 0
        i32 = [mean_iou(refs[n][0][t:t + 1], r64s[n][0][t:t + 1], k) for n, t in pairs]
        i64 = [mean_iou(outs[n][0][t:t + 1], r64s[n][0][t:t + 1], k) for n, t in pairs]
        parity["fp64"] = dict(mean_iou_ref_fp32_vs_ref_fp64=round(float(sum(i32) / len(i32)), 6), min_iou_ref_fp32_vs_ref_fp64=round(float(min(i32)), 6),
                              mean_iou_engine_vs_ref_fp64=round(float(sum(i64) / len(i64)), 6),
                              max_abs_dprob_ref_fp32_vs_fp64=round(float(r.max()), 6), max_abs_dprob_engine_vs_fp64=round(float(e.max()), 6),
                              per_frame_engine_vs_fp64=[round(float(x), 6) for x in e], per_frame_ref_fp32_vs_fp64=[round(float(x), 6) for x in r],
                              worst_frame_ratio=round(float((e[live] / r[live]).max()), 3) if bool(live.any()) else 0.0,
                              median_frame_ratio=round(float((e[live] / r[live]).median()), 3) if bool(live.any()) else 0.0,
                              gate=f"per frame: |engine - fp64| <= {GATE_FACTOR} |reference_fp32 - fp64| + 2.5e-4",
                              gate_passed=bool((e <= GATE_FACTOR * r + 2.5e-4).all()))
    fused = core.propagated - (n_frames - 1) if len(order) > 1 else 0
    return dict(value=round(core.propagated / dt, 4), unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"mini session on the first {n_frames} frames of the same clip ({k} objects, top_k={top_k}): interact at {order}, "
                       f"{core.propagated} propagated frames ({fused} of them fused); oracle/stm_oracle.py on PyTorch-CPU fp32 "
                       f"(oracle == unmodified reference bit-exactly on this torch build: tests/test_oracle_golden.py)",
                seconds=round(dt, 2), host_cores=os.cpu_count()), parity


def run_sessions(torch, ops, shard, cfg, images, gt, prop, fuse, dev, mem_freq, warmup, steps, profile_every, lanes=1, preroll=0):
    """Repeat the configuration's session (fresh InferenceCore over the HBM-resident clip) until `warmup + steps` steps
    have run.  Returns (clock, masks of the first session's first interaction).

    lanes > 1: that many sessions are in flight on this GPU, each on its own HIP stream, advanced in turn one propagated frame at a
    time (InferenceCore.interact_steps) - what eval_suite.run_suite(lanes=...) does with the clips of a suite.  A step is still one
    propagated frame (of whichever session); the timed region still holds exactly `steps` of them."""
    clock = StepClock(warmup + preroll, steps, profile_every, ops, shard, torch)      # preroll: see window_phase
    T = images.shape[1]
    first = None
    if lanes <= 1:
        while not clock.done:
            core = make_core(prop, fuse, images, cfg["objects"], mem_freq, dev)
            clock.cores = [core]
            clock.arm()
            for i in cfg["interactions"]:
                idx = i % T
                out = core.interact(gt[idx], idx, step_cb=clock)
                if first is None:
                    first = out.copy()
            del core
    else:
        clock.cores = [None] * lanes
        firsts = [None] * lanes

        def session_loop(lane):
            while not clock.done:
                im_l, gt_l = LANE_CLIPS.get(lane, (images, gt))      # lanes > 0 work on clips of their own (other frames, other masks; same shape)
                core = make_core(prop, fuse, im_l, cfg["objects"], mem_freq, dev)
                clock.cores[lane] = core
                clock.arm()
                for i in cfg["interactions"]:
                    idx = i % T
                    out = yield from core.interact_steps(gt_l[idx], idx, step_cb=clock)
                    if firsts[lane] is None:
                        firsts[lane] = out.copy()
                    if clock.done:
                        break
                del core
        streams = lane_streams(torch, dev, lanes)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        loops = [session_loop(lane) for lane in range(lanes)]
        live = list(range(lanes))
        with ops.chip_share(lanes):
            while live:
                for lane in list(live):
                    with torch.cuda.stream(streams[lane]):
                        try:
                            next(loops[lane])
                        except StopIteration:
                            live.remove(lane)
        first = firsts[0]
    ops.PROFILE = None
    gpu_sync(torch)
    return clock, first


def run_full_session(torch, shard, cfg, images, gt, prop, fuse, dev, mem_freq, lanes=1):
    """ONE complete session of the configuration (fresh InferenceCore over the HBM-resident clip, nothing pre-encoded), timed
    from before the first interact() to after the last, whatever `--steps/--warmup` selected for the headline window: the
    driver's `--steps 20 --warmup 5` window only sees plain propagation against a 2-6 frame bank, the session also holds the
    fused half and the bank's growth ("69 + 68").  Includes the per-interaction work outside the
    do_pass loop (memorize of the interacted frame, final argmax + D2H of the masks).  lanes > 1: that many complete sessions in
    flight, one HIP stream each, advanced in turn frame by frame; steps = the frames of all of them."""
    T = images.shape[1]
    clips = [LANE_CLIPS.get(lane, (images, gt)) if lanes > 1 else (images, gt) for lane in range(lanes)]
    cores = [make_core(prop, fuse, clips[lane][0], cfg["objects"], mem_freq, dev) for lane in range(lanes)]
    gpu_sync(torch); shard.barrier(); gpu_sync(torch)
    t0 = time.perf_counter()
    per = []
    if lanes <= 1:
        core = cores[0]
        for i in cfg["interactions"]:
            before = core.propagated_frames
            core.interact(gt[i % T], i % T)
            per.append(core.propagated_frames - before)
    else:
        def session(core, gt_l):
            for i in cfg["interactions"]:
                yield from core.interact_steps(gt_l[i % T], i % T)
        streams = lane_streams(torch, dev, lanes)
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream())
        from mivos_amd import ops
        loops = [session(c, clips[lane][1]) for lane, c in enumerate(cores)]
        live = list(range(lanes))
        with ops.chip_share(lanes):
            while live:
                for lane in list(live):
                    with torch.cuda.stream(streams[lane]):
                        if next(loops[lane], "end") == "end":
                            live.remove(lane)
        per = [lanes * (T - 1)] + [lanes * (T - 2)] * (len(cfg["interactions"]) - 1)
    gpu_sync(torch); shard.barrier(); gpu_sync(torch)
    dt = time.perf_counter() - t0
    n = sum(c.propagated_frames for c in cores)
    return dict(value=round(n / dt, 3), unit="frames/s", ms_per_step=round(dt / n * 1e3, 3), steps=n, plain=per[0], fused=sum(per[1:]),
                seconds=round(dt, 4), sessions_in_flight=lanes,
                note="whole session(s) incl. memorize of the interacted frames and the final argmax + D2H; untimed by --steps/--warmup")


def hbm_budget_check(torch, dev, cfg, T, K, lanes, mem_freq):
    """First-contact guard of the multi-GPU run: what `lanes` sessions of this configuration keep resident per GPU (mem_profile 0) against the
    free HBM of THIS rank's device, with a one-line reason instead of an out-of-memory error somewhere inside the first session.  Per session:
    the padded clip (fp32), the probabilities [K+1, T], the cached query features + decoder skip branches of every frame (the dominant
    term: f16 1024 + f8 512 + f4 256 channels, the three skip maps, key / value, the compress block's partial sums - measured 90 MB per frame
    at 480p = 217 bytes per padded pixel), two memory banks (fp32 + split keys) of T / mem_freq + 3 slots."""
    nh, nw = (cfg["height"] + 15) // 16 * 16, (cfg["width"] + 15) // 16 * 16
    P = nh * nw
    per_session = T * P * (3 * 4 + (K + 1) * 4) + min(T, 106) * P * 217 + 2 * K * (T // mem_freq + 3) * (P // 256) * (128 * 2 + 512) * 4     # (the query cache holds <= 106 frames)
    need = lanes * per_session + (3 << 30)                       # + weights, packed plans, workspaces, the allocator's slack
    free, total = torch.cuda.mem_get_info(torch.device(dev))
    if need > free:
        raise SystemExit(f"bench.py: {lanes} session(s) of config {cfg['name']} ({T} frames of {nh}x{nw}, {K} objects) need ~{need / 1e9:.1f} GB resident on {dev}, "
                         f"{free / 1e9:.1f} of {total / 1e9:.1f} GB are free: lower --lanes / --frames")
    return need


def self_spawn(args_list, n):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks (one per GPU) through
    torch.distributed.run and pass their output through."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + args_list
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 1) // n))))      # (torchrun's own default is 1 per rank)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 8 sessions of configs 2/3, the whole clip of config 5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before (default: one session of configs 2/3, 8 steps of config 5)")
    ap.add_argument("--config", type=lambda v: v if v in ("s2m", "train") else int(v), default=3, choices=(2, 3, 4, 5, "s2m", "train"),
                    help="2 / 3 / 4 / 5: BASELINE configs; s2m: the scribble-to-mask step in front of the path (SURVEY 8(f)1); "
                         "train: FusionNet training iterations (SURVEY 8(f)4), data parallel over the ranks")
    ap.add_argument("--frames", type=int, default=None, help="clip length override (config 5: default 1000)")
    ap.add_argument("--objects", type=int, default=None)
    ap.add_argument("--top-k", type=int, default=None)
    ap.add_argument("--mem-freq", type=int, default=5)
    ap.add_argument("--clips", type=int, default=474, help="config 4: how many of the 474 suite clips to run (default: all; ~4 min on one GPU)")
    ap.add_argument("--lanes", type=int, default=None,
                    help="clips / sessions in flight per GPU, each on its own HIP stream and advanced in turn frame by frame (eval_suite.run_suite(lanes=...), "
                         "run_sessions(lanes=...)).  Configs 2 / 3 / 5: `value` is ALWAYS one session in flight; --lanes > 1 adds the labelled `several_clips_in_flight` "
                         "fields.  Config 4 (the suite): the lanes of run_suite.  Default: DEFAULT_LANES per config (2 for config 3, 3 for configs 2 / 4; env MIVOS_BENCH_LANES), else 1")
    ap.add_argument("--stub-engine", action="store_true",
                    help="PLUMBING TEST ONLY (tests/test_bench_multirank.py): config 4 with a numpy stand-in for InferenceCore, so that argument "
                         "parsing, self-spawn, sharding, the record gather and the JSON line can be exercised with world_size 2 on a machine "
                         "without a GPU; the line says stub_engine: true and its value is not a measurement")
    ap.add_argument("--generator", action="store_true",
                    help="config 4 as the offline fusion-data generator (generate_fusion.py:68-120): every 5th frame of a clip is a reference "
                         "frame whose masks are propagated to both ends of the clip (FusionGenerator), sharded over the ranks like the suite")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames of the CPU-oracle mini session (0 = skip; default 6 = 9 propagated and compared frames for config 3, config 5: 2)")
    ap.add_argument("--no-cpu-fp64", dest="cpu_fp64", action="store_false",
                    help="skip the fp64 run of the oracle on the mini session (the arbitration truth of the parity block; ~4x the fp32 oracle's time)")
    ap.add_argument("--no-full-session", action="store_true", help="skip the extra whole-session measurement (configs 2/3) printed as full_session")
    ap.add_argument("--no-sustained", action="store_true", help="skip the extra >= 5 s measurement with sampled clocks / power (configs 2/3) printed as sustained")
    ap.add_argument("--other-configs", action=argparse.BooleanOptionalAction, default=os.environ.get("MIVOS_BENCH_EXTRAS", "1") != "0",
                    help="single-GPU config-3 runs only: append short runs of configs 2 / 4 / 5 on the same box as `other_configs` (default on; ~2.5 min)")
    ap.add_argument("--exact-f32-steps", type=int, default=None,
                    help="steps of the extra exact-fp32-MFMA measurement (CONV_PRECISION='f32'); default one session for config 3, 0 otherwise")
    ap.add_argument("--profile-every", type=int, default=None,
                    help="HIP-event sample every n-th timed step (0 = off); default 21 (>= 400 timed steps), 7 (>= 40) or 3: co-prime "
                         "with InferenceCore.QUERY_BATCH and mem_freq.  The events sit inside the timed region: measured on one box, "
                         "every 7th step costs 1.3 %% of the reported rate, every 21st 0.3 %%")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(sys.argv[1:], args.gpus))
    if args.lanes is None:
        args.lanes = DEFAULT_LANES.get(args.config, 1)
    # The other BASELINE configurations, each in a process of its own, BEFORE this process creates its GPU context: a second process that holds hardware queues
    # (even idle) costs a child with seven streams a third of its rate (config 2, three clips in flight: 350 vs 583 frames/s, same box, profiles/r06z_*).
    extras = None
    if (args.other_configs and args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.config == 3 and not args.stub_engine
            and (args.cpu_frames is None or args.cpu_frames > 1)):      # (the full line only: tuning runs pass --cpu-frames 0)
        extras = other_configs()

    import torch
    torch.set_grad_enabled(False)
    from mivos_amd import ops, shard
    from mivos_amd.model.fusion_net import FusionNet
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    from mivos_amd.util import synthetic

    rank, world, local = shard.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # host threads: N ranks on one node must not each start a thread per host core (256 on the GPU boxes) for torch's CPU ops
    # (clip generation, the one-hot masks); the CPU-oracle leg (rank 0 of a single-process run only) sets its own count
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 1) // max(world, 1))))
    if args.stub_engine:
        if args.config == 4:
            return bench_suite(args, torch, ops, shard, rank, world, "cpu")
        if args.config not in (2, 3, 5):
            raise SystemExit("--stub-engine is the plumbing test of configs 2 / 3 / 4 / 5")
        global CORE_FACTORY
        CORE_FACTORY = lambda images, objects: StubCore(images.shape[1])
        dev, args.lanes = "cpu", 1
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
        local = local % torch.cuda.device_count()      # (more ranks than GPUs only in the MIVOS_DIST_BACKEND=gloo plumbing test)
        torch.cuda.set_device(local)
        dev = f"cuda:{local}"

    if args.config == 4:
        return bench_suite(args, torch, ops, shard, rank, world, dev)
    if args.config == "s2m":
        return bench_s2m(args, torch, ops, shard, rank, world, dev)
    if args.config == "train":
        return bench_train(args, torch, ops, shard, rank, world, dev, local)

    cfg = dict(CONFIGS[args.config])
    for key, val in (("frames", args.frames), ("objects", args.objects), ("top_k", args.top_k)):
        if val is not None:
            cfg[key] = val
    K, T = cfg["objects"], cfg["frames"]
    session = sum((T - 1) if n == 0 else (T - 2) for n in range(len(cfg["interactions"])))
    warmup = args.warmup if args.warmup is not None else (8 if args.config == 5 else session)
    steps = args.steps if args.steps is not None else (session - warmup if args.config == 5 else 8 * session)
    cpu_frames = args.cpu_frames if args.cpu_frames is not None else (2 if args.config == 5 else 6)
    exact_steps = args.exact_f32_steps if args.exact_f32_steps is not None else (session - 8 if args.config == 3 else 0)

    if args.stub_engine:
        prop = fuse = None
        images, gt = torch.zeros((1, T, 1, 1, 1)), [None] * T
        exact_steps = cpu_frames = 0
    else:
        hbm_budget_check(torch, dev, cfg, T, K, max(1, args.lanes), args.mem_freq)
        prop, fuse = PropagationNetwork(top_k=cfg["top_k"]), FusionNet()
        prop.load_state_dict(bench_states(synthetic)[0])
        fuse.load_state_dict(bench_states(synthetic)[1])
        prop, fuse = prop.to(dev).eval(), fuse.to(dev).eval()
    if args.stub_engine:
        pass
    elif T * cfg["height"] * cfg["width"] > 70 * 480 * 864:
        images, gt = synthetic.synthetic_clip_device(T, cfg["height"], cfg["width"], K, seed=100 + rank, device=dev)
    else:
        images, gt = synthetic.synthetic_clip(T, cfg["height"], cfg["width"], K, seed=100 + rank)
        images, gt = images.to(dev), gt.to(dev)                # resident in HBM before the clock starts

    if args.lanes > 1 and not args.stub_engine and not os.environ.get("MIVOS_BENCH_SAME_CLIP"):      # the other lanes propagate OTHER clips (same shape, other seed): nothing of theirs is warm in a cache because lane 0 read it
        for lane in range(1, args.lanes):
            LANE_CLIPS[lane] = synthetic.synthetic_clip_device(T, cfg["height"], cfg["width"], K, seed=1000 * lane + 100 + rank, device=dev)
    if args.profile_every is None:
        args.profile_every = 21 if steps >= 400 else (7 if steps >= 40 else 3)
    # HEADLINE: ONE session (clip) in flight - the reference's workload is one interactive session at a time (and rounds 1-4 measured that).
    # The per-launch HIP-event samples (roofline) are taken in this pass too: beside another stream's kernels an event pair would also
    # measure the CUs the neighbour holds.  Several clips in flight per GPU (--lanes) is a separate, labelled field below.
    preroll, n_plain, n_fused = window_phase(cfg, T, warmup, steps)
    sustained_is_headline = args.config in (2, 3) and steps >= 5 * session        # the default `python bench.py` (8 sessions): the headline IS sustained
    with ClockSampler(local, enabled=rank == 0) as smi:
        clock, masks_first = run_sessions(torch, ops, shard, cfg, images, gt, prop, fuse, dev, args.mem_freq, warmup, steps, args.profile_every, lanes=1, preroll=preroll)
    elapsed = shard.max_over_ranks(clock.t1 - clock.t0, device=dev)
    headline_clocks = smi.summary(clock.t0, clock.t1) if rank == 0 else None
    samples = clock.samples
    several = None
    if args.lanes > 1:                      # the same window with `lanes` sessions in flight, each on its own HIP stream (the suite mode of eval_suite.run_suite)
        pre_n, pl_n, fu_n = window_phase(cfg, T, warmup, steps, lanes=args.lanes)
        cn, _ = run_sessions(torch, ops, shard, cfg, images, gt, prop, fuse, dev, args.mem_freq, warmup, steps, 0, lanes=args.lanes, preroll=pre_n)
        en = shard.max_over_ranks(cn.t1 - cn.t0, device=dev)
        several = dict(sessions_in_flight=args.lanes, value=round(world * steps / en, 3), unit="frames/s", ms_per_step=round(en / steps * 1e3, 3), steps=steps, warmup=warmup,
                       untimed_steps_before_warmup=pre_n, plain_steps=pl_n, fused_steps=fu_n,
                       note=f"same window with {args.lanes} sessions in flight per GPU, one HIP stream each, advanced in turn frame by frame (a step is one propagated frame of "
                            "any of them): aggregate throughput of independent clips (eval_suite.run_suite(lanes=...)), NOT the latency of one interactive session")
    recs = shard.gather_records([dict(rank=rank, steps=steps, seconds=round(clock.t1 - clock.t0, 6))])
    ranks_seen = shard.collective_ranks(dev)
    mem_gb = torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0
    full = None
    if args.config in (2, 3) and not args.no_full_session:
        full = run_full_session(torch, shard, cfg, images, gt, prop, fuse, dev, args.mem_freq)
        full["seconds"] = round(shard.max_over_ranks(full["seconds"], device=dev), 4)
        full["value"] = round(world * full["steps"] / full["seconds"], 3)
        full["ms_per_step"] = round(full["seconds"] / full["steps"] * 1e3, 3)
        if args.lanes > 1:                  # `lanes` complete sessions in flight, next to the single-session figure
            multi = run_full_session(torch, shard, cfg, images, gt, prop, fuse, dev, args.mem_freq, lanes=args.lanes)
            multi["seconds"] = round(shard.max_over_ranks(multi["seconds"], device=dev), 4)
            multi["value"] = round(world * multi["steps"] / multi["seconds"], 3)
            multi["ms_per_step"] = round(multi["seconds"] / multi["steps"] * 1e3, 3)
            full["several_clips_in_flight"] = multi
    # SUSTAINED: >= 5 s of whole sessions (8 sessions of config 3, 16 of config 2) with the shader clock and the package power sampled beside them - the
    # driver's 20-step window lasts 0.1 s and runs at the boost clock (2.4 GHz); after about a second under this load the chip settles at 1.9 GHz
    sustained = None
    if args.config in (2, 3) and not args.no_sustained:
        if sustained_is_headline:
            sustained = dict(value=round(world * steps / elapsed, 3), unit="frames/s", ms_per_step=round(elapsed / steps * 1e3, 3), steps=steps, seconds=round(elapsed, 3),
                             sessions_in_flight=1, clocks=headline_clocks, note="the headline window itself (>= 4 s)")
        else:
            n_sess = 8 if args.config == 3 else 16
            sustained = {}
            for ln in sorted({1, args.lanes}):
                with ClockSampler(local, enabled=rank == 0) as smi2:
                    cs, _ = run_sessions(torch, ops, shard, cfg, images, gt, prop, fuse, dev, args.mem_freq, session, n_sess * session, 0, lanes=ln)
                es = shard.max_over_ranks(cs.t1 - cs.t0, device=dev)
                rec = dict(value=round(world * n_sess * session / es, 3), unit="frames/s", ms_per_step=round(es / (n_sess * session) * 1e3, 3), steps=n_sess * session,
                           warmup=session, seconds=round(es, 3), sessions_in_flight=ln, clocks=smi2.summary(cs.t0, cs.t1) if rank == 0 else None)
                if ln == 1:
                    sustained.update(rec)
                    sustained["note"] = f"{n_sess} whole sessions after one untimed session, one session in flight (the headline's mode), clock and power sampled beside the timed region"
                else:
                    sustained["several_clips_in_flight"] = rec

    if sustained and sustained.get("value") and args.config in (2, 3) and not args.stub_engine:
        # the whole sustained region against the f16x3 roofline, at the nominal 2.4 GHz and at the shader clock the sampler saw (the matrix pipes' peak
        # scales with it: under this load the chip holds ~1.9 GHz, MI355X_MICROARCH.md quotes the peak at 2.4)
        base, extra = {2: (0.47, 0.03), 3: (1.34, 0.16)}[args.config]
        pf = (base * (T - 1) + (base + extra) * (T - 2) * (len(cfg["interactions"]) - 1)) / session
        ach = pf * sustained["value"] / world
        clk = (sustained.get("clocks") or {}).get("sclk_mhz_median")
        sustained["roofline_timed_region"] = dict(algorithmic_tflop_per_frame=round(pf, 4), achieved=round(ach, 2), peak=round(F16X3_PEAK_TFLOPS, 1), unit="TFLOP/s",
                                                  frac=round(ach / F16X3_PEAK_TFLOPS, 4), nominal_clock_mhz=2400, sampled_clock_mhz=clk,
                                                  frac_at_sampled_clock=round(ach / (F16X3_PEAK_TFLOPS * clk / 2400.0), 4) if clk else None)

    exact = None
    if exact_steps > 0 and rank == 0 and world == 1:
        old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f32"
        c2, _ = run_sessions(torch, ops, shard, cfg, images, gt, prop, fuse, dev, args.mem_freq, 8, exact_steps, 0)
        ops.CONV_PRECISION = old
        exact = dict(value=round(exact_steps / (c2.t1 - c2.t0), 3), unit="frames/s", ms_per_step=round((c2.t1 - c2.t0) / exact_steps * 1e3, 3),
                     steps=exact_steps, warmup=8, dtype="f32 (every convolution and the affinity on exact fp32 MFMA, CONV_PRECISION='f32')")
    if rank != 0:
        return
    if args.stub_engine:          # plumbing line: the keys of the real line that do not need a GPU
        print(json.dumps(dict(metric="plumbing test (numpy stand-in engine): not a measurement", value=round(world * steps / elapsed, 3), unit="frames/s", n_gpus=world, steps=steps,
                              warmup=warmup, ms_per_step=round(elapsed / steps * 1e3, 3), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="none", data="synthetic",
                              stub_engine=True, config=dict(workload=f"{cfg['name']} (BASELINE config {args.config}) with a numpy stand-in for InferenceCore", baseline_config=args.config,
                                                            session_steps=session, untimed_steps_before_warmup=preroll, plain_steps=n_plain, fused_steps=n_fused,
                                                            parallelism=f"sequence-sharded x{world}", clips_in_flight_per_gpu=1),
                              full_session=full, sustained=sustained, roofline=None, cpu_baseline=None, per_rank=recs, **ranks_seen)))
        return
    ev_overhead = event_pair_overhead(torch)
    # the select instantiation that serves this configuration's banks (csrc/memory_read.hip launch_select: the 256-query kernel
    # from 200 k memory positions, the wave-uniform skip of the append path from 32 k): the PMC traffic record must be ITS
    sel = None
    if ops.CONV_PRECISION == "f16x3":
        import ctypes
        from mivos_amd import _lib
        plan = (ctypes.c_int32 * 8)()
        n_q = ((cfg["height"] + 15) // 16) * ((cfg["width"] + 15) // 16)
        deepest = ((T - 1) // args.mem_freq + 2) * n_q                       # memory positions of the deepest bank a pass reads
        _lib.load().mivos_memory_read_plan(K, deepest, n_q, cfg["top_k"], 1, plan)
        sel = {256: "memread_select256_kernel", 128: "memread_select32_kernel<0,true>"}.get(
            plan[6], "memread_select_kernel<0,true,true>" if deepest >= 32768 else "memread_select_kernel<0,false,true>")
    roof, aff, table = kernel_rooflines(samples, ev_overhead, args.config, sel)
    if roof is not None:
        roof["event_pair_overhead_us"] = round(ev_overhead * 1e6, 2)
        roof["affinity"] = aff
        # the whole timed region against the same roofline: algorithmic FLOP of one propagated frame (every convolution + the
        # affinity, plain propagation; the fused half of a session adds 0.16 / 0.03 TFLOP per frame) x frames per second of the job.
        # With several clips in flight per GPU a per-launch event pair cannot see what the neighbour stream adds; this figure does.
        per_frame = {2: 0.47, 3: 1.34}.get(args.config)        # (config 5's 14.7 TFLOP is the figure at the full 200-frame bank; the bank grows over the run)
        if per_frame and ops.CONV_PRECISION == "f16x3":
            fused_extra = {2: 0.03, 3: 0.16}[args.config]        # attention + FusionNet of a fused frame (1.50 - 1.34 TFLOP at 5 objects)
            per_frame = round((n_plain * per_frame + n_fused * (per_frame + fused_extra)) / steps, 4)
            ach = per_frame * (steps / elapsed)
            roof["timed_region"] = dict(algorithmic_tflop_per_frame=per_frame, plain_steps=n_plain, fused_steps=n_fused, frames_per_second_per_gpu=round(steps / elapsed, 3), achieved=round(ach, 2),
                                        peak=round(F16X3_PEAK_TFLOPS, 1), unit="TFLOP/s", frac=round(ach / F16X3_PEAK_TFLOPS, 4),
                                        note="all kernels of the timed region together (HBM-bound pointwise kernels, selection, launch gaps included)")
        # In the timed region the fusion branch of frame t runs on a side stream BESIDE the propagation kernels of frame t + 1
        # (mivos_amd/inference_core.py: FUSE_ON_SIDE_STREAM): a HIP-event pair around a propagation launch then also measures the
        # CUs the other stream holds.  One extra, untimed session with the branch back on the main stream gives the same kernels'
        # rates without a neighbour: `isolated` is the kernel's own efficiency, `frac` above what it delivers in the running system.
        roof["isolated"] = None
        if len(cfg["interactions"]) > 1 and world == 1 and not args.no_full_session:
            try:
                from mivos_amd.inference_core import InferenceCore
                side, InferenceCore.FUSE_ON_SIDE_STREAM = InferenceCore.FUSE_ON_SIDE_STREAM, False
                try:
                    iso = []
                    core = InferenceCore(prop, fuse, images, K, mem_profile=0, mem_freq=args.mem_freq, device=dev)
                    ops.PROFILE = iso
                    for i in cfg["interactions"]:
                        core.interact(gt[i % T], i % T)
                    ops.PROFILE = None
                    torch.cuda.synchronize()
                    del core
                finally:
                    ops.PROFILE = None
                    InferenceCore.FUSE_ON_SIDE_STREAM = side
                r2, a2, _ = kernel_rooflines(iso, ev_overhead, args.config, sel)
                agg2 = {}
                for variant, flops, e0, e1, shape in iso:
                    x = agg2.setdefault(variant, [0.0, 0.0, 0])
                    x[0] += flops; x[1] += max(e0.elapsed_time(e1) * 1e-3 - ev_overhead, 1e-7); x[2] += 1
                same = [v for v in agg2 if VARIANT_NAMES[v] == roof["kernel"]]
                if same:
                    f, t_, n_ = agg2[same[0]]
                    roof["isolated"] = dict(kernel=roof["kernel"], achieved=round(f / t_ / 1e12, 2), frac=round(f / t_ / 1e12 / roof["peak"], 4),
                                            avg_launch_us=round(t_ / n_ * 1e6, 2), launches_sampled=n_,
                                            affinity_frac=(a2 or {}).get("frac"),
                                            affinity_avg_launch_us=(a2 or {}).get("avg_launch_us"),
                                            note="one untimed session, every launch sampled, fusion branch on the main stream (no concurrent kernels)")
            except Exception as e:                      # an attribution extra must never cost the line
                roof["isolated"] = dict(error=repr(e)[:200])
    if os.environ.get("MIVOS_BENCH_SHAPES"):          # debug: per-shape conv time inside the timed region
        agg = {}
        for variant, flops, e0, e1, shape in samples:
            a = agg.setdefault((variant,) + tuple(shape), [0.0, 0.0, 0])
            a[0] += flops; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        tot = sum(a[1] for a in agg.values())
        for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
            print(f"# {VARIANT_NAMES[key[0]]:38s} {str(key[1:]):48s} n={a[2]:4d} avg {a[1] / a[2] * 1e6:8.1f} us  {a[0] / a[1] / 1e12:6.1f} TF/s  "
                  f"share {a[1] / tot * 100:5.1f}%", file=sys.stderr)
    metric = {2: "propagated frames/sec, DAVIS-2017 480p single-object", 3: "propagated frames/sec, DAVIS-2017 480p multi-object",
              5: "propagated frames/sec, 1080p long clip"}[args.config]
    inter = [i % T for i in cfg["interactions"]]
    out = dict(metric=metric, value=round(world * steps / elapsed, 3), unit="frames/s", n_gpus=world, steps=steps, warmup=warmup,
               ms_per_step=round(elapsed / steps * 1e3, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f16x3 convolutions and memory-read affinity (fp16 hi+lo split operands, 3 fp16 MFMA products per term, fp32 accumulate); fusion attention exact f32",
               data="synthetic",
               config=dict(workload=f"{cfg['name']} (BASELINE config {args.config}): {cfg['height']}x{cfg['width']} clip of {T} frames per GPU, {K} objects, "
                                    f"top_k={cfg['top_k']}, mem_freq={args.mem_freq}; session = interact at frames {inter} = {session} steps "
                                    f"({T - 1} plain" + (f" + {T - 2} fused" if len(inter) > 1 else "") + f"); ONE session in flight; timed steps {(preroll + warmup) % session}..{(preroll + warmup) % session + steps} of "
                                    f"the repeated session = {n_plain} plain + {n_fused} fused"
                                    + (f" ({preroll} untimed steps - one whole session + {preroll - session} - run before the {warmup} warm-up steps so that the window straddles the plain / fused "
                                       f"boundary in the session's proportion and holds no first-use cost)" if preroll else ""),
                           baseline_config=args.config, objects=K, frames=T, height=cfg["height"], width=cfg["width"], top_k=cfg["top_k"],
                           mem_freq=args.mem_freq, session_steps=session, sessions_timed=round(steps / session, 3),
                           prepaid_frames=0, lookahead_entries_dropped_at_t0=clock.dropped, parallelism=f"sequence-sharded x{world}",
                           clips_in_flight_per_gpu=1, untimed_steps_before_warmup=preroll, plain_steps=n_plain, fused_steps=n_fused),
               clocks=headline_clocks, several_clips_in_flight=several, sustained=sustained, roofline=roof, full_session=full, conv_kernels=table, exact_f32=exact,
               hbm_peak_allocated_gb=round(mem_gb, 2), per_rank=recs, **ranks_seen)
    if world == 1 and cpu_frames > 1:
        out["cpu_baseline"], out["parity"] = cpu_baseline(torch, cfg, images, gt, args.mem_freq, prop, fuse, dev, cpu_frames, args.cpu_fp64)
    else:
        out["cpu_baseline"] = None
    if extras is not None:
        out["other_configs"] = extras
    print(json.dumps(out))


def other_configs():
    """The other BASELINE configurations on the SAME box, each as a short run of this script in a fresh process (clean allocator, nothing shared
    with the headline measurement), summarised: config 2 (one session + whole sessions), config 4 (the first 48 clips of the suite, 3 clips in
    flight), config 5 (the full 1000-frame 1080p clip, bank growing to 200 frames).  Only in the default single-GPU config-3 run (--other-configs /
    MIVOS_BENCH_EXTRAS=0 to skip), before the calling process touches the GPU; a run that fails or exceeds its time limit is recorded as an error string -
    it must never cost the line."""
    runs = {"config2": ["--config", "2", "--cpu-frames", "0"],        # default window: 8 sessions after one untimed session
            "config4_48clips": ["--config", "4", "--clips", "48"],
            "config5": ["--config", "5", "--cpu-frames", "0"]}
    res = {}
    # A throw-away child first: the FIRST process that drives several streams on a fresh box reads low in everything that allocates on them (same box, same command,
    # first vs fifth process: three clips in flight 348 vs 580 frames/s, one fresh session 234 vs 421; scripts/gpu/r7w.sh) - that is the box warming up, not the engine.
    try:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "2", "--steps", "138", "--warmup", "69", "--cpu-frames", "0", "--no-sustained"],
                       capture_output=True, text=True, timeout=300)
    except Exception:
        pass
    for name, extra in runs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra, capture_output=True, text=True, timeout=300)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                res[name] = dict(error=(r.stderr or r.stdout)[-300:])
                continue
            d = json.loads(lines[-1])
            rec = dict(value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], steps=d["steps"], warmup=d["warmup"], workload=d["config"]["workload"][:160],
                       clips_in_flight_per_gpu=d["config"].get("clips_in_flight_per_gpu"), wall_seconds=round(time.perf_counter() - t0, 1))
            for k in ("several_clips_in_flight", "full_session"):
                if d.get(k):
                    rec[k] = {kk: d[k].get(kk) for kk in ("value", "ms_per_step", "steps", "sessions_in_flight")}
            aff = ((d.get("roofline") or {}).get("affinity") or {})
            if aff:
                rec["affinity"] = {kk: aff.get(kk) for kk in ("frac", "achieved", "avg_launch_us")}
            res[name] = rec
        except Exception as e:
            res[name] = dict(error=repr(e)[:300])
    return res


def bench_suite(args, torch, ops, shard, rank, world, dev):
    """--config 4: the synthetic YouTube-VOS-like suite, split over the ranks (strong scaling)."""
    from mivos_amd import eval_suite as ES
    specs = ES.synthetic_suite(474)[:args.clips]
    if args.stub_engine:
        import numpy as np

        class StubCore:                     # "propagates" by rolling the first mask: deterministic, world-size independent
            def __init__(self, spec):
                self.spec = spec

            def interact(self, mask, idx):
                return np.stack([np.roll(mask, t, axis=1) for t in range(self.spec.frames)], 0).astype(np.uint8)

            def interact_steps(self, mask, idx):
                for _ in range(self.spec.frames - 1):
                    yield
                return self.interact(mask, idx)

        def factory(spec):
            r = np.random.RandomState(spec.seed)
            return StubCore(spec), (r.rand(12, 20) * (spec.objects + 1)).astype(np.uint8)
        sync = lambda: None
    else:
        from mivos_amd.inference_core import InferenceCore
        from mivos_amd.model.fusion_net import FusionNet
        from mivos_amd.model.propagation.prop_net import PropagationNetwork
        from mivos_amd.util import synthetic
        prop, fuse = PropagationNetwork(top_k=args.top_k or 50), FusionNet()
        prop.load_state_dict(synthetic.make_prop_state(0))
        fuse.load_state_dict(synthetic.make_fuse_state(0))
        prop, fuse = prop.to(dev).eval(), fuse.to(dev).eval()
        if args.generator:
            return bench_generator(args, torch, shard, ES, specs, prop, rank, world, dev)

        def factory(spec):
            images, gt = synthetic.synthetic_clip_device(spec.frames, spec.height, spec.width, spec.objects, seed=spec.seed, device=dev)
            return InferenceCore(prop, fuse, images, spec.objects, mem_profile=0, mem_freq=args.mem_freq, device=dev), gt[0]
        sync = torch.cuda.synchronize

    lane_ctx = ES.stream_lanes(dev, args.lanes) if args.lanes > 1 and not args.stub_engine else None
    # warm-up (untimed): one clip per lane, so that every lane's stream has its allocator pool and scratch buffers before the clock starts
    ES.run_suite([ES.ClipSpec(-1 - i, 12, 3, 480, 853, 7 + i) for i in range(max(1, args.lanes))], factory, 0, 1, sync=sync, lanes=args.lanes, lane_ctx=lane_ctx)
    sync(); shard.barrier(); sync()
    t0 = time.perf_counter()
    recs = ES.run_suite(specs, factory, rank, world, sync=sync, lanes=args.lanes, lane_ctx=lane_ctx)
    sync(); shard.barrier(); sync()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev)
    allrecs = shard.gather_records(recs)
    ranks_seen = shard.collective_ranks(dev)
    if rank != 0:
        return
    s = ES.summarize(allrecs, len(specs))
    per_rank = {}
    for r in allrecs:
        a = per_rank.setdefault(r["rank"], dict(rank=r["rank"], clips=0, frames=0, seconds=0.0))
        a["clips"] += 1; a["frames"] += r["frames"]; a["seconds"] = round(a["seconds"] + r["seconds"], 4)
    print(json.dumps(dict(
        metric="propagated frames/sec, YouTube-VOS-like suite sharded over the GPUs", value=round(s["frames"] / elapsed, 3), unit="frames/s",
        n_gpus=world, steps=s["frames"], warmup=11, ms_per_step=round(elapsed / s["frames"] * 1e3, 3), higher_is_better=True, scaling="strong",
        vs_baseline=None, dtype="f16x3 convolutions and memory-read affinity (fp16 hi+lo split operands, 3 fp16 MFMA products per term, fp32 accumulate)",
        data="synthetic", stub_engine=bool(args.stub_engine), **ranks_seen,
        config=dict(workload=f"youtubevos_like_suite (BASELINE config 4): {'all' if len(specs) == 474 else 'first ' + str(len(specs)) + ' of'} 474 synthetic clips, lengths 5*U{{4..36}}, K~U{{1..5}}, 480x853, "
                             f"interact(first frame); clips assigned longest-first to {world} rank(s), no data-path collective; clip generation (GPU) inside the timed region",
                    baseline_config=4, clips=len(specs), parallelism=f"sequence-sharded x{world}", clips_in_flight_per_gpu=args.lanes, suite_checksum=s["checksum"]),
        roofline=None, cpu_baseline=None, wall_seconds=round(elapsed, 3), busiest_rank_engine_seconds=round(s["busiest_rank_seconds"], 3),
        per_rank=sorted(per_rank.values(), key=lambda r: r["rank"]),
        # multi-GPU readiness: the measured per-clip cost model (shard.clip_cost assumes frames * (1 + 0.32 * objects)) and the load
        # imbalance the longest-first assignment of the FULL 474-clip suite would have at 2 / 4 / 8 ranks under either model
        cost_model=dict(measured=ES.fit_cost_model(allrecs), assumed_per_object_ratio=0.32,
                        predicted_imbalance_474_clips=dict(assumed=ES.predicted_imbalance(ES.synthetic_suite(474)),
                                                           measured=(ES.predicted_imbalance(ES.synthetic_suite(474), per_object=ES.fit_cost_model(allrecs)["per_object_ratio"])
                                                                     if ES.fit_cost_model(allrecs) and ES.fit_cost_model(allrecs)["per_object_ratio"] else None))))))


def bench_s2m(args, torch, ops, shard, rank, world, dev):
    """--config s2m: the scribble-to-mask network of an interaction (davis_processor.py:52-70: one DeepLabV3+ forward per object) at
    480x854 (padded to 480x864), `--objects` (default 5) objects as ONE batched forward.  step = one interaction's S2M work."""
    from mivos_amd.model.s2m.s2m_network import deeplabv3plus_resnet50
    from mivos_amd.util import synthetic
    K = args.objects or 5
    net = deeplabv3plus_resnet50()
    net.load_state_dict(synthetic.make_s2m_state(0))
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(5 + rank)
    x = torch.randn(K, 6, 480, 864, generator=g).to(dev)
    x[:, 3:] = (x[:, 3:] > 1.5).float()
    warmup, steps = args.warmup if args.warmup is not None else 3, args.steps if args.steps is not None else 20
    for _ in range(warmup):
        net(x)
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = net(x)
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev)
    if rank != 0:
        return
    # 2 * MACs of DeepLabV3+ / ResNet-50 (output stride 16) per 480x864 sample, from the layer shapes (conv hooks on the oracle)
    print(json.dumps(dict(metric="scribble-to-mask forwards/sec (objects/sec), DAVIS 480p", value=round(world * steps * K / elapsed, 3), unit="objects/s",
                          n_gpus=world, steps=steps, warmup=warmup, ms_per_step=round(elapsed / steps * 1e3, 3), higher_is_better=True, scaling="weak",
                          vs_baseline=None, dtype="f16x3 convolutions (fp16 hi+lo split operands, 3 fp16 MFMA products per term, fp32 accumulate)",
                          data="synthetic", config=dict(workload=f"s2m_interaction: DeepLabV3+/ResNet-50 (6 input channels) on {K} x 480x864 samples in one batched "
                                                                  f"forward = the scribble-to-mask work of one interaction with {K} objects",
                                                        objects=K, height=480, width=864, parallelism=f"replicas x{world}"),
                          roofline=None, cpu_baseline=None, logit_range=[round(float(out.min()), 2), round(float(out.max()), 2)])))


def bench_train(args, torch, ops, shard, rank, world, dev, local):
    """--config train: FusionModel.do_pass iterations (model/fusion_model.py:54-131) on a synthetic batch of 384 x 384 crops
    (dataset/fusion_dataset.py:61), `--objects` samples per rank (default 4), data parallel: each rank its own batch, the flat
    gradient all-reduced (RCCL) every step.  step = one iteration (attention maps + forward + loss + backward + all-reduce + Adam)."""
    from mivos_amd.model.fusion_model import FusionModel
    from mivos_amd.util import synthetic
    B = args.objects or 4
    torch.set_grad_enabled(False)
    model = FusionModel(dict(lr=1e-4, steps=[20000], gamma=0.1, iterations=30000), local_rank=local, world_size=world, distributed=world > 1)
    sd = synthetic.make_prop_state(0)
    model.net.load_state_dict(synthetic.make_fuse_state(0))
    model.prop_net.load_state_dict({k: v for k, v in sd.items() if not k.startswith("decoder.")}, strict=False)
    data = synthetic.synthetic_fusion_batch(B, 384, 384, seed=rank, device=dev)
    warmup, steps = args.warmup if args.warmup is not None else 2, args.steps if args.steps is not None else 10
    it0 = 10000                                                          # inside BootstrappedCE's warm-up window (top-p selection active)
    losses = []
    for i in range(warmup):
        losses.append(float(model.do_pass(dict(data), it0 + i)["losses"]["total_loss"]))
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = model.do_pass(dict(data), it0 + warmup + i)
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev)
    losses.append(float(out["losses"]["total_loss"]))
    recs = shard.gather_records([dict(rank=rank, param_checksum=float(model.flat.double().sum()), first_loss=losses[0], last_loss=losses[-1])])
    if rank != 0:
        return
    print(json.dumps(dict(metric="FusionNet training iterations/sec", value=round(steps / elapsed, 3), unit="iterations/s", n_gpus=world, steps=steps,
                          warmup=warmup, ms_per_step=round(elapsed / steps * 1e3, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
                          dtype="f16x3 forward / data gradients, exact fp32 MFMA weight gradients, fp32 loss and Adam", data="synthetic",
                          config=dict(workload=f"fusion_training (model/fusion_model.py:54-131): {B} samples of 384x384 per rank (2 FusionNet calls each), "
                                               f"BootstrappedCE inside its warm-up, Adam, one all-reduce of 39 905 gradients per step",
                                      batch_per_rank=B, global_batch=B * world, parallelism=f"data-parallel x{world}"),
                          roofline=None, cpu_baseline=None, per_rank=recs,
                          replicas_in_sync=len({r["param_checksum"] for r in recs}) == 1)))


def bench_generator(args, torch, shard, ES, specs, prop, rank, world, dev):
    """--config 4 --generator: the second caller of the network API as a batch workload."""
    from mivos_amd.generation.fusion_generator import FusionGenerator
    from mivos_amd.util import synthetic

    def factory(spec):
        images, gt = synthetic.synthetic_clip_device(spec.frames, spec.height, spec.width, spec.objects, seed=spec.seed, device=dev)
        return FusionGenerator(prop, images, args.mem_freq), gt[:, 1:]

    ES.run_generator_suite([ES.ClipSpec(-1, 10, 2, 480, 853, 7)], factory, 0, 1, sync=torch.cuda.synchronize)       # warm-up clip (untimed)
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = ES.run_generator_suite(specs, factory, rank, world, sync=torch.cuda.synchronize)
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev)
    allrecs = shard.gather_records(recs)
    if rank != 0:
        return
    s = ES.summarize(allrecs, len(specs))
    print(json.dumps(dict(
        metric="propagated frames/sec, fusion-data generator over a YouTube-VOS-like suite sharded over the GPUs", value=round(s["frames"] / elapsed, 3),
        unit="frames/s", n_gpus=world, steps=s["frames"], warmup=18, ms_per_step=round(elapsed / s["frames"] * 1e3, 3), higher_is_better=True,
        scaling="strong", vs_baseline=None,
        dtype="f16x3 convolutions and memory-read affinity (fp16 hi+lo split operands, 3 fp16 MFMA products per term, fp32 accumulate)", data="synthetic",
        config=dict(workload=f"fusion_data_generator (generate_fusion.py:68-120 on BASELINE config 4's suite): first {len(specs)} of 474 synthetic clips, every "
                             f"5th frame a reference frame propagated to both ends of its clip, no fusion; query features cached per clip; clips assigned "
                             f"longest-first to {world} rank(s), no data-path collective; clip generation + uint8 egress inside the timed region",
                    baseline_config=4, clips=len(specs), reference_frames=sum(r["reference_frames"] for r in allrecs),
                    parallelism=f"sequence-sharded x{world}", suite_checksum=s["checksum"]),
        roofline=None, cpu_baseline=None, wall_seconds=round(elapsed, 3), busiest_rank_engine_seconds=round(s["busiest_rank_seconds"], 3))))


if __name__ == "__main__":
    try:
        main()
    finally:
        try:
            import torch.distributed as _dist
            if _dist.is_available() and _dist.is_initialized():
                _dist.destroy_process_group()
        except Exception:
            pass
