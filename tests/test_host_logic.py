"""CPU-only checks of the host side: schedule planner vs the reference's golden trace, state_dict
compatibility, BN folding / weight packing, C-ABI symbol export, loud failure without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mivos_amd import _lib
from mivos_amd.inference_core import InferenceCore, plan_pass
from mivos_amd.model.fusion_net import FusionNet
from mivos_amd.model.propagation.prop_net import PropagationNetwork
from mivos_amd.ops import ConvLayer
from mivos_amd.util.tensor_util import pad_divide_by, unpad, compute_np_iou

torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def simulate_trace(t, k, mem_freq, interactions):
    """Token stream of InferenceCore.interact in SURVEY.md §3 notation, from plan_pass alone."""
    trace, interacted, cached, n_certain = [], set(), set(), 0
    for idx in interactions:
        interacted.add(idx)
        trace.append("M")
        n_certain += 1
        for fwd in (True, False):
            closest, total, steps = plan_pass(t, interacted, idx, fwd, mem_freq, n_certain)
            for st in steps:
                if st.ti not in cached:
                    cached.add(st.ti)
                    trace.append("Q")
                trace.append(f"S{st.n_read}")
                assert st.n_read <= total and (st.slot is None or st.slot < total)
                if st.slot is not None:
                    trace.append("M")
                if st.fuse:
                    nc, nr = abs(closest - st.ti) / abs(closest - idx), abs(idx - st.ti) / abs(closest - idx)
                    trace += [f"F({nc:.2f},{nr:.2f})"] * k
    return trace


def test_schedule_matches_reference_golden_trace(golden_dir):
    with np.load(os.path.join(golden_dir, "e2e_small.npz")) as z:
        c, golden = json.loads(str(z["config"])), str(z["trace"])
    assert " ".join(simulate_trace(c["t"], c["k"], c["mem_freq"], c["interactions"])) == golden


def test_schedule_survey_example():
    # SURVEY.md §3 golden control-flow trace: T=13, K=2, mem_freq=5, interactions 0, 12, 6
    tr = " ".join(simulate_trace(13, 2, 5, [0, 12, 6]))
    first = "M Q S1 M " + "Q S2 M " * 5 + "Q S3 M " * 5 + "Q S4"
    assert tr.startswith(first)
    second = tr[len(first) + 1:]
    assert second.startswith("M S2 M F(0.92,0.08) F(0.92,0.08)")
    assert "F(0.50,0.50)" in second and tr.count("Q") == 12


def test_schedule_edge_cases():
    # interacting on the last frame: nothing to do forward
    closest, total, steps = plan_pass(5, {4}, 4, True, 5, 1)
    assert steps == [] and closest == 5
    # adjacent interacted frames: no frame in between
    assert plan_pass(5, {2, 3}, 2, True, 5, 2)[2] == []
    # mem_freq 1 keeps every frame
    _, total, steps = plan_pass(6, {0}, 0, True, 1, 1)
    assert [s.n_read for s in steps] == [1, 2, 3, 4, 5] and total == 7   # (5 // 1) + 1 + 1, the reference's bank size formula


def test_state_dict_layout_matches_reference(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    assert {k: list(v.shape) for k, v in PropagationNetwork().state_dict().items()} == keys["prop"]
    assert {k: list(v.shape) for k, v in FusionNet().state_dict().items()} == keys["fuse"]


def test_load_state_dict_strict(synthetic_states):
    sd, fsd = synthetic_states
    p, f = PropagationNetwork(top_k=20), FusionNet()
    p.load_state_dict(sd, strict=True)
    f.load_state_dict(fsd, strict=True)
    assert torch.equal(p.decoder.pred.weight, sd["decoder.pred.weight"])


def test_pack_folds_batchnorm(synthetic_states):
    """ConvLayer.pack: conv*scale + bias == BN(conv + b) of the oracle, OHWI weights, channel padding."""
    sd = synthetic_states[0]
    p = PropagationNetwork()
    p.load_state_dict(sd)
    blk = p.mask_rgb_encoder.layer2[0]
    L = blk.conv2.pack(blk.bn2)
    x = torch.randn(1, 128, 10, 12)
    ref = F.batch_norm(F.conv2d(x, sd["mask_rgb_encoder.layer2.0.conv2.weight"], sd["mask_rgb_encoder.layer2.0.conv2.bias"], stride=2, padding=1),
                       sd["mask_rgb_encoder.layer2.0.bn2.running_mean"], sd["mask_rgb_encoder.layer2.0.bn2.running_var"],
                       sd["mask_rgb_encoder.layer2.0.bn2.weight"], sd["mask_rgb_encoder.layer2.0.bn2.bias"], False, 0., 1e-5)
    w = L.w.permute(0, 3, 1, 2)   # OHWI -> OIHW
    got = F.conv2d(x, w, None, stride=2, padding=1) * L.scale[None, :, None, None] + L.bias[None, :, None, None]
    assert float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    stem = p.mask_rgb_encoder.conv1.pack(p.mask_rgb_encoder.bn1, cin_pad=8)
    assert stem.w.shape == (64, 7, 7, 8) and float(stem.w[..., 5:].abs().max()) == 0.0
    kv = p.kv_m_f16.compile()
    assert kv.cout == 640 and kv.split == 128 and kv.w.shape == (640, 3, 3, 1024)


def test_c_abi_exports_every_declared_symbol():
    """libmivos_hip.so loads and exports every function include/mivos_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "mivos_hip.h")).read()
    declared = set(re.findall(r"\b(mivos_[a-z0-9_]+)\s*\(", hdr)) - {"mivos_status"}
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().mivos_version() == 1


def test_lds_dma_conv_host_side_geometry_and_tile_selection():
    """Host logic of the precision-2 (LDS-DMA) convolution path that needs no GPU: Act geometry (SH32 tensors inside a
    one-pixel zero border), packed-weight size, tile selection exported through the C ABI."""
    from mivos_amd import ops
    lib = _lib.load()
    n, h, w, c = 2, 5, 7, 64
    a = ops.Act(torch.zeros(n, h + 2, w + 2, c), n, h, w, c)
    assert a.shape == (n, h, w, c)
    assert a.strides() == ((h + 2) * (w + 2) * c, (w + 2) * c, c)                         # image, row, pixel (floats)
    assert a.interior_ptr() - a.buf.data_ptr() == 4 * ((w + 2) * c + c)                   # pixel (0, 0) behind the border
    sub = a[1:2]
    assert sub.shape == (1, h, w, c) and sub.interior_ptr() - a.interior_ptr() == 4 * a.strides()[0]
    with pytest.raises(_lib.MivosHipError):
        ops.alloc_act(1, 4, 4, 48, torch.device("cpu"))                                    # 32-channel lines only
    # weights: 128 zero bytes + one 128-byte hi|lo line per (K step of 32 channels x 1 tap, output channel)
    assert lib.mivos_pack_weights_f16x3_dma_bytes(256, 3, 3, 256) == 128 + (256 // 32) * 9 * 256 * 128
    assert lib.mivos_pack_weights_f16x3_dma_bytes(96, 1, 1, 64) == 128 + 2 * 96 * 128
    # tiles: 128x64 for narrow layers, 128x256 only for long-K layers whose tiles fill whole rounds of 256 CUs, else 128x128
    pp = lib.mivos_conv2d_variant_pp
    assert pp(129600, 64, 18) == 22 and pp(129600, 32, 9) == 22
    assert pp(129600, 256, 72) == 21                                                      # decoder 3x3: 1013 tiles = 3.96 rounds
    assert pp(129600, 256, 2) == 20                                                       # 1x1 64->256: short K, two workgroups per CU
    assert pp(8100, 512, 288) == 20                                                       # 128 tiles of 128x256 would fill half the chip
    assert pp(32400, 200, 144) == 20                                                      # Cout not a multiple of 256
    # a DMA-staged operand cannot be modified on load / SH32 outputs need the LDS-DMA kernels: refused before any launch
    L = ops.ConvLayer.pack(torch.randn(64, 64, 3, 3), None, None, 1, 1)
    with pytest.raises(_lib.MivosHipError):
        ops.conv(a, L, relu_in=True)
    with pytest.raises(_lib.MivosHipError):
        ops.conv(torch.zeros(1, 5, 7, 64), L, out_act=True)


def test_sh32_restatement_round_trip_and_layout():
    """oracle/sh32.py (the CPU definition the HIP pack kernels are compared with on the GPU): hi + lo keeps 22 bits, the
    border stays zero, a 128-byte line = 32 hi halves then 32 lo halves, weight lines are chunk-swizzled by (n >> 1) & 7."""
    from oracle import sh32
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 5, 64, generator=g) * 7.0
    buf = sh32.pack_activation(x)
    assert buf.shape == (2, 5, 7, 64) and buf.dtype == torch.float32
    assert float((sh32.unpack_activation(buf) - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())
    edge = buf.clone()
    edge[:, 1:-1, 1:-1] = 0
    assert int(edge.view(torch.int32).abs().max()) == 0
    line = buf[1, 2, 3].view(torch.float16)                     # pixel (1, 2) of image 1: 2 groups x (32 hi | 32 lo)
    hi, lo = sh32.split_hi_lo(x[1, 1, 2])
    assert torch.equal(line[0:32], hi[0:32]) and torch.equal(line[32:64], lo[0:32]) and torch.equal(line[64:96], hi[32:64])
    assert torch.equal(sh32.pack_activation(x, relu=True), sh32.pack_activation(torch.relu(x)))
    w = torch.randn(5, 1, 1, 32, generator=g)
    packed = sh32.pack_weights_dma(w, 2.0).view(torch.float16)
    assert int(packed[:64].view(torch.int16).abs().max()) == 0  # the zero line
    whi, wlo = sh32.split_hi_lo(w * 2.0)
    for n in range(5):
        row = packed[64 + n * 64: 64 + (n + 1) * 64].view(8, 8)
        for lc in range(8):
            src = (wlo if lc >= 4 else whi)[n, 0, 0, (lc & 3) * 8:(lc & 3) * 8 + 8]
            assert torch.equal(row[lc ^ ((n >> 1) & 7)], src)


def test_cpu_tensors_fail_loudly(synthetic_states):
    """No CPU fallback: the product path refuses to run without an MI355X."""
    p = PropagationNetwork()
    with pytest.raises(_lib.MivosHipError):
        p.get_query_values(torch.zeros(1, 3, 64, 64))
    with pytest.raises(_lib.MivosHipError):
        FusionNet()(torch.zeros(1, 3, 16, 16), torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16), torch.zeros(1, 2, 16, 16), torch.zeros(1, 2))
    with pytest.raises(_lib.MivosHipError):
        InferenceCore(p, FusionNet(), torch.zeros(1, 2, 3, 32, 32), 1, device="cpu")


def test_tensor_util_matches_oracle():
    from oracle import stm_oracle as O
    for shape in [(1, 3, 480, 854), (2, 1, 120, 150), (1, 1, 101, 35), (1, 1, 64, 64)]:
        x = torch.randn(*shape)
        a, pa = pad_divide_by(x, 16)
        b, pb = O.pad_divide_by(x, 16)
        assert pa == pb and torch.equal(a, b)
        assert torch.equal(unpad(a, pa), x)
    s, g = np.zeros((4, 4), bool), np.zeros((4, 4), bool)
    s[:2], g[1:3] = True, True
    assert abs(float(compute_np_iou(s, g)) - 1 / 3) < 1e-5


@pytest.mark.parametrize("n_obj,n_mem,n_q,top_k", [(5, 7 * 1620, 1620, 50), (1, 5 * 1620, 1620, 20), (3, 200 * 8160, 8160, 50),
                                                   (1, 80, 80, 20), (2, 160, 80, 20), (1, 24, 24, 20), (1, 1620, 1620, 50), (5, 15 * 1620, 1620, 50)])
@pytest.mark.parametrize("f16x3", [0, 1])
def test_memory_read_work_partition(n_obj, n_mem, n_q, top_k, f16x3):
    """Host logic of the persistent memory-read kernel (csrc/memory_read.hip::make_plan): replay the kernel's segment
    loop for every workgroup and check that the tiles of every stream are covered exactly once, that the segment -> list
    slot mapping is a bijection onto 0..n-1 per stream and fits the allocated slots, and the finalize kernel's
    (w_first, w_last) formula names the same workgroups."""
    import ctypes as C
    lib = _lib.load()
    out = (C.c_int32 * 7)()
    assert lib.mivos_memory_read_plan(n_obj, n_mem, n_q, top_k, f16x3, out) == 0
    n_wg, per_wg, slots, tps, streams, L, qt = list(out)
    assert qt == (256 if f16x3 and n_mem >= 200000 else 64)         # 8 waves x 32 queries per workgroup for long memories (fp16 kernel)
    assert streams == n_obj * -(-n_q // qt) and tps == -(-n_mem // 32) and L == top_k + 16
    total = streams * tps
    assert 1 <= n_wg <= 256 and (n_wg - 1) * per_wg < total <= n_wg * per_wg and slots <= 12
    covered = {s: [] for s in range(streams)}
    used = {s: set() for s in range(streams)}
    for w in range(n_wg):
        t, end = w * per_wg, min((w + 1) * per_wg, total)
        while t < end:
            s = t // tps
            lo = t - s * tps
            hi = min(tps, lo + end - t)
            slot = w - (s * tps) // per_wg
            assert 0 <= slot < slots and slot not in used[s]
            used[s].add(slot)
            covered[s].append((lo, hi))
            t += hi - lo
    for s in range(streams):
        segs = sorted(covered[s])
        assert segs[0][0] == 0 and segs[-1][1] == tps and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
        w_first, w_last = (s * tps) // per_wg, ((s + 1) * tps - 1) // per_wg
        assert used[s] == set(range(w_last - w_first + 1))
    assert lib.mivos_memory_read_workspace_bytes(n_obj, n_mem, n_q, top_k) >= 64 + streams * slots * qt * L * 8       # 64-byte plan header


def test_activation_batches_are_capped_by_bytes():
    """the LDS-DMA kernels address one SH32 tensor with 32-bit offsets (< 2 GB); object / query batches are chunked
    by ops.max_act_batch (the largest bordered activation of the trunks / decoder is 256 channels at 1/4 resolution)."""
    from mivos_amd import ops
    per_image = lambda h, w: (h // 4 + 2) * (w // 4 + 2) * 256 * 4
    for h, w in [(480, 864), (1088, 1920), (2160, 3840)]:
        cap = ops.max_act_batch(h // 4, w // 4, 256)
        assert cap >= 1 and cap * per_image(h, w) <= ops.ACT_BYTES_LIMIT < (cap + 1) * per_image(h, w)
    assert ops.max_act_batch(120, 216, 256) == 78 and ops.max_act_batch(272, 480, 256) == 15 and ops.max_act_batch(540, 960, 256) == 4
    assert ops.max_act_batch(100000, 100000, 256) == 1            # never zero: a single image is attempted (and refused by the library)


def test_split_key_rows_layout_and_f16x3_affinity_error():
    """oracle/sh32.py's definition of the pre-split key rows (compared bitwise with mivos_memory_split_keys on the GPU) and
    the size of the f16x3 affinity's deviation from the exact product: a few 1e-7 of the score scale, below what fp32
    accumulation order already moves."""
    from oracle import sh32
    g = torch.Generator().manual_seed(3)
    keys = torch.randn(50, 128, generator=g) * 3
    rows = sh32.split_key_rows(keys).view(torch.float16).view(50, 4, 4, 2, 8)          # [row, b, ks, part, e]
    hi, lo = sh32.split_hi_lo(keys)
    for b, ks in ((0, 0), (3, 1), (1, 3)):
        c0 = 32 * ks + 8 * b
        assert torch.equal(rows[:, b, ks, 0], hi[:, c0:c0 + 8]) and torch.equal(rows[:, b, ks, 1], lo[:, c0:c0 + 8])
    qk = torch.randn(40, 128, generator=g) * 3
    exact = keys.double() @ (qk.double() / (128 ** 0.5)).t()
    err16 = float((sh32.affinity_f16x3(keys, qk) - exact).abs().max())
    err32 = float(((keys @ (qk / (128 ** 0.5)).t()).double() - exact).abs().max())
    assert err16 < 2e-5 and err16 < 4 * err32 + 1e-6, (err16, err32)


def test_plan_cache_staleness_rules():
    """model/plan_cache.py: what drops a network's packed weights (CPU-side bookkeeping only, no kernels)."""
    import torch
    from mivos_amd.model.fusion_net import FusionNet
    from mivos_amd.model.s2m.s2m_network import S2M
    for net in (FusionNet(), S2M()):
        p = next(net.parameters())
        net._plan = "packed"; net._stamp_plan()
        net.refresh_plan_if_stale()
        assert net._plan == "packed"                      # nothing changed
        net.to("cpu")
        assert net._plan == "packed"                      # a no-op move keeps every storage
        with torch.no_grad():
            p.mul_(1.5)                                   # what an optimiser step does: autograd's version counter moves
        net.refresh_plan_if_stale()
        assert net._plan is None
        net._plan = "packed"; net._stamp_plan()
        p.data.copy_(torch.zeros_like(p))                 # invisible by design (documented): needs invalidate_plan()
        net.refresh_plan_if_stale()
        assert net._plan == "packed"
        net.invalidate_plan()
        assert net._plan is None
        net._plan = "packed"; net._stamp_plan()
        net.load_state_dict(net.state_dict())
        assert net._plan is None
        net._plan = "packed"; net._stamp_plan()
        net.double()                                      # dtype move: new storages
        assert net._plan is None


def test_counting_cut_model():
    """csrc/memory_read.hip::count_kth, the cut of a candidate compaction, restated in Python and checked against a sort: the cut
    keeps between k and k + slack entries and they are the largest ones, for pools of both signs, narrow bands, heavy and
    total ties (which must fall through to the index bisection) - the invariants the kernel's exact top-k tests rely on.
    The model mirrors the kernel step by step: score words -> 64 power-of-two buckets over [lo, max] -> suffix counts -> the
    bucket where the count reaches k -> 6 more bits per level -> index bisection among equal score words."""
    import numpy as np

    def f2ord(f):
        u = int(np.float32(f).view(np.uint32))
        return (~u & 0xffffffff) if (u & 0x80000000) else (u | 0x80000000)

    def count_kth(keys, k, slack, lo):
        h = [x >> 32 for x in keys]
        rng_ = max(h) - lo
        shift = max(0, 26 - (32 - rng_.bit_length())) if rng_ else 0
        width_m1, above, levels = 0xffffffff, 0, 0
        while True:
            levels += 1
            hist = [0] * 64
            for x in h:
                d = (x - lo) & 0xffffffff
                if x >= lo and d <= width_m1:
                    assert (d >> shift) < 64
                    hist[63 - (d >> shift)] += 1
            pre = np.cumsum(hist)
            need = k - above
            assert need >= 1 and pre[-1] >= need                       # the range always holds the cut
            lane = int(np.argmax(pre >= need))
            total = above + int(pre[lane])
            lo = lo + ((63 - lane) << shift)
            if total <= k + slack or shift == 0:
                break
            above += int(pre[lane - 1]) if lane else 0
            width_m1 = (1 << shift) - 1
            shift = shift - 6 if shift > 6 else 0
        prefix = lo << 32
        if total > k + slack:                                          # equal score words straddle the cut
            for b in range(31, -1, -1):
                trial = prefix | (1 << b)
                c = sum(1 for x in keys if x >= trial)
                if c >= k:
                    prefix, total = trial, c
                    if c <= k + slack:
                        break
        return prefix, total, levels

    rs = np.random.RandomState(0)
    levels = []
    for trial in range(600):
        n = int(rs.randint(67, 245))
        mode = trial % 6
        if mode == 0:
            s = rs.randn(n).astype(np.float32) * 3
        elif mode == 1:
            s = (rs.randn(n) * 5 + 20).astype(np.float32)
        elif mode == 2:
            s = rs.choice(np.float32([1.5, 2.5, -1.0, 0.0]), n)
        elif mode == 3:
            s = np.full(n, np.float32(7.25))
        elif mode == 4:
            s = (rs.randn(n) * 1e-3 + 40).astype(np.float32)
        else:
            s = np.concatenate([rs.randn(n - 30).astype(np.float32), np.float32(rs.choice([0.5, 0.75], 30))])
        idx = rs.permutation(100000)[:n]
        keys = [(f2ord(x) << 32) | (0xffffffff - int(i)) for x, i in zip(s, idx)]
        k, slack = int(rs.choice([20, 50, 64])), 16
        if n <= k + slack:
            continue
        lo = f2ord(np.float32(-np.inf)) if trial % 2 else min(x >> 32 for x in keys)      # first compaction / a converged threshold
        p, c, lv = count_kth(keys, k, slack, lo)
        levels.append(lv)
        assert k <= c <= k + slack and c == sum(1 for x in keys if x >= p)
        assert set(x for x in keys if x >= p) == set(sorted(keys, reverse=True)[:c])
    assert np.median(levels) <= 2 and max(levels) <= 6


def test_bench_roofline_helpers_on_committed_profiles():
    """bench.py's host-side helpers, which only ever run on the GPU box otherwise: the PMC lookups are keyed by (config, kernel
    instantiation) and must name the committed pass they came from (a traffic / utilisation figure of another config or kernel
    variant is worse than none: VERDICT r2 weak #4), and kernel_rooflines() turns HIP-event samples into the `roofline` block of
    the JSON line."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # committed records are used only for the kernels they were read from (`_meta.csrc_fingerprint`, scripts/csrc_fingerprint.py): work
    # on copies stamped with this tree's fingerprint, then on an unstamped copy
    import glob
    import json
    import shutil
    import tempfile
    src = bench.PROFILES_DIR
    tmp = tempfile.mkdtemp(prefix="mivos_profiles_")
    try:
        for f in glob.glob(os.path.join(src, "r04*config*pmc_traffic.json")) + glob.glob(os.path.join(src, "r04*config*mfma_util.json")) + \
                glob.glob(os.path.join(src, "r03h*config5*.json")) + glob.glob(os.path.join(src, "r03f*config5*.json")):
            table = json.load(open(f))
            table["_meta"] = dict(csrc_fingerprint=bench._csrc_fingerprint())
            json.dump(table, open(os.path.join(tmp, os.path.basename(f)), "w"))
        bench.PROFILES_DIR = tmp
        t = bench.pmc_traffic(3, "conv_f16x3_pp_kernel<128,128,2,4,0>")
        assert t and "config3" in t["source"] and t["bytes_per_launch"] == t["read"] + t["write"] and t["launches_profiled"] > 1000
        assert bench.pmc_traffic(3, "conv_f16x3_pp_kernel<128,128,2,4,9>") is None             # another instantiation: no figure
        assert bench.pmc_traffic(2, "memread_select256_kernel") is None                        # no pass of that kernel in that config
        t5 = bench.pmc_traffic(5, "memread_select256_kernel")
        assert t5 and "config5" in t5["source"] and "note" in t5                               # a single-read pass says so
        u = bench.pmc_mfma_util(3, "conv_f16x3_pp_kernel<128,256,2,4,0>")
        assert u and 40 < u["mean_pct"] < 90 and u["min_pct"] <= u["mean_pct"] <= u["max_pct"]
        assert bench.pmc_mfma_util(5, "memread_select256_kernel")["dispatches"] >= 1
        assert bench.pmc_mfma_util(4, "memread_select256_kernel") is None
        # a record read from other kernels (no or another fingerprint) answers with a marker, never with numbers
        name = os.path.basename(t["source"])
        table = json.load(open(os.path.join(tmp, name)))
        table["_meta"] = dict(csrc_fingerprint="0" * 16)
        json.dump(table, open(os.path.join(tmp, name), "w"))
        stale = bench.pmc_traffic(3, "conv_f16x3_pp_kernel<128,128,2,4,0>")
        assert stale["stale"] is True and stale["source"] == name and "bytes_per_launch" not in stale and stale["tree"] == bench._csrc_fingerprint()
        table["_meta"] = dict(csrc_fingerprint=bench._csrc_fingerprint())
        json.dump(table, open(os.path.join(tmp, name), "w"))

        class Ev:
            def __init__(self, t):
                self.t = t

            def elapsed_time(self, other):
                return other.t - self.t                                                       # milliseconds, like torch.cuda.Event

        samples = [(20, 12e9, Ev(0.0), Ev(0.05), (8100, 256, 256, 3, 1, 0)), (21, 50e9, Ev(0.0), Ev(0.25), (129600, 256, 256, 3, 1, 1)),
                   (90, 28e9, Ev(0.0), Ev(0.32), (5, 11340, 1620, 50, 36e6)), (91, 0.0, Ev(0.0), Ev(0.08), (5, 1620, 50, 8e8))]
        roof, aff, table = bench.kernel_rooflines(samples, 0.0, 3, "memread_select_kernel<0,false,true>")
        assert roof["kernel"] == "conv_f16x3_pp_kernel<128,256,2,4,0>" and abs(roof["achieved"] - 200.0) < 1e-6 and abs(roof["frac"] - 200.0 / 833.3) < 1e-3
        assert roof["traffic"]["source"].endswith("pmc_traffic.json") and roof["mfma_util_pmc"]["source"].endswith("mfma_util.json")
        assert abs(aff["achieved"] - 87.5) < 1e-6 and abs(aff["frac"] - 87.5 / 833.3) < 1e-3 and "frac_of_f32_mfma_peak" not in aff and aff["finalize"]["avg_launch_us"] == 80.0
        assert set(table) == {"conv_f16x3_pp_kernel<128,128,2,4,0>", "conv_f16x3_pp_kernel<128,256,2,4,0>", "memread_select_kernel", "memread_finalize_kernel"}
        assert bench.kernel_rooflines([], 0.0, 3, None) == (None, None, {})
    finally:
        bench.PROFILES_DIR = src
        shutil.rmtree(tmp, ignore_errors=True)


def test_hi_first_bound_model():
    """The bound the hi-first select kernel (csrc/memory_read.hip, memread_select256_kernel<true>) skips key tiles by: with
    x = hi + lo the fp16 split of the engine, |sum (kh ql + kl qh)| <= 1.25 x 2^-10 |k| |q| + 1e-6 (|k| + |q|), over magnitudes
    from 1e-6 (fp16 subnormals) to 1e2 and sign-aligned vectors (the worst case of the Cauchy-Schwarz step)."""
    import numpy as np

    def split(x):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float64), lo.astype(np.float64)

    rs = np.random.RandomState(1)
    worst = 0.0
    for trial in range(4000):
        k = (rs.randn(128) * [1, 3, 1e-3, 100, 1e-6][trial % 5]).astype(np.float32)
        q = (rs.randn(128) * [1, 3, 10, 1e-2, 1e3][(trial // 5) % 5]).astype(np.float32)
        if trial % 7 == 0:
            k, q = np.abs(k), np.abs(q)
        kh, kl = split(k)
        qh, ql = split(q)
        dropped = abs(np.sum(kl * qh) + np.sum(kh * ql))
        kn, qn = np.sqrt(np.sum((kh + kl) ** 2)), np.sqrt(np.sum((qh + ql) ** 2))
        eps = kn * qn * (1.25 / 1024) + 1e-6 * (kn + qn)
        assert dropped <= eps
        worst = max(worst, dropped / eps)
    assert worst < 0.5                                   # (and the fp32 accumulation of 128 terms, <= 1e-5 |k| |q|, fits in the margin)


def test_fusion_checkpoint_layout_is_torch_optim_compatible():
    """FusionModel's checkpoint halves (model/fusion_model.py: adam_state_dict / multistep_state_dict / flat_from_adam_state_dict) against
    REAL torch.optim.Adam + MultiStepLR objects, the classes the reference's FusionModel saves and restores (fusion_model.py:152-175):
    a torch-written state loads into the flat vectors, the flat vectors load back into torch, and both continue identically."""
    import torch
    from mivos_amd.model.fusion_model import adam_state_dict, flat_from_adam_state_dict, multistep_state_dict
    from mivos_amd.util.synthetic import fuse_spec
    shapes = [tuple(v) for v in fuse_spec().values()]
    g = torch.Generator().manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-7)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [2, 5], 0.1)
    for _ in range(3):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
        sch.step()
    m, v, step = flat_from_adam_state_dict(opt.state_dict(), shapes, "cpu")
    assert step == 3 and m.numel() == v.numel() == sum(p.numel() for p in params) == 39905
    off = 0
    for i, p in enumerate(params):
        assert torch.equal(m[off:off + p.numel()].view_as(p), opt.state[p]["exp_avg"]) and torch.equal(v[off:off + p.numel()].view_as(p), opt.state[p]["exp_avg_sq"])
        off += p.numel()
    ours = adam_state_dict(shapes, m, v, step, sch.get_last_lr()[0], 1e-4)
    assert set(ours) == {"state", "param_groups"} and ours["param_groups"][0]["params"] == list(range(12)) and ours["param_groups"][0]["lr"] == sch.get_last_lr()[0]
    clones = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt2 = torch.optim.Adam(clones, lr=1e-4, weight_decay=1e-7)
    sch2 = torch.optim.lr_scheduler.MultiStepLR(opt2, [2, 5], 0.1)
    opt2.load_state_dict(ours)
    sch2.load_state_dict(multistep_state_dict([2, 5], 0.1, 1e-4, step))
    assert sch2.get_last_lr() == sch.get_last_lr() and sch2.last_epoch == sch.last_epoch == 3
    for _ in range(3):                                          # across the second milestone
        for p, q in zip(params, clones):
            p.grad = torch.randn(p.shape, generator=g)
            q.grad = p.grad.clone()
        opt.step(); opt2.step(); sch.step(); sch2.step()
    assert all(torch.equal(p, q) for p, q in zip(params, clones)) and sch.get_last_lr() == sch2.get_last_lr()
    fresh = adam_state_dict(shapes, m * 0, v * 0, 0, 1e-4, 1e-4)                      # before the first step: no per-parameter state, like torch
    assert fresh["state"] == {} and flat_from_adam_state_dict(fresh, shapes, "cpu")[2] == 0


def test_xcd_contiguous_dealing_model():
    """csrc/conv_common.h::xcd_remap (restated): hardware block b runs on XCD b % 8; the remap hands XCD x the logical indices of ONE contiguous range, is a
    bijection for every grid size, and is the identity below 16 blocks.  The element- / tile-walking kernels (upsample, maxpool, FusionNet head, the select
    kernel's chunks) use it so that neighbouring outputs, which read the same input lines, meet in one of the eight unshared L2s."""
    def remap(b, n):
        if n < 16:
            return b
        q, r, xcd, pos = n >> 3, n & 7, b & 7, b >> 3
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + pos
    for n in (1, 7, 15, 16, 17, 31, 130, 255, 256, 257, 2048, 8700):
        out = [remap(b, n) for b in range(n)]
        assert sorted(out) == list(range(n)), n
        if n >= 16:
            for x in range(8):
                mine = sorted(remap(b, n) for b in range(x, n, 8))
                assert mine == list(range(mine[0], mine[0] + len(mine))), (n, x)           # one contiguous run per XCD
            starts = [min(remap(b, n) for b in range(x, n, 8)) for x in range(8)]
            assert starts == sorted(starts)                                                  # XCD 0 first ... XCD 7 last


def test_chip_share_context_nests_and_restores():
    """ops.chip_share(n): the launch-geometry hint the lanes drivers set (mivos_conv_desc.chip_share); nests multiplicatively where a two-pass
    interaction runs inside a two-lane suite (InferenceCore._run_passes: PASS_CHIP_SHARE * ops.CHIP_SHARE) and restores on exit, also on errors."""
    from mivos_amd import ops
    assert ops.CHIP_SHARE == 1
    with ops.chip_share(3):
        assert ops.CHIP_SHARE == 3
        with ops.chip_share(2 * ops.CHIP_SHARE):
            assert ops.CHIP_SHARE == 6
        assert ops.CHIP_SHARE == 3
        try:
            with ops.chip_share(0):                      # clamped to 1
                assert ops.CHIP_SHARE == 1
                raise RuntimeError("x")
        except RuntimeError:
            pass
        assert ops.CHIP_SHARE == 3
    assert ops.CHIP_SHARE == 1


def test_csrc_fingerprint_gates_the_committed_pmc_records():
    """scripts/csrc_fingerprint.py is deterministic, and bench.py's lookup of the newest committed config-3 PMC record answers with numbers exactly when
    that record carries this tree's fingerprint - with a `stale` marker otherwise (a kernel edit after the counters were read must not quote them)."""
    import glob
    import importlib.util
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    try:
        from csrc_fingerprint import csrc_fingerprint
    finally:
        sys.path.pop(0)
    fp = csrc_fingerprint(root)
    assert len(fp) == 16 and fp == csrc_fingerprint(root)
    spec = importlib.util.spec_from_file_location("bench_module_fp", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    newest = sorted(glob.glob(os.path.join(root, "profiles", "*config3*pmc_traffic.json")))[-1]
    fresh = json.load(open(newest)).get("_meta", {}).get("csrc_fingerprint") == fp
    got = bench.pmc_traffic(3, "conv_f16x3_pp_kernel<128,128,2,4,0>")
    assert got is not None and (("bytes_per_launch" in got) if fresh else (got.get("stale") is True and "bytes_per_launch" not in got))


def test_condition_state_touches_only_what_it_says():
    """synthetic.condition_state (the closed-loop fixtures' post-hoc gains): gains of 1 return the golden weights bit for bit, each knob changes exactly its tensors,
    and synthetic_clip(texture=0) is the clip every golden vector was made with."""
    import torch
    from mivos_amd.util import synthetic
    sd = synthetic.make_prop_state(0)
    same = synthetic.condition_state(sd)
    assert list(same) == list(sd) and all(torch.equal(same[k], sd[k]) for k in sd)
    c = synthetic.condition_state(sd, **synthetic.CLOSED_LOOP_CONDITIONING)
    changed = sorted(k for k in sd if not torch.equal(c[k], sd[k]))
    assert changed == ["decoder.pred.bias", "decoder.pred.weight", "mask_rgb_encoder.conv1.weight"]
    g, m = synthetic.CLOSED_LOOP_CONDITIONING["logit_gain"], synthetic.CLOSED_LOOP_CONDITIONING["mask_gain"]
    assert torch.equal(c["decoder.pred.weight"], sd["decoder.pred.weight"] * g)
    w0, w1 = sd["mask_rgb_encoder.conv1.weight"], c["mask_rgb_encoder.conv1.weight"]
    assert torch.equal(w1[:, :3], w0[:, :3]) and torch.equal(w1[:, 3:], w0[:, 3:] * m)
    k = synthetic.condition_state(sd, key_gain=2.0)
    assert sorted(x for x in sd if not torch.equal(k[x], sd[x])) == ["kv_m_f16.key_proj.bias", "kv_m_f16.key_proj.weight", "kv_q_f16.key_proj.bias", "kv_q_f16.key_proj.weight"]
    a, ga = synthetic.synthetic_clip(3, 48, 64, 2, seed=5)
    b, gb = synthetic.synthetic_clip(3, 48, 64, 2, seed=5, texture=0.0)
    t, _ = synthetic.synthetic_clip(3, 48, 64, 2, seed=5, texture=0.5)
    assert torch.equal(a, b) and torch.equal(ga, gb) and not torch.equal(a, t) and t.shape == a.shape
