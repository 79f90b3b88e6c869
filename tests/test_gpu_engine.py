"""Network-level and end-to-end parity of the MI355X engine with the CPU oracle and with the golden
vectors produced by the unmodified reference (needs an MI355X).  Tolerances are the north star's:
mask IoU >= 0.999 vs the reference path, logits within 1e-3."""
import json
import os

import numpy as np
import pytest
import torch

from mivos_amd.inference_core import InferenceCore
from mivos_amd.model.aggregate import aggregate_wbg
from mivos_amd.model.fusion_net import FusionNet
from mivos_amd.model.propagation.prop_net import PropagationNetwork
from mivos_amd.util.tensor_util import compute_np_iou
from oracle import stm_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda:0"
LOGIT_TOL = 1e-3


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def nets(synthetic_states):
    sd, fsd = synthetic_states
    prop, fuse = PropagationNetwork(top_k=20), FusionNet()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    return prop.to(DEV).eval(), fuse.to(DEV).eval()


@pytest.fixture(scope="module")
def ops_golden(golden_dir):
    with np.load(os.path.join(golden_dir, "ops_small.npz")) as z:
        return {k: z[k] for k in z.files}


def mean_iou(a, b, k):
    return float(np.mean([compute_np_iou(a == j, b == j) for j in range(1, k + 1)]))


# Round 5: factor 2.0 -> 1.5.  On a well-conditioned session the engine sits at 1.06 - 1.14 x the reference's own distance from fp64 (medians over the
# 137 frames of the long config-3 session, profiles/r05e_long_session_parity.json; the exact-fp32 mode 1.08 - 1.18), and of the 83 frames
# in the committed records of the whole suite (profiles/r04g_parity_ratios.jsonl) all but the K = 1 session's two documented tie frames keep >= 50 % slack at 1.5.
ARBITRATION_FACTOR, ARBITRATION_FLOOR = 1.5, 2.5e-4
TIE_MARGIN, TIE_QUANTILE, TIE_CAP = 1e-3, 1e-4, 5e-2
# FROZEN (round 4): the tie clause's quantile is the documented 1e-4.  A test that needs more passes its own
# `tie_quantile` with the justification written at the call site; every record names the clause each frame passed through and
# the quantile in force, and the record of the full suite is committed per round (profiles/r04*_parity_ratios.jsonl,
# scripts/parity_clauses.py lists the frames that did not pass the strict clause).


def fp64_gate(tag, eng_prob, ref32_prob, ref64_prob, margins=None, tie_quantile=TIE_QUANTILE):
    """The closed-loop parity gate, fp64-arbitrated.  Per frame f, with e = |engine - fp64| and r = |reference_fp32 -
    fp64| (the reference's OWN fp32 arithmetic against an fp64 run of the same algorithm):

      strict clause   max e_f <= 1.5 max r_f + 2.5e-4       (2.5e-4 = the probability equivalent of the north star's 1e-3 logit bar)

    The algorithm is discontinuous: a memory position whose rank-k / rank-(k+1) affinities are closer than the fp32 rounding noise of
    the keys (55 convolutions deep: 1e-4 ... 3e-4 on keys of magnitude 10 for the reference's fp32 run and for the engine alike, i.e.
    ~3e-4 on a score) enters the top-k set in one fp32 implementation and not in another, and the pixels behind it move by 1e-3 ...
    2e-2 - in the REFERENCE's fp32 run too (480p K=3: both runs sit 2.19e-2 from fp64 on the same frame).  Which run flips at
    which query is rounding luck, so a maximum over 4e5 pixels cannot be compared run against run when such a pair exists.  For
    those sessions - and only for them: `margins` holds, per propagated frame, the smallest rank-k / k+1 gap of the fp64 run; a frame
    qualifies when that gap is below TIE_MARGIN = 1e-3 somewhere in the session so far (flips travel through the memory bank) -

      tie clause      all but `tie_quantile` (default TIE_QUANTILE = 1e-4) of the frame's pixels obey the strict bound (quantile
                      against quantile) and max e_f <= TIE_CAP = 5e-2 (what a flipped neighbour of weight ~1/k can move).  One
                      flipped neighbour at the 1/16-resolution grid reaches ~32 x 32 output pixels through the decoder = 0.25 % of
                      a 480p frame: a test whose session holds such a flip passes tie_quantile=5e-3 and says so at its call site.

    Prints the numbers, appends them to gpurun_out/parity_ratios.jsonl (a record per run) and returns (passed, record)."""
    e = (eng_prob.cpu().double() - ref64_prob).abs()
    r = (ref32_prob.cpu().double() - ref64_prob).abs()
    T = e.shape[1]
    ef, rf = e.amax(dim=(0, 2, 3, 4)), r.amax(dim=(0, 2, 3, 4))                      # per frame
    live = rf > 0                                                                  # (interacted frames are exact on both sides)
    ratio = float((ef[live] / rf[live]).max()) if bool(live.any()) else 0.0
    strict = ef <= ARBITRATION_FACTOR * rf + ARBITRATION_FLOOR
    clause, eq, rq = [], [], []
    running_min = float("inf")
    order = sorted(margins) if margins else []
    seen = {}
    for t in order:                                                                # frames in index order: a conservative "earlier" relation
        running_min = min(running_min, margins[t])
        seen[t] = running_min
    ok = True
    for t in range(T):
        if bool(strict[t]):
            clause.append("strict"); eq.append(None); rq.append(None)
            continue
        et, rt = e[:, t].reshape(-1), r[:, t].reshape(-1)
        kth = max(1, int(round(et.numel() * (1.0 - tie_quantile))))
        e_q, r_q = float(et.kthvalue(kth).values), float(rt.kthvalue(kth).values)
        eq.append([e_q] + [float(et.kthvalue(max(1, int(round(et.numel() * (1.0 - qq))))).values) for qq in (5e-3, 1e-3, 1e-4)])   # at tie_quantile, 5e-3, 1e-3, 1e-4
        rq.append([r_q] + [float(rt.kthvalue(max(1, int(round(rt.numel() * (1.0 - qq))))).values) for qq in (5e-3, 1e-3, 1e-4)])
        near_tie = margins is not None and min(seen.values(), default=float("inf")) < TIE_MARGIN
        if near_tie and e_q <= ARBITRATION_FACTOR * r_q + ARBITRATION_FLOOR and float(ef[t]) <= TIE_CAP:
            clause.append("tie")
        else:
            clause.append("FAIL"); ok = False
    margin = float((ARBITRATION_FACTOR * rf + ARBITRATION_FLOOR - ef).min())
    rec = dict(test=tag, frames=int(T), engine_vs_fp64_max=float(ef.max()), ref32_vs_fp64_max=float(rf.max()),
               worst_frame_ratio=round(ratio, 3), gate_margin=margin, clauses=clause, tie_quantile=tie_quantile, passed=ok,
               min_topk_margin_fp64=(min(margins.values()) if margins else None),
               frac_gt_1e3_engine=float((e > 1e-3).double().mean()), frac_gt_1e3_ref32=float((r > 1e-3).double().mean()),
               per_frame_engine=[float(x) for x in ef], per_frame_ref32=[float(x) for x in rf],
               per_frame_engine_quantiles=eq, per_frame_ref32_quantiles=rq)
    print(f"{tag}: per-frame max|dprob| engine-fp64 {rec['engine_vs_fp64_max']:.2e} vs reference(fp32)-fp64 {rec['ref32_vs_fp64_max']:.2e}; "
          f"worst per-frame ratio {ratio:.2f} (strict gate: e <= {ARBITRATION_FACTOR} r + {ARBITRATION_FLOOR}, margin {margin:.2e}); clauses {sorted(set(clause))}; "
          f"min fp64 top-k margin {rec['min_topk_margin_fp64']}; frac(|d| > 1e-3) engine {rec['frac_gt_1e3_engine']:.1e} reference {rec['frac_gt_1e3_ref32']:.1e}")
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_ratios.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return ok, rec


def test_query_encoder_and_memorize_golden(nets, ops_golden):
    prop, _ = nets
    g = ops_golden
    q = prop.get_query_values(T(g["en_frame"]).to(DEV))
    for got, name in zip(q, ("en_f16", "en_f8", "en_f4", "en_qk", "en_qv")):
        ref = T(g[name])
        assert got.shape == ref.shape
        assert float((got.cpu() - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max())), name
    k, v = prop.memorize(T(g["en_frame"]).to(DEV), T(g["en_masks"]).to(DEV))
    assert k.shape == g["en_mk"].shape and v.shape == g["en_mv"].shape
    # 55 fp32 layers deep, different summation order than MKL: compare relative to the value range
    assert float((k.cpu() - T(g["en_mk"])).abs().max()) < 3e-5 * float(np.abs(g["en_mk"]).max())
    assert float((v.cpu() - T(g["en_mv"])).abs().max()) < 3e-5 * float(np.abs(g["en_mv"]).max())


def test_decoder_golden(nets, ops_golden):
    """The reference's Decoder.forward (prop_net.py:23-31) on a random m4 [2,1024,4,6] + the golden skip features."""
    prop, _ = nets
    g = ops_golden
    from mivos_amd import ops
    from mivos_amd.model.propagation.modules import run_resblock, run_skip_branch, run_up_branch
    from mivos_amd.model.propagation.prop_net import _nhwc
    dec = prop.plan()["dec"]
    f8, f4 = (_nhwc(T(g[n]).to(DEV)) for n in ("en_f8", "en_f4"))
    if ops.act_path():
        f8, f4 = ops.to_act(f8), ops.to_act(f4)
    x = run_resblock(dec["compress"], _nhwc(T(g["de_m4"]).to(DEV)))
    x = run_up_branch(dec["up_16_8"], run_skip_branch(dec["up_16_8"], f8), x)
    x = run_up_branch(dec["up_8_4"], run_skip_branch(dec["up_8_4"], f4), x)
    lo = ops.conv(x, dec["pred"], relu_in=True)
    got = ops.resize_bilinear(lo.view(2, lo.shape[1], lo.shape[2]), 4 * lo.shape[1], 4 * lo.shape[2]).cpu()
    ref = T(g["de_out"])[:, 0]
    print(f"decoder golden: max|dlogit| {float((got - ref).abs().max()):.2e}, range [{float(ref.min()):.1f}, {float(ref.max()):.1f}]")
    assert got.shape == ref.shape and float((got - ref).abs().max()) < LOGIT_TOL


def test_attention_read_network_golden(golden_dir, synthetic_states):
    """model/attn_network.py:30-80 (training-time twin of get_attention) + dense W + aggregate_wbg_channel vs the
    reference's outputs (tests/golden/attn_small.npz)."""
    from mivos_amd.model.aggregate import aggregate_wbg_channel
    from mivos_amd.model.attn_network import AttentionReadNetwork
    with np.load(os.path.join(golden_dir, "attn_small.npz")) as z:
        g = {k: T(z[k]) for k in z.files}
    net = AttentionReadNetwork()
    net.load_state_dict({k: v for k, v in synthetic_states[0].items() if not k.startswith("decoder.")})
    net = net.to(DEV).eval()
    a1, a2 = net(*(g[n].to(DEV) for n in ("an_image", "an_m11", "an_m21", "an_m12", "an_m22", "an_query")))
    d1, d2 = float((a1.cpu() - g["an_out1"]).abs().max()), float((a2.cpu() - g["an_out2"]).abs().max())
    print(f"AttentionReadNetwork: max|d| {d1:.2e} {d2:.2e} (values up to {float(g['an_out1'].max()):.3f})")
    assert a1.shape == g["an_out1"].shape and d1 < 2e-5 and d2 < 2e-5
    W = net.memory(g["aw_mk"].to(DEV), g["aw_qk"].to(DEV))
    assert W.shape == g["aw_out"].shape and float((W.cpu() - g["aw_out"]).abs().max()) < 1e-6
    assert float((W.sum(1) - 1).abs().max()) < 1e-5
    for hard in (0, 1):
        lg, sm = aggregate_wbg_channel(g["ac_in"].to(DEV), keep_bg=True, hard=bool(hard))
        assert float((lg.cpu() - g[f"ac_logits_{hard}"]).abs().max()) < 2e-5 * 1000 ** hard
        assert float((sm.cpu() - g[f"ac_soft_{hard}"]).abs().max()) < 2e-6
        lg2, sm2 = aggregate_wbg_channel(g["ac_in"].to(DEV), keep_bg=False, hard=bool(hard))
        assert torch.equal(lg2, lg) and torch.equal(sm2, sm[:, 1:])


def test_get_W_matches_reference_semantics(nets):
    """PropagationNetwork.get_W / AttentionMemory.forward (prop_net.py:115-129,183-185): dense softmax over the memory
    positions, and pos @ W equals what get_attention's fused kernel produces."""
    prop, _ = nets
    g = torch.Generator().manual_seed(3)
    mk, qk = torch.randn(2, 128, 1, 7, 9, generator=g) * 2, torch.randn(1, 128, 7, 9, generator=g) * 2
    W = prop.get_W(mk.to(DEV), qk.to(DEV)).cpu()
    ref = torch.softmax(O.affinity(mk, qk), dim=1)
    assert W.shape == (2, 63, 63) and float((W - ref).abs().max()) < 1e-6


def test_fusion_net_golden(nets, ops_golden):
    _, fuse = nets
    g = ops_golden
    out = fuse(*(T(g[k]).to(DEV) for k in ("fu_im", "fu_s1", "fu_s2", "fu_at", "fu_tm")))
    assert out.shape == g["fu_out"].shape
    assert float((out.cpu() - T(g["fu_out"])).abs().max()) < LOGIT_TOL


def test_segment_with_query_public_api_vs_oracle(nets, synthetic_states):
    """Reference-style call sequence on NCHW tensors (what generation/fusion_generator.py does)."""
    prop, _ = nets
    sd = synthetic_states[0]
    images, gt = O.synthetic_clip(3, 128, 160, 2, seed=5)
    f0, f1 = images[:, 0], images[:, 1]
    k0, v0 = prop.memorize(f0.to(DEV), gt[0, 1:].to(DEV))
    q = prop.get_query_values(f1.to(DEV))
    prob = prop.segment_with_query(k0, v0, *q)
    ok, ov = O.memorize(sd, f0, gt[0, 1:])
    oq = O.get_query_values(sd, f1)
    ref_logit = O.segment_logits(sd, ok, ov, *oq, top_k=20)
    assert prob.shape == (2, 1, 128, 160)
    assert float((prob.cpu() - torch.sigmoid(ref_logit)).abs().max()) < 2.5e-4     # d sigmoid <= dlogit / 4
    kr = k0.permute(0, 2, 3, 4, 1).reshape(2, -1, 128)
    vr = v0.permute(0, 2, 3, 4, 1).reshape(2, -1, 512)
    from mivos_amd.model.propagation.prop_net import QueryFeatures, _nhwc
    logit = prop.segment(kr, vr, QueryFeatures(*(_nhwc(t) for t in q)), logits=True)
    assert float((logit.cpu().unsqueeze(1) - ref_logit).abs().max()) < LOGIT_TOL
    out = aggregate_wbg(prob, keep_bg=True)
    assert float((out.cpu() - O.aggregate_wbg(torch.sigmoid(ref_logit), keep_bg=True)).abs().max()) < 2.5e-4


@pytest.mark.parametrize("B,H,W", [(1, 48, 80), (5, 480, 864), (2, 37, 61)])
def test_fusion_net_forward_entry_vs_the_layer_by_layer_path(nets, B, H, W):
    """mivos_fusion_net_forward (one C-ABI call per FusionNet forward): conv1 + two fused residual-block launches +
    the exact-fp32 head, against FusionNet.run(layered=True) = the six convolutions of fusion_net.py:32-50 issued one by one
    (the path the golden vector of the reference pins); on the benchmark's batch (5 objects at 480x864) and on ragged sizes.
    The residual blocks use the same products in the same order (rounding-level agreement); the head is exact fp32 instead
    of f16x3 (~1e-6 of the logit range)."""
    _, fuse = nets
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.zeros(B, H, W, 16)
    x[..., :9] = torch.randn(B, H, W, 9, generator=g)
    x = x.to(DEV)
    one_call = fuse.run(x)
    layered = fuse.run(x, layered=True)
    assert one_call.shape == layered.shape == (B, H, W, 1) and bool(torch.isfinite(one_call).all())
    d = float((one_call - layered).abs().max())
    print(f"FusionNet one call vs layer by layer [{B}x{H}x{W}]: max|dlogit| {d:.2e} on logits up to {float(layered.abs().max()):.2f}")
    assert d < 2e-5 * max(1.0, float(layered.abs().max()))


def test_fusion_generator_call_pattern_vs_oracle(nets, synthetic_states):
    """generation/fusion_generator.py:43-78: the second caller of the network API grows its bank with
    torch.cat on the logical [K,C,T,h,w] tensors and keeps a temporary last-frame entry.  Teacher-forced:
    the oracle reads the engine's bank and previous output, so every step is a single-frame comparison
    (memorize is compared separately at every step)."""
    prop, _ = nets
    sd = synthetic_states[0]
    images, gt = O.synthetic_clip(5, 128, 160, 2, seed=11)
    mem_freq = 2
    m0 = aggregate_wbg(gt[0, 1:].to(DEV), keep_bg=True)
    keys, values = prop.memorize(images[:, 0].to(DEV), m0[1:])
    prev = None
    last_ti = 0
    for ti in range(1, 5):
        this_k, this_v = (keys, values) if prev is None else (torch.cat([keys, prev[0]], 2), torch.cat([values, prev[1]], 2))
        assert this_k.shape == (2, 128, this_k.shape[2], 8, 10) and this_v.shape == (2, 512, this_k.shape[2], 8, 10)
        q = prop.get_query_values(images[:, ti].to(DEV))
        out = aggregate_wbg(prop.segment_with_query(this_k, this_v, *q), keep_bg=True)
        oq = O.get_query_values(sd, images[:, ti])
        O.TOPK_GAP = []
        ref = O.aggregate_wbg(O.segment_with_query(sd, this_k.cpu(), this_v.cpu(), *oq, top_k=20), keep_bg=True)
        margin, O.TOPK_GAP = min(O.TOPK_GAP), None
        d = float((out.cpu() - ref).abs().max())
        print(f"frame {ti}: bank T={this_k.shape[2]}  max|dprob| {d:.2e}  top-k margin {margin:.1e}")
        # the bar is "logits within 1e-3"; the aggregated probability moves by at most half the logit error (softmax over the
        # objects' logits).  A 20th/21st-neighbour tie inside fp32 rounding may legitimately resolve either way
        assert d < (5e-4 if margin > 1e-4 else 2.5e-3)
        assert mean_iou(out.argmax(0).cpu().numpy(), ref.argmax(0).numpy(), 2) >= 0.999
        if ti != 4:
            prev = prop.memorize(images[:, ti].to(DEV), out[1:])
            ok, ov = O.memorize(sd, images[:, ti], out[1:].cpu())
            # 55 fp32 layers, soft (non-binary) masks in: rounding-level agreement relative to the value range
            assert float((prev[0].cpu() - ok).abs().max()) < 1e-4 * float(ok.abs().max())
            assert float((prev[1].cpu() - ov).abs().max()) < 1e-4 * float(ov.abs().max())
            if abs(ti - last_ti) >= mem_freq:
                last_ti = ti
                keys, values = torch.cat([keys, prev[0]], 2), torch.cat([values, prev[1]], 2)
                prev = None


def test_fusion_generator_golden(nets, golden_dir, synthetic_states):
    """The engine's FusionGenerator (reference generation/fusion_generator.py:12-101) on the clip of tests/golden/gen_small.npz - two
    reference frames, the second with range limits inside the clip, the query cache shared between them - against the unmodified
    reference's probabilities (fp32), arbitrated by an fp64 run of the oracle's restatement."""
    from mivos_amd.generation.fusion_generator import FusionGenerator
    prop, _ = nets
    with np.load(os.path.join(golden_dir, "gen_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    c = json.loads(str(g["config"]))
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    gen = FusionGenerator(prop, images.to(DEV), c["mem_freq"])
    g64 = O.OracleGenerator(synthetic_states[0], images, c["mem_freq"], top_k=c["top_k"], dtype=torch.float64, record_margins=True)
    for n, (idx, left, right) in enumerate(c["calls"]):
        gen.reset(c["k"])
        g64.reset(c["k"])
        out = gen.interact_mask(gt[idx, 1:].to(DEV), idx, left, right)
        r64 = g64.interact_mask(gt[idx, 1:], idx, left, right)
        ref = T(g[f"prob_{n}"])
        assert out.shape == ref.shape == (c["k"] + 1, c["t"], c["h"], c["w"])
        assert mean_iou(out.argmax(0).cpu().numpy(), ref.argmax(0).numpy(), c["k"]) >= 0.999
        ok, rec = fp64_gate(f"fusion_generator[{n}]", out.unsqueeze(2), ref.unsqueeze(2), r64.unsqueeze(2), g64.topk_margin)
        assert ok, rec
        if left > 0:
            assert float(out[:, :left].abs().max()) == 0.0                   # frames outside the range keep reset()'s zeros
    assert gen.propagated_frames == 5 + 4 and len(gen.query_buf) == 6        # 6 frames encoded ONCE for both reference frames
    # the two passes from a reference frame advance in turn on two HIP streams (round 5): bit for bit the one-after-the-other order when both
    # run with the same launch geometry (PASS_CHIP_SHARE = 1 here)
    runs = {}
    for conc in (False, True):
        g2 = FusionGenerator(prop, images.to(DEV), c["mem_freq"])
        g2.CONCURRENT_PASSES, g2.PASS_CHIP_SHARE = conc, 1
        outs = []
        for idx, left, right in c["calls"]:
            g2.reset(c["k"])
            outs.append(g2.interact_mask(gt[idx, 1:].to(DEV), idx, left, right).clone())
        runs[conc] = outs
    assert all(torch.equal(a, b) for a, b in zip(runs[False], runs[True]))


def test_end_to_end_golden(nets, golden_dir, synthetic_states):
    """Same 3-interaction session (incl. fusion) the unmodified reference ran to produce e2e_small.npz."""
    prop, fuse = nets
    with np.load(os.path.join(golden_dir, "e2e_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    c = json.loads(str(g["config"]))
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    core = InferenceCore(prop, fuse, images, c["k"], mem_freq=c["mem_freq"], device=DEV)
    o64 = O.OracleCore(synthetic_states[0], synthetic_states[1], images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"], dtype=torch.float64, record_margins=True)
    for n, idx in enumerate(c["interactions"]):
        out = core.interact(gt[idx], idx)
        o64.interact(gt[idx], idx)
        ref = g[f"masks_{n}"]
        assert out.shape == ref.shape and out.dtype == np.uint8
        iou = mean_iou(out, ref, c["k"])
        print(f"interaction {n}: IoU {iou:.6f}  mismatching px {int((out != ref).sum())}")
        assert iou >= 0.999
        # the golden probabilities are the unmodified reference's fp32 run; the fp64 run of the oracle arbitrates
        ok, rec = fp64_gate(f"e2e_golden[{n}]", core.prob, T(g[f"prob_{n}"]), o64.prob, o64.topk_margin)
        assert ok, rec
    assert core.propagated_frames == 6 + 5 + 4


def test_gui_call_pattern_under_autocast_golden(golden_dir):
    """How the reference's GUI drives the processor (interactive_gui.py:550, 616, 626, 636-642, 889-897, 955-960, all inside
    `torch.cuda.amp.autocast`, :990), replayed PyQt-free by oracle/gui_replay.py on the engine: `prob[:, i].clone()` as an edit's start,
    `update_mask_only` after every stroke read back through `np_masks[i]`, in-place `masks[i].zero_()` / `np_masks[i].fill(0)`, the
    `current_mask` alias, progress callbacks, re-interaction after a reset - against what the UNMODIFIED reference InferenceCore produced
    for the same scripted session (tests/golden/gui_small.npz, oracle/make_golden_gui.py)."""
    from oracle import gui_replay as G
    with np.load(os.path.join(golden_dir, "gui_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    c = json.loads(str(g["config"]))
    assert c == G.SESSION and float(g["self_iou_fp32_vs_fp64"].min()) >= 0.9995       # a fixture the reference reproduces itself on
    sd, fsd = G.session_states()
    prop, fuse = PropagationNetwork(top_k=c["top_k"]), FusionNet()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    prop, fuse = prop.to(DEV).eval(), fuse.to(DEV).eval()
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    with torch.cuda.amp.autocast(enabled=True):
        core = InferenceCore(prop, fuse, images, c["k"], mem_profile=0, mem_freq=c["mem_freq"], device=DEV)
        gui, local = G.scripted_session(core, gt)
    assert [n for n, _ in gui.events] == [str(n) for n in g["event_names"]]
    assert [p[1] if p[0] == "total" else -1 for p in gui.progress] == g["progress"].tolist()
    assert np.array_equal(gui.events[0][1], g["current_mask_0"])       # the first event only holds the argmax of a one-hot input: exact
    for i, (name, cur) in enumerate(gui.events):
        ref = g[f"current_mask_{i}"]
        assert cur.shape == ref.shape and cur.dtype == np.uint8
        iou = mean_iou(cur, ref, c["k"])
        print(f"event {i} {name}: IoU vs the reference's GUI state {iou:.6f}, mismatching px {int((cur != ref).sum())}")
        assert iou >= 0.999, (i, name, iou)
    assert core.prob.dtype == torch.float32 and core.masks.dtype == torch.uint8              # autocast changed no public dtype
    assert float((core.prob.cpu() - T(g["final_prob"])).abs().max()) < 5e-3
    assert mean_iou(core.np_masks, g["final_np_masks"], c["k"]) >= 0.999
    assert np.array_equal(local["image"], g["local_image"]) and tuple(local["pad"]) == tuple(int(v) for v in g["local_pad"])
    assert float(np.abs(local["prev_soft_mask"] - g["local_prev_soft_mask"]).max()) < 5e-3
    # the reset reached the processor's own buffers, and the alias the GUI relies on holds: interact returns np_masks itself
    assert gui.current_mask is not None and core.interact(gui.processor.prob[:, 1].clone(), 1) is core.np_masks


def test_update_mask_only_golden(nets, golden_dir):
    """InferenceCore.update_mask_only (reference inference_core.py:273-293; 5 of the 8 interactions of a DAVIS session,
    davis_processor.py:75-82) against the unmodified reference's results (tests/golden/update_small.npz): the argmax over the K+1
    channels of soft probabilities with exact ties (first index wins), of a one-hot mask and of a propagated frame's own
    probabilities, written to masks[idx] (padded) and np_masks[idx] (cropped; 100x141 frame padded unevenly to 112x144) -
    bit-exact on the updated frame; every other frame keeps what the propagation left there."""
    from oracle.make_golden_update import update_inputs
    prop, fuse = nets
    with np.load(os.path.join(golden_dir, "update_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    c = json.loads(str(g["config"]))
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    core = InferenceCore(prop, fuse, images, c["k"], mem_freq=c["mem_freq"], device=DEV)
    assert tuple(core.pad) == tuple(int(v) for v in g["pad"])
    # before any propagation: only the updated frame changes, the rest stays zero (np_masks is allocated at construction, :77-81)
    first = core.update_mask_only(T(g["input_2"]).to(DEV), 4)
    assert first.shape == (c["t"], c["h"], c["w"]) and first.dtype == np.uint8
    assert np.array_equal(first[4], g["np_masks_2"][c["calls"][2]]) and int(first[:4].sum()) == 0
    base = core.interact(gt[0], 0).copy()
    assert mean_iou(base, g["masks_interact"], c["k"]) >= 0.999
    done = set()
    for n, (idx, pm) in enumerate(zip(c["calls"], update_inputs(c, T(g["input_2"])))):
        out = core.update_mask_only(pm if n % 2 else pm.to(DEV), idx)                # host and device arguments
        done.add(idx)
        assert out is core.np_masks and out.dtype == np.uint8 and out.shape == (c["t"], c["h"], c["w"])
        assert np.array_equal(out[idx], g[f"np_masks_{n}"][idx])
        assert np.array_equal(core.masks[idx].cpu().numpy(), g[f"masks_idx_{n}"])
        for t in range(c["t"]):
            if t not in done:
                assert np.array_equal(out[t], base[t])
    assert core.propagated_frames == c["t"] - 1                                      # no propagation happened in between


@pytest.mark.parametrize("K", [1, 3])
def test_480p_single_step_logits_vs_oracle(nets, synthetic_states, K):
    """BASELINE config 2/3 geometry (480x854 -> 480x864, HW = 1620).  One propagation step with identical
    inputs on both sides ("teacher forced"): decoder logits within 1e-3, masks IoU >= 0.999."""
    prop, _ = nets
    sd = synthetic_states[0]
    images, gt = O.synthetic_clip(2, 480, 854, K, seed=30 + K)
    img, pad = O.pad_divide_by(images, 16)
    m0, _ = O.pad_divide_by(gt[0], 16)
    ok, ov = O.memorize(sd, img[:, 0], m0[1:])
    oq = O.get_query_values(sd, img[:, 1])
    O.TOPK_GAP = []
    ref = O.segment_logits(sd, ok, ov, *oq, top_k=20)[:, 0]
    margin, O.TOPK_GAP = min(O.TOPK_GAP), None
    k, v = prop.memorize_into(img[:, 0].to(DEV), m0[1:].to(DEV))
    assert float((k.cpu().permute(0, 3, 1, 2) - ok[:, :, 0]).abs().max()) < 3e-5 * float(ok.abs().max())
    assert float((v.cpu().permute(0, 3, 1, 2) - ov[:, :, 0]).abs().max()) < 3e-5 * float(ov.abs().max())
    q = prop.encode_query(img[:, 1].to(DEV))
    got = prop.segment(k.reshape(K, -1, 128), v.reshape(K, -1, 512), q, logits=True).cpu()
    d = (got - ref).abs()
    print(f"K={K}: max|dlogit| {float(d.max()):.2e}, logit range [{float(ref.min()):.1f}, {float(ref.max()):.1f}], top-k margin {margin:.1e}")
    assert float(d.max()) < LOGIT_TOL
    a = O.aggregate_wbg(torch.sigmoid(got[:, None]), keep_bg=True).argmax(0).numpy()
    b = O.aggregate_wbg(torch.sigmoid(ref[:, None]), keep_bg=True).argmax(0).numpy()
    assert mean_iou(a, b, K) >= 0.999


@pytest.mark.parametrize("K", [1, 3])
def test_480p_propagation_vs_oracle(nets, synthetic_states, K):
    """Closed loop at 480p, top_k = 20 (config 2's setting): 3 propagated frames, then a second interaction that fuses the
    frames in between.  Masks IoU >= 0.999; probabilities gated like the headline test below: per frame the engine may be at
    most 1.5 times as far from an fp64 run of the oracle as the fp32 oracle itself, plus the 1e-3-logit equivalent."""
    prop, fuse = nets
    sd, fsd = synthetic_states
    images, gt = O.synthetic_clip(4, 480, 854, K, seed=20 + K)
    core = InferenceCore(prop, fuse, images, K, mem_freq=2, device=DEV)
    ocore = O.OracleCore(sd, fsd, images, K, mem_freq=2, top_k=20)
    o64 = O.OracleCore(sd, fsd, images, K, mem_freq=2, top_k=20, dtype=torch.float64, record_margins=True)
    for idx in (0, 3):                                               # second one fuses frames 1, 2
        out, ref, _ = core.interact(gt[idx], idx), ocore.interact(gt[idx], idx), o64.interact(gt[idx], idx)
        iou = mean_iou(out, ref, K)
        assert iou >= 0.999
        # K=1 holds the one session of the suite with a top-k flip on the FIRST propagated frame (the fp64 run's rank-20 / 21 gap is
        # 1.7e-4 on frame 1 and 2.0e-5 on frame 2, against ~3e-4 of fp32 score noise): one flipped neighbour moves a ~32 x 32 pixel
        # patch (0.25 % of the frame) by up to 1e-3, so this test - and only this one - runs the tie clause at the 5e-3 quantile.
        # What the flip cannot hide is asserted directly by test_480p_single_step_logits_vs_oracle[1] (|dlogit| < 1e-3 on identical inputs).
        ok, rec = fp64_gate(f"480p_closed_loop[K={K},interact({idx})]", core.prob, ocore.prob, o64.prob, o64.topk_margin,
                            tie_quantile=5e-3 if K == 1 else TIE_QUANTILE)
        assert ok, rec


@pytest.mark.parametrize("K,top_k,frames", [(5, 50, 8), (2, 50, 5)])
def test_headline_config_parity_with_fp64_arbitration(synthetic_states, K, top_k, frames):
    """The benchmark's configuration in the test-suite: 480x854, K objects, top_k=50, mem_freq=5, interact(0) then
    interact(last) so that every frame in between is fused (K=5 is BASELINE config 3).  Masks: IoU >= 0.999 vs the fp32
    oracle.  Probabilities: the algorithm is closed-loop, so any two fp32 implementations drift apart over fed-back
    frames; the gate is that the engine stays as close to an fp64 run of the oracle as the fp32 oracle itself does,
    per frame: |engine - fp64| <= 1.5 |oracle_fp32 - fp64| + 2.5e-4 (ARBITRATION_FACTOR; 2 until round 4).  (Both are maxima of rounding noise over 4e5 pixels; where both are tiny - ~3e-4, ratios up to 2 occur - the floor of 2.5e-4, the
    probability equivalent of the 1e-3 logit bar, decides; a drift to 2e-3 fails unless the reference's own fp32 drifts as far.)"""
    sd, fsd = synthetic_states
    prop, fuse = PropagationNetwork(top_k=top_k), FusionNet()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    images, gt = O.synthetic_clip(frames, 480, 854, K, seed=60 + K)
    core = InferenceCore(prop.eval(), fuse.eval(), images, K, mem_freq=5, device=DEV)
    o32 = O.OracleCore(sd, fsd, images, K, mem_freq=5, top_k=top_k)
    o64 = O.OracleCore(sd, fsd, images, K, mem_freq=5, top_k=top_k, dtype=torch.float64, record_margins=True)
    for idx in (0, frames - 1):
        out, r32, r64 = core.interact(gt[idx], idx), o32.interact(gt[idx], idx), o64.interact(gt[idx], idx)
        iou = mean_iou(out, r32, K)
        print(f"K={K} interact({idx}): IoU vs fp32 oracle {iou:.6f}, vs fp64 {mean_iou(out, r64, K):.6f}")
        assert iou >= 0.999
        ok, rec = fp64_gate(f"headline[K={K},interact({idx})]", core.prob, o32.prob, o64.prob, o64.topk_margin)
        assert ok, rec
    assert core.propagated_frames == o32.propagated == 2 * frames - 3


# ------------------------------------------------------------------ InferenceCore behaviour / edge cases

def _small_session(prop, fuse, images, gt, K, order, **kw):
    core = InferenceCore(prop, fuse, images, K, device=DEV, **kw)
    out = None
    for idx in order:
        out = core.interact(gt[idx], idx)
    return core, out


def test_mem_profiles_give_identical_results(nets):
    """mem_profile only moves buffers between host and HBM and bounds the caches (reference :44-63), incl. a
    cache flush per frame (q_buf_size = 1 for profile 3).  Profiles with a small cache encode queries one
    frame at a time, the others in batches (different GEMM tiling / split-K slicing => fp32 rounding-level
    differences), so probabilities agree to 1e-5, masks to a handful of pixels."""
    prop, fuse = nets
    images, gt = O.synthetic_clip(6, 100, 141, 2, seed=41)          # H, W not multiples of 16
    ref_core, ref = _small_session(prop, fuse, images, gt, 2, [0, 5], mem_freq=2, mem_profile=0)
    for mp in (1, 2, 3):
        core, out = _small_session(prop, fuse, images, gt, 2, [0, 5], mem_freq=2, mem_profile=mp)
        assert float((out != ref).mean()) < 1e-4, mp
        assert float((core.prob.cpu() - ref_core.prob.cpu()).abs().max()) < 2e-4, mp
        assert core.images.device.type == "cpu" and core.prob.device.type == ("cuda" if mp == 1 else "cpu")
    assert ref.shape == (6, 100, 141) and ref_core.pad == (1, 2, 6, 6) and ref_core.prob.shape == (3, 6, 1, 112, 144)


def test_concurrent_passes_and_suite_lanes_are_bit_identical(nets):
    """Two independent propagation chains advanced in turn on two HIP streams - the forward / backward pass of a mid-clip interaction
    (InferenceCore._run_passes) and two clips of a suite (eval_suite.run_suite(lanes=2)) - must give the bits of the one-after-the-other
    order: same kernels on the same inputs; anything else is a race (shared scratch, a missing stream dependency)."""
    from mivos_amd import eval_suite as ES
    from mivos_amd.util import synthetic
    prop, fuse = nets
    images, gt = O.synthetic_clip(11, 240, 432, 3, seed=46)
    runs = {}
    from mivos_amd import ops
    for conc in (False, True):
        core = InferenceCore(prop, fuse, images, 3, mem_freq=2, device=DEV)
        core.CONCURRENT_PASSES = conc
        # (round 6: the concurrent order tells the kernels that two streams share the chip - ops.chip_share - and that hint no longer
        # enters the split-K slicing, so the DEFAULT settings of both orders are compared bit for bit; round 5 had to force PASS_CHIP_SHARE = 1)
        outs = [core.interact(gt[idx], idx).copy() for idx in (0, 10, 5, 7)]        # 5 and 7: both passes exist, both fused
        runs[conc] = (outs, core.prob.clone(), core.propagated_frames)
        assert (core._pass_stream is not None) == conc
    for a, b in zip(runs[False][0], runs[True][0]):
        assert np.array_equal(a, b)
    assert torch.equal(runs[False][1], runs[True][1]) and runs[False][2] == runs[True][2]

    specs = [ES.ClipSpec(0, 9, 3, 240, 432, 11), ES.ClipSpec(1, 14, 1, 240, 432, 12), ES.ClipSpec(2, 6, 5, 240, 432, 13), ES.ClipSpec(3, 12, 2, 240, 432, 14)]

    def factory(spec):
        im, g = synthetic.synthetic_clip_device(spec.frames, spec.height, spec.width, spec.objects, seed=spec.seed, device=DEV)
        return InferenceCore(prop, fuse, im, spec.objects, mem_freq=3, device=DEV), g[0]
    # one clip at a time (no hint), two and three clips in flight (ops.chip_share(2 / 3) inside run_suite): identical mask checksums - a suite's
    # results must not depend on how many clips happened to be in flight (round 5: three different checksums, profiles/r05c_lanes_ab.txt)
    one = ES.run_suite(specs, factory, sync=torch.cuda.synchronize)
    two = ES.run_suite(specs, factory, sync=torch.cuda.synchronize, lanes=2, lane_ctx=ES.stream_lanes(DEV, 2))
    three = ES.run_suite(specs, factory, sync=torch.cuda.synchronize, lanes=3, lane_ctx=ES.stream_lanes(DEV, 3))
    assert [(r["clip"], r["checksum"]) for r in one] == [(r["clip"], r["checksum"]) for r in two] == [(r["clip"], r["checksum"]) for r in three]
    assert all(r["lanes"] == 2 for r in two) and all(r["lanes"] == 3 for r in three)
    with ops.chip_share(4):                                         # the hint alone, on one stream: same bits
        hinted = ES.run_suite(specs, factory, sync=torch.cuda.synchronize)
    assert [(r["clip"], r["checksum"]) for r in one] == [(r["clip"], r["checksum"]) for r in hinted]


def test_interaction_order_and_reinteraction_vs_oracle(nets, synthetic_states):
    """Last frame first, then the first frame (pure backward + fused pass), then the SAME frame again
    (certain memory grows by a duplicate, exactly like the reference's torch.cat at :243-245)."""
    prop, fuse = nets
    sd, fsd = synthetic_states
    images, gt = O.synthetic_clip(5, 128, 160, 1, seed=43)
    core = InferenceCore(prop, fuse, images, 1, mem_freq=1, device=DEV)
    ocore = O.OracleCore(sd, fsd, images, 1, mem_freq=1, top_k=20)
    o64 = O.OracleCore(sd, fsd, images, 1, mem_freq=1, top_k=20, dtype=torch.float64, record_margins=True)
    for n, idx in enumerate((4, 0, 0)):
        out, ref, _ = core.interact(gt[idx], idx), ocore.interact(gt[idx], idx), o64.interact(gt[idx], idx)
        assert mean_iou(out, ref, 1) >= 0.999
        ok, rec = fp64_gate(f"reinteraction[{n}:interact({idx})]", core.prob, ocore.prob, o64.prob, o64.topk_margin)
        assert ok, rec
    assert core.certain_mem_k.shape == (1, 128, 3, 8, 10) and core.certain_mem_v.shape == (1, 512, 3, 8, 10)
    assert core.propagated_frames == ocore.propagated


def test_callbacks_and_single_frame_clip(nets):
    prop, fuse = nets
    images, gt = O.synthetic_clip(4, 128, 160, 2, seed=44)
    core = InferenceCore(prop, fuse, images, 2, mem_freq=5, device=DEV)
    totals, steps = [], []
    core.interact(gt[1], 1, total_cb=totals.append, step_cb=lambda: steps.append(1))
    assert totals == [3] and len(steps) == 3                       # front 4 - back -1 - 2 = 3 frames to process
    one, gt1 = O.synthetic_clip(1, 128, 160, 2, seed=45)
    c1 = InferenceCore(prop, fuse, one, 2, device=DEV)
    m = c1.interact(gt1[0], 0)                                      # nothing to propagate: mask is the input
    assert np.array_equal(m[0], gt1[0].argmax(0)[0].numpy().astype(np.uint8))


def test_no_topk_network_closed_loop_vs_oracle(synthetic_states):
    """PropagationNetwork(top_k=None) - the reference's full-softmax reader (prop_net.py:99-102, README's "no top-k" row) - through
    InferenceCore: 4 frames at 240x432, 2 objects, interact(0) then interact(3) (frames 1, 2 fused), against the oracle with
    top_k=None in fp32 and fp64 (fp64-arbitrated gate).  No top-k membership to flip: the two fp32 runs stay at rounding level."""
    sd, fsd = synthetic_states
    prop, fuse = PropagationNetwork(top_k=None), FusionNet()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    images, gt = O.synthetic_clip(4, 240, 432, 2, seed=91)
    core = InferenceCore(prop.to(DEV).eval(), fuse.to(DEV).eval(), images, 2, mem_freq=2, device=DEV)
    o32 = O.OracleCore(sd, fsd, images, 2, mem_freq=2, top_k=None)
    o64 = O.OracleCore(sd, fsd, images, 2, mem_freq=2, top_k=None, dtype=torch.float64)
    for idx in (0, 3):
        out, r32, _ = core.interact(gt[idx], idx), o32.interact(gt[idx], idx), o64.interact(gt[idx], idx)
        assert mean_iou(out, r32, 2) >= 0.999
        ok, rec = fp64_gate(f"no_topk_closed_loop[interact({idx})]", core.prob, o32.prob, o64.prob)
        assert ok, rec
    assert core.propagated_frames == o32.propagated == 5


def test_fp16_range_overflow_is_detected(synthetic_states):
    """The f16x3 operands are fp16 hi + lo pairs: values beyond +-65504 become inf (INTEGRATION.md "Limits").  A clip whose frames are
    1e6 times brighter than anything an image normalisation produces must raise instead of returning garbage masks; the exact-fp32
    mode runs the same clip.  (This check is on the INPUT; overflows arising inside the network: the next test.)"""
    from mivos_amd import ops
    sd, fsd = synthetic_states
    prop, fuse = PropagationNetwork(top_k=20), FusionNet()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    prop, fuse = prop.to(DEV).eval(), fuse.to(DEV).eval()
    images, gt = O.synthetic_clip(3, 128, 160, 1, seed=50)
    with pytest.raises(ops.MivosHipError, match="fp16 range"):
        InferenceCore(prop, fuse, images * 1e6, 1, device=DEV)
    with pytest.raises(ops.MivosHipError, match="fp16 range"):       # NaN frames are refused too (NaN compares false against any bound)
        InferenceCore(prop, fuse, images * float("nan"), 1, device=DEV)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f32"
    try:
        out = InferenceCore(prop, fuse, images * 1e6, 1, device=DEV).interact(gt[0], 0)
    finally:
        ops.CONV_PRECISION = old
    assert out.shape == (3, 128, 160)


def test_fp16_range_overflow_inside_the_network_is_detected(synthetic_states):
    """Round 5: an overflow that arises INSIDE the network (frames fine, a layer's output beyond 65504) is no longer silent - the convolution
    epilogues raise a status word (mivos_conv_desc.status) and InferenceCore reads it once per interaction (ops.check_activation_range).  Weights
    whose decoder blows up: a 3x3 layer of the decoder scaled by 1e6.  The exact-fp32 mode runs the same weights; afterwards the flag is clear and
    the ordinary weights run as before."""
    from mivos_amd import ops
    sd, fsd = synthetic_states
    images, gt = O.synthetic_clip(3, 128, 160, 1, seed=50)
    wild = dict(sd)
    wild["decoder.up_16_8.out_conv.conv1.weight"] = sd["decoder.up_16_8.out_conv.conv1.weight"] * 1e6
    prop, fuse = PropagationNetwork(top_k=20), FusionNet()
    prop.load_state_dict(wild)
    fuse.load_state_dict(fsd)
    prop, fuse = prop.to(DEV).eval(), fuse.to(DEV).eval()
    with pytest.raises(ops.MivosHipError, match="left the fp16 range"):
        InferenceCore(prop, fuse, images, 1, device=DEV).interact(gt[0], 0)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f32"
    try:
        out = InferenceCore(prop, fuse, images, 1, device=DEV).interact(gt[0], 0)
    finally:
        ops.CONV_PRECISION = old
    assert out.shape == (3, 128, 160)
    ops.check_activation_range(torch.device(DEV))                    # the raise cleared the flag
    prop.load_state_dict(sd)
    prop = prop.to(DEV).eval()
    assert InferenceCore(prop, fuse, images, 1, device=DEV).interact(gt[0], 0).shape == (3, 128, 160)


def test_topk_larger_than_memory_raises_like_reference(nets):
    """64x96 frame -> 24 memory positions at T=1 < top_k=50: the reference dies in torch.topk
    ('selected index k out of range'); so do we (SURVEY.md §7 hard part 2)."""
    prop50 = PropagationNetwork(top_k=50)
    prop50.load_state_dict(nets[0].state_dict())
    images, gt = O.synthetic_clip(3, 64, 96, 1, seed=46)
    core = InferenceCore(prop50, nets[1], images, 1, device=DEV)
    with pytest.raises(RuntimeError, match="out of range"):
        core.interact(gt[0], 0)


def test_1080p_three_objects_with_fusion_vs_oracle(synthetic_states):
    """BASELINE config 5 geometry with its object count: 1080x1920, K = 3, top_k = 50, 3 frames, interact(0) then
    interact(2): bank depth up to T = 2, the middle frame fused (3 propagated frames), closed loop.  Masks IoU >= 0.999 vs
    the fp32 oracle; probabilities through the fp64-arbitrated gate (fp64_gate; one fp64 frame of the CPU oracle costs ~40 s
    at this size).  The deep-bank regime of config 5 (400 k - 1.2 M memory positions per object, > 2^31-byte bank strides)
    is pinned by tests/test_gpu_ops.py::test_memory_read_deep_bank_1080p_vs_chunked_oracle."""
    sd, fsd = synthetic_states
    K = 3
    prop, fuse = PropagationNetwork(top_k=50), FusionNet()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    images, gt = O.synthetic_clip(3, 1080, 1920, K, seed=71)
    core = InferenceCore(prop.eval(), fuse.eval(), images, K, mem_freq=1, device=DEV)
    o32 = O.OracleCore(sd, fsd, images, K, mem_freq=1, top_k=50)
    o64 = O.OracleCore(sd, fsd, images, K, mem_freq=1, top_k=50, dtype=torch.float64, record_margins=True)
    for idx in (0, 2):
        out, r32, _ = core.interact(gt[idx], idx), o32.interact(gt[idx], idx), o64.interact(gt[idx], idx)
        iou = mean_iou(out, r32, K)
        print(f"1080p K=3 interact({idx}): IoU vs fp32 oracle {iou:.6f}")
        assert iou >= 0.999
        ok, rec = fp64_gate(f"1080p_K3_closed_loop[interact({idx})]", core.prob, o32.prob, o64.prob, o64.topk_margin)
        assert ok, rec
    assert core.propagated_frames == 3 and core.prob.shape == (4, 3, 1, 1088, 1920)


def test_1080p_memory_read_and_single_step_vs_oracle(nets, synthetic_states):
    """BASELINE config 5 geometry (1080x1920 -> 1088x1920, HW = 8160), K = 1, T = 1: keys / values / logits."""
    prop, _ = nets
    sd = synthetic_states[0]
    images, gt = O.synthetic_clip(2, 1080, 1920, 1, seed=47)
    img, _ = O.pad_divide_by(images, 16)
    m0, _ = O.pad_divide_by(gt[0], 16)
    ok, ov = O.memorize(sd, img[:, 0], m0[1:])
    oq = O.get_query_values(sd, img[:, 1])
    ref = O.segment_logits(sd, ok, ov, *oq, top_k=20)[:, 0]
    k, v = prop.memorize_into(img[:, 0].to(DEV), m0[1:].to(DEV))
    assert k.shape == (1, 68, 120, 128)
    q = prop.encode_query(img[:, 1].to(DEV))
    got = prop.segment(k.reshape(1, -1, 128), v.reshape(1, -1, 512), q, logits=True).cpu()
    d = (got - ref).abs()
    print(f"1080p: max|dlogit| {float(d.max()):.2e}  frac(|d|>1e-3) {float((d > 1e-3).float().mean()):.2e}")
    assert float(d.max()) < LOGIT_TOL
