"""Pins oracle/stm_oracle.py (the CPU restatement) to the golden vectors that
oracle/make_golden.py produced by running the UNMODIFIED reference.
The reference ships no tests of its own for this path, so these fixtures are the pin."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import stm_oracle as O
from oracle import weights as Wt

torch.set_grad_enabled(False)
TOL = 2e-6   # same ops, same library: expected bit-exact on this image; slack for other CPUs


@pytest.fixture(scope="module")
def ops(golden_dir):
    with np.load(os.path.join(golden_dir, "ops_small.npz")) as z:
        return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_state_dict_layout_matches_reference(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    assert len(keys["prop"]) == 597 and len(keys["fuse"]) == 12
    assert {k: list(v) for k, v in Wt.prop_spec().items()} == keys["prop"]
    assert {k: list(v) for k, v in Wt.fuse_spec().items()} == keys["fuse"]


def test_synthetic_weights_are_reproducible(ops, synthetic_states):
    sd, fsd = synthetic_states
    assert Wt.state_fingerprint(sd) == pytest.approx(float(ops["fingerprint_prop"]), rel=1e-12)
    assert Wt.state_fingerprint(fsd) == pytest.approx(float(ops["fingerprint_fuse"]), rel=1e-12)


def test_memory_read(ops):
    out = O.memory_read(T(ops["mr_mk"]), T(ops["mr_mv"]), T(ops["mr_qk"]), 20)
    assert float((out - T(ops["mr_out"])).abs().max()) <= TOL


def test_chunked_memory_read_equals_the_materialised_oracle(ops):
    """oracle/chunked_read.py (the checker of the deep-bank GPU tests, where the [THW x HW] affinity cannot be materialised)
    against stm_oracle.memory_read == the unmodified reference: golden vector, top-k and full-softmax cases, ragged query
    blocks, fp32 and fp64."""
    from oracle import chunked_read as CR
    mk, mv, qk = T(ops["mr_mk"]), T(ops["mr_mv"]), T(ops["mr_qk"])
    assert float((CR.memory_read(mk, mv, qk, 20) - T(ops["mr_out"])).abs().max()) <= 5e-6
    g = torch.Generator().manual_seed(5)
    mk, mv, qk = torch.randn(2, 128, 3, 7, 9, generator=g) * 1.5, torch.randn(2, 512, 3, 7, 9, generator=g), torch.randn(1, 128, 7, 9, generator=g) * 1.5
    for top_k in (5, 50, None):
        for dt in (torch.float32, torch.float64):
            ref = torch.cat([O.memory_read(mk[o:o + 1].to(dt), mv[o:o + 1].to(dt), qk.to(dt), top_k) for o in range(2)], 0)
            got = CR.memory_read(mk, mv, qk, top_k, dtype=dt, qblock=16)          # 63 queries: blocks of 16 + a ragged one
            assert got.dtype == dt and float((got - ref).abs().max()) <= (5e-6 if dt == torch.float32 else 1e-12), (top_k, dt)
    # index sets / weights / margins of the row-layout entry point
    keys, vals = mk.reshape(2, 128, -1).transpose(1, 2), mv.reshape(2, 512, -1).transpose(1, 2)
    r = CR.memory_read_rows(keys, vals, qk.reshape(128, -1).t(), 50, dtype=torch.float64, qblock=16)
    a = O.affinity(mk.double(), qk.double())                                       # [2, THW, HW]
    v, i = torch.topk(a, 51, dim=1)
    assert torch.equal(r["idx"], i[:, :50].transpose(1, 2)) and float((r["margin"] - (v[:, 49] - v[:, 50])).abs().max()) < 1e-12
    assert float((r["weights"].sum(2) - 1).abs().max()) < 1e-12


def test_memory_read_topk_larger_than_memory_raises(ops):
    # reference behaviour: torch.topk raises when THW < top_k (SURVEY.md §7 hard part 2)
    with pytest.raises(RuntimeError):
        O.memory_read(T(ops["mr_mk"])[:, :, :1, :2, :2], T(ops["mr_mv"])[:, :, :1, :2, :2], T(ops["mr_qk"])[:, :, :2, :2], 20)


def test_aggregate(ops):
    p = T(ops["ag_in"])
    assert float((O.aggregate_wbg(p, True) - T(ops["ag_soft"])).abs().max()) <= TOL
    assert float((O.aggregate_wbg(p, True, hard=True) - T(ops["ag_hard"])).abs().max()) <= TOL
    assert float((O.aggregate_sbg(p, True) - T(ops["ag_sbg"])).abs().max()) <= TOL
    assert float((O.aggregate_wbg(p, False) - T(ops["ag_soft"])[1:]).abs().max()) <= TOL


def test_get_attention(ops):
    a = O.get_attention(T(ops["at_mk"]), T(ops["at_pos"]), T(ops["at_neg"]), T(ops["at_qk"]))
    assert float((a - T(ops["at_out"])).abs().max()) <= TOL


def test_fusion_net(ops, synthetic_states):
    o = O.fusion_net(synthetic_states[1], T(ops["fu_im"]), T(ops["fu_s1"]), T(ops["fu_s2"]), T(ops["fu_at"]), T(ops["fu_tm"]))
    assert float((o - T(ops["fu_out"])).abs().max()) <= 1e-5


def test_encoders_and_decoder(ops, synthetic_states):
    sd = synthetic_states[0]
    k, v = O.memorize(sd, T(ops["en_frame"]), T(ops["en_masks"]))
    assert float((k - T(ops["en_mk"])).abs().max()) <= 1e-5
    assert float((v - T(ops["en_mv"])).abs().max()) <= 1e-5
    q = O.get_query_values(sd, T(ops["en_frame"]))
    for got, name in zip(q, ("en_f16", "en_f8", "en_f4", "en_qk", "en_qv")):
        assert float((got - T(ops[name])).abs().max()) <= 1e-5, name
    d = O.decoder(sd, T(ops["de_m4"]), q[1], q[2])
    assert float((d - T(ops["de_out"])).abs().max()) <= 1e-4


def test_pad_divide_by():
    x = torch.ones(1, 3, 480, 854)
    y, pad = O.pad_divide_by(x, 16)
    assert y.shape[-2:] == (480, 864) and pad == (5, 5, 0, 0)
    y, pad = O.pad_divide_by(torch.ones(1, 1, 120, 150), 16)
    assert y.shape[-2:] == (128, 160) and pad == (5, 5, 4, 4)
    y, pad = O.pad_divide_by(torch.ones(1, 1, 101, 35), 16)     # odd deltas: low side gets floor
    assert pad == (6, 7, 5, 6)


def test_end_to_end_inference_core(golden_dir, synthetic_states):
    """Three interactions (incl. fusion between interacted frames) vs the reference's
    InferenceCore: identical masks, probabilities and schedule trace."""
    sd, fsd = synthetic_states
    with np.load(os.path.join(golden_dir, "e2e_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    c = json.loads(str(g["config"]))
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    core = O.OracleCore(sd, fsd, images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"])
    for n, idx in enumerate(c["interactions"]):
        out = core.interact(gt[idx], idx)
        assert (out != g[f"masks_{n}"]).mean() <= 1e-4
        assert float((core.prob - T(g[f"prob_{n}"])).abs().max()) <= 1e-4
    assert " ".join(core.trace) == str(g["trace"])


def test_attention_read_network_and_channel_aggregate(golden_dir, synthetic_states):
    """attn_small.npz: the reference's AttentionReadNetwork / AttentionMemory (model/attn_network.py) and
    aggregate_wbg_channel (model/aggregate.py:39-53)."""
    with np.load(os.path.join(golden_dir, "attn_small.npz")) as z:
        g = {k: T(z[k]) for k in z.files}
    o1, o2 = O.attention_read_network(synthetic_states[0], g["an_image"], g["an_m11"], g["an_m21"], g["an_m12"], g["an_m22"], g["an_query"])
    assert float((o1 - g["an_out1"]).abs().max()) <= TOL and float((o2 - g["an_out2"]).abs().max()) <= TOL
    assert float((O.attention_weights(g["aw_mk"], g["aw_qk"]) - g["aw_out"]).abs().max()) <= TOL
    for hard in (0, 1):
        lg, sm = O.aggregate_wbg_channel(g["ac_in"], keep_bg=True, hard=bool(hard))
        assert float((lg - g[f"ac_logits_{hard}"]).abs().max()) <= TOL * 1000 ** hard
        assert float((sm - g[f"ac_soft_{hard}"]).abs().max()) <= TOL


def test_s2m_network(golden_dir):
    """s2m_small.npz / s2m_state_dict_keys.json: the reference's deeplabv3plus_resnet50 (model/s2m) - state_dict layout of the
    synthetic weights and the forward pass of the CPU restatement."""
    from oracle import s2m_oracle as SO
    keys = json.load(open(os.path.join(golden_dir, "s2m_state_dict_keys.json")))
    assert len(keys) == 368 and {k: list(v) for k, v in Wt.s2m_spec().items()} == keys
    with np.load(os.path.join(golden_dir, "s2m_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    sd = Wt.make_s2m_state(0)
    assert Wt.state_fingerprint(sd) == pytest.approx(float(g["fingerprint"]), rel=1e-12)
    out = SO.s2m_forward(sd, T(g["s2m_in"]))
    assert float((out - T(g["s2m_out"])).abs().max()) <= 1e-4
    m = torch.zeros(1, 1, 9, 9)
    m[0, 0, 4, 4] = m[0, 0, 0, 8] = 1
    d = SO.dilate3x3(m)
    assert d.sum() == 9 + 4 and d[0, 0, 3:6, 3:6].min() == 1            # 3x3 max filter, zero outside the image


def test_train_step_oracle_matches_the_reference(golden_dir):
    """oracle/train_oracle.py (the checker of the GPU training-step tests at sizes the fixture does not hold) against
    tests/golden/train_small.npz = the unmodified reference's FusionNet / aggregate_wbg_channel / LossComputer / Adam on the
    same batch, before, inside and after BootstrappedCE's warm-up."""
    from oracle import train_oracle as TO
    with np.load(os.path.join(golden_dir, "train_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    cfg = json.loads(str(g["config"]))
    data = {k[3:]: T(v) for k, v in g.items() if k.startswith("in.")}
    fsd = Wt.make_fuse_state(0)
    for it in cfg["its"]:
        tag = f"it{it}."
        with torch.enable_grad():
            r = TO.train_step(fsd, data, T(g[tag + "attn1"]), T(g[tag + "attn2"]), it, cfg["iterations"], cfg["lr"])
        assert float((r["logits"] - T(g[tag + "logits"])).abs().max()) <= 1e-5
        assert float((r["mask"] - T(g[tag + "mask"])).abs().max()) <= 1e-6
        assert abs(r["total_loss"] - float(g[tag + "total_loss"])) <= 1e-6 * max(1.0, abs(float(g[tag + "total_loss"])))
        assert abs(r["p"] - float(g[tag + "p"])) <= 1e-12
        for n in fsd:
            ref = T(g[tag + "grad." + n])
            assert float((r["grads"][n] - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (it, n)
            assert float((r["new"][n] - T(g[tag + "new." + n])).abs().max()) <= 2e-6, (it, n)       # updates are ~lr = 1e-4


def test_generator_oracle_matches_the_reference(golden_dir, synthetic_states):
    """oracle.OracleGenerator vs tests/golden/gen_small.npz (the unmodified generation/fusion_generator.py, oracle/make_golden_gen.py)."""
    with np.load(os.path.join(golden_dir, "gen_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    c = json.loads(str(g["config"]))
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    gen = O.OracleGenerator(synthetic_states[0], images, c["mem_freq"], top_k=c["top_k"])
    for n, (idx, left, right) in enumerate(c["calls"]):
        gen.reset(c["k"])
        out = gen.interact_mask(gt[idx, 1:], idx, left, right)
        assert out.shape == g[f"prob_{n}"].shape and float((out - T(g[f"prob_{n}"])).abs().max()) <= TOL


def test_update_mask_only_golden(golden_dir, synthetic_states):
    """OracleCore.update_mask_only vs the unmodified reference's (inference_core.py:273-293; tests/golden/update_small.npz,
    oracle/make_golden_update.py): soft probabilities with exact ties, a one-hot mask, a propagated frame's own probabilities;
    unevenly padded frame (100x141 -> 112x144)."""
    from oracle.make_golden_update import update_inputs
    with np.load(os.path.join(golden_dir, "update_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    c = json.loads(str(g["config"]))
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    core = O.OracleCore(*synthetic_states, images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"])
    assert tuple(core.pad) == tuple(int(v) for v in g["pad"]) == (1, 2, 6, 6)
    assert np.array_equal(core.interact(gt[0], 0), g["masks_interact"])
    assert float((core.prob[:, c["calls"][2]] - T(g["input_2"])).abs().max()) <= TOL
    for n, (idx, pm) in enumerate(zip(c["calls"], update_inputs(c, T(g["input_2"])))):
        out = core.update_mask_only(pm, idx)
        assert out.dtype == np.uint8 and np.array_equal(out, g[f"np_masks_{n}"])
        assert np.array_equal(core.masks[idx].numpy(), g[f"masks_idx_{n}"])


def test_gui_call_pattern_golden(golden_dir, synthetic_states):
    """The GUI's way of driving the processor (interactive_gui.py:550, 616, 626, 636-642, 889-897, 955-960; oracle/gui_replay.py) replayed
    on the oracle: `current_mask` after every handler, the progress-bar calls, the final buffers and what local mode reads equal what the
    UNMODIFIED reference InferenceCore produced (tests/golden/gui_small.npz, oracle/make_golden_gui.py)."""
    from oracle import gui_replay as G
    from oracle.make_golden_gui import pack
    sd, fsd = G.session_states()
    with np.load(os.path.join(golden_dir, "gui_small.npz")) as z:
        gold = {k: z[k] for k in z.files}
    cfg = json.loads(str(gold["config"]))
    assert cfg == G.SESSION
    images, gt = O.synthetic_clip(cfg["t"], cfg["h"], cfg["w"], cfg["k"], cfg["seed"])
    core = O.OracleCore(sd, fsd, images, cfg["k"], mem_freq=cfg["mem_freq"], top_k=cfg["top_k"])
    g, local = G.scripted_session(core, gt)
    out = pack(g, local, core)
    assert [n for n, _ in g.events] == [str(n) for n in gold["event_names"]]
    assert out["progress"].tolist() == gold["progress"].tolist() == [6] + [-1] * 6 + [5] + [-1] * 5 + [4] + [-1] * 4 + [1, -1]
    for k in gold:
        if k in ("config", "event_names", "final_prob", "local_prev_soft_mask", "self_iou_fp32_vs_fp64"):
            continue
        assert np.array_equal(out[k], gold[k]), k
    assert float(gold["self_iou_fp32_vs_fp64"].min()) >= 0.9995       # the fixture's admission: the reference agrees with its own fp64 run
    assert np.abs(out["final_prob"] - gold["final_prob"]).max() <= TOL
    assert np.abs(out["local_prev_soft_mask"] - gold["local_prev_soft_mask"]).max() <= TOL
    # the in-place reset of frame 2 reached the processor's own buffers and the next interaction rebuilt them
    names = [n for n, _ in g.events]
    reset = g.events[names.index("reset")][1]
    assert not reset[2].any() and reset[3].any()


def test_long_horizon_fixtures_are_pinned_to_the_oracle(golden_dir):
    """tests/golden/long_s<seed>_k<K>.npz (oracle/make_golden_long.py: the UNMODIFIED reference's fp32 run and an fp64 run of config 3's full 137-step
    session) - what a CPU suite can check in seconds: every fixture is complete and self-consistent (shapes, the admission record it carries, the
    conditioning it was made with = the FROZEN synthetic.CLOSED_LOOP_CONDITIONING), and for one of them the FIRST propagated frame of the session,
    recomputed here by the oracle on the same clip and weights, reproduces the stored reference masks and fp64 probabilities."""
    import glob
    from mivos_amd.util import synthetic
    from mivos_amd.util.tensor_util import compute_np_iou
    from oracle import make_golden_long as G
    paths = sorted(glob.glob(os.path.join(golden_dir, "long_s*_k*.npz")))
    assert len(paths) >= 3
    for path in paths:
        z = G.load(path)
        cfg = json.loads(str(z["config"]))
        K, T, sub = cfg["objects"], cfg["frames"], cfg["sub"]
        assert cfg["conditioning"] == synthetic.CLOSED_LOOP_CONDITIONING and cfg["interactions"] == [0, T - 1] and T == 70 and cfg["top_k"] == 50
        nh, nw = (cfg["height"] + 15) // 16 * 16, (cfg["width"] + 15) // 16 * 16
        kept = z["frames"]
        for n in range(2):
            assert z[f"masks32_{n}"].shape == (T, cfg["height"], cfg["width"]) and z[f"masks32_{n}"].dtype == np.uint8 and z[f"masks64_{n}"].shape == z[f"masks32_{n}"].shape
            assert z[f"p64_{n}"].shape == (K + 1, len(kept), -(-nh // sub), -(-nw // sub)) == z[f"d32_{n}"].shape
            adm = z[f"admission_{n}"]
            live = adm[~np.isnan(adm)]
            assert len(live) == (T - 1 if n == 0 else T - 2)
            # the stored admission IS the IoU of the two stored mask sets
            t = int(np.nanargmin(adm))
            iou = float(np.mean([compute_np_iou(z[f"masks32_{n}"][t] == j, z[f"masks64_{n}"][t] == j) for j in range(1, K + 1)]))
            assert abs(iou - adm[t]) < 1e-9
        worst = min(float(np.nanmin(z["admission_0"])), float(np.nanmin(z["admission_1"])))
        assert abs(worst - cfg["worst_self_iou"]) < 1e-9 and cfg["admitted"] == (worst >= cfg["admission_bar"])
    # first propagated frame of the smallest session, recomputed (fp32: masks; fp64: probabilities at the kept samples of frame 0 .. the first kept frame > 0 is 5)
    path = min(paths, key=lambda p: json.loads(str(np.load(p)["config"]))["objects"])
    z = G.load(path)
    cfg = json.loads(str(z["config"]))
    K = cfg["objects"]
    images, gt = synthetic.synthetic_clip(cfg["frames"], cfg["height"], cfg["width"], K, seed=cfg["seed"])
    sd = synthetic.condition_state(synthetic.make_prop_state(0), **cfg["conditioning"])
    fsd = synthetic.make_fuse_state(0)
    core = O.OracleCore(sd, fsd, images[:, :2], K, mem_freq=cfg["mem_freq"], top_k=cfg["top_k"])
    masks = core.interact(gt[0], 0)
    iou = float(np.mean([compute_np_iou(masks[1] == j, z["masks32_0"][1] == j) for j in range(1, K + 1)]))
    assert iou >= 0.99995, iou            # (bit-identical up to the host's thread count: the fixture was made with another number of threads)
