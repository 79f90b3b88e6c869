"""Generate tests/golden/* by running the UNMODIFIED reference on CPU (this container only).

    python -m oracle.make_golden            # needs /root/reference

Writes
  calib_*_seed0.npz      calibrated BatchNorm running statistics + conv gains of the synthetic weights
  state_dict_keys.json   the reference's state_dict names + shapes (PropagationNetwork, FusionNet)
  ops_small.npz          per-op known-answer vectors produced by the reference's own modules
  e2e_small.npz          end-to-end InferenceCore run (3 interactions incl. fusion): masks,
                         probabilities, schedule trace
and prints how far oracle/stm_oracle.py is from each of them (should be ~1e-6 or exact).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import stm_oracle as O            # noqa: E402
from oracle import weights as Wt              # noqa: E402
from oracle.ref_loader import build_reference_networks  # noqa: E402

G = Wt.GOLDEN_DIR
E2E = dict(t=7, h=120, w=150, k=2, seed=3, mem_freq=2, top_k=20, interactions=[0, 6, 3])


def _np(x):
    return x.detach().cpu().numpy()


def make_attn_golden():
    """attn_small.npz: the reference's AttentionReadNetwork (model/attn_network.py:30-80, the training-time twin of
    get_attention) and aggregate_wbg_channel (model/aggregate.py:39-53) on seeded inputs."""
    torch.set_grad_enabled(False)
    ref = __import__("oracle.ref_loader", fromlist=["load_reference"]).load_reference()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref["attn_network"].AttentionReadNetwork().eval()
    sd = Wt.make_prop_state(0)
    missing = net.load_state_dict({k: v for k, v in sd.items() if not k.startswith("decoder.")}, strict=True)
    r = np.random.RandomState(11)
    b, h, w = 2, 64, 96
    image = torch.from_numpy(r.standard_normal((b, 3, h, w)).astype(np.float32))
    query = torch.from_numpy(r.standard_normal((b, 3, h, w)).astype(np.float32))
    m = [torch.from_numpy((r.rand(b, 1, h, w) > thr).astype(np.float32)) for thr in (0.5, 0.6, 0.7, 0.55)]
    a1, a2 = net(image, m[0], m[1], m[2], m[3], query)
    out = dict(an_image=_np(image), an_query=_np(query), an_m11=_np(m[0]), an_m21=_np(m[1]), an_m12=_np(m[2]), an_m22=_np(m[3]),
               an_out1=_np(a1), an_out2=_np(a2))
    o1, o2 = O.attention_read_network(sd, image, m[0], m[1], m[2], m[3], query)
    print("attn_network |oracle-ref|", float((o1 - a1).abs().max()), float((o2 - a2).abs().max()))
    # dense W of AttentionMemory (attn_network.py:17-28) on small random keys
    mk, qk = torch.from_numpy(r.standard_normal((2, 128, 4, 6)).astype(np.float32)), torch.from_numpy(r.standard_normal((2, 128, 4, 6)).astype(np.float32))
    W = net.memory(mk, qk)
    out.update(aw_mk=_np(mk), aw_qk=_np(qk), aw_out=_np(W))
    print("dense W      |oracle-ref|", float((O.attention_weights(mk, qk) - W).abs().max()))
    # aggregate_wbg_channel
    p = torch.from_numpy(r.rand(2, 3, 16, 20).astype(np.float32))
    p[:, :, :2] = 0.0
    p[:, :, 2:4] = 1.0
    for hard in (False, True):
        lg, sm = ref["aggregate"].aggregate_wbg_channel(p, keep_bg=True, hard=hard)
        out.update({f"ac_logits_{int(hard)}": _np(lg), f"ac_soft_{int(hard)}": _np(sm)})
        olg, osm = O.aggregate_wbg_channel(p, keep_bg=True, hard=hard)
        print("aggregate_ch |oracle-ref|", float((olg - lg).abs().max()), float((osm - sm).abs().max()))
    out["ac_in"] = _np(p)
    np.savez_compressed(os.path.join(G, "attn_small.npz"), **out)
    print("attn_small.npz", os.path.getsize(os.path.join(G, "attn_small.npz")) // 1024, "KiB")


def make_s2m_golden():
    """calib_s2m_seed0.npz, s2m_state_dict_keys.json, s2m_small.npz: the reference's scribble-to-mask network
    (model/s2m/s2m_network.py:56-65 deeplabv3plus_resnet50) on a seeded 64x96 input."""
    import contextlib, importlib, io
    from oracle import s2m_oracle as SO
    from oracle.ref_loader import load_reference
    torch.set_grad_enabled(False)
    load_reference()
    calib = Wt.calibrate_s2m(0)
    np.savez_compressed(os.path.join(G, "calib_s2m_seed0.npz"), **{k: _np(v) for k, v in calib.items()})
    with contextlib.redirect_stdout(io.StringIO()):
        net = importlib.import_module("model.s2m.s2m_network").deeplabv3plus_resnet50().eval()
    with open(os.path.join(G, "s2m_state_dict_keys.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in net.state_dict().items()}, f, indent=0)
    sd = Wt.make_s2m_state(0)
    net.load_state_dict(sd, strict=True)
    # a realistic input (normalised synthetic frame, imperfect current mask, sparse positive / negative scribbles): the
    # synthetic BN statistics are calibrated on this kind of data, white noise would give logits of +-400
    images, gt = O.synthetic_clip(2, 64, 96, 2, seed=321)
    g = torch.Generator().manual_seed(5)
    cur = gt[1:2, 1] * (torch.rand(1, 1, 64, 96, generator=g) > 0.25).float()
    pos = (torch.rand(1, 1, 64, 96, generator=g) > 0.96).float() * gt[1:2, 1]
    neg = (torch.rand(1, 1, 64, 96, generator=g) > 0.96).float() * (1 - gt[1:2, 1])
    x = torch.cat([images[0, 1:2], cur, pos, neg], 1)
    out = net(x)
    o = SO.s2m_forward(sd, x)
    print("s2m          |oracle-ref|", float((o - out).abs().max()), "logit range", float(out.min()), float(out.max()), "std", float(out.std()))
    np.savez_compressed(os.path.join(G, "s2m_small.npz"), s2m_in=_np(x), s2m_out=_np(out), fingerprint=np.float64(Wt.state_fingerprint(sd)))


def main():
    if "--attn-only" in sys.argv:
        return make_attn_golden()
    if "--s2m-only" in sys.argv:
        return make_s2m_golden()
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    os.makedirs(G, exist_ok=True)

    # 1. BN calibration --------------------------------------------------------------
    pc, fc = Wt.calibrate(seed=0)
    np.savez_compressed(os.path.join(G, "calib_prop_seed0.npz"), **{k: _np(v) for k, v in pc.items()})
    np.savez_compressed(os.path.join(G, "calib_fuse_seed0.npz"), **{k: _np(v) for k, v in fc.items()})
    print("calibration:", len(pc), "+", len(fc), "tensors")

    # 2. reference networks with the synthetic state ---------------------------------
    ref, prop, fuse = build_reference_networks(top_k=E2E["top_k"])
    keys = {"prop": {k: list(v.shape) for k, v in prop.state_dict().items()},
            "fuse": {k: list(v.shape) for k, v in fuse.state_dict().items()}}
    with open(os.path.join(G, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
    sd, fsd = Wt.make_prop_state(0), Wt.make_fuse_state(0)
    prop.load_state_dict(sd, strict=True)
    fuse.load_state_dict(fsd, strict=True)
    fp = {"prop": Wt.state_fingerprint(sd), "fuse": Wt.state_fingerprint(fsd)}
    print("fingerprints", fp)

    # 3. per-op vectors --------------------------------------------------------------
    r = np.random.RandomState(7)
    f32 = lambda *s: torch.from_numpy(r.standard_normal(s).astype(np.float32))
    ops = {"fingerprint_prop": np.float64(fp["prop"]), "fingerprint_fuse": np.float64(fp["fuse"])}

    # memory read, T=3, 8x10 grid, top-20 (prop_net.py:81-108)
    mk, mv, qk = f32(1, 128, 3, 8, 10), f32(1, 512, 3, 8, 10), f32(1, 128, 8, 10)
    out = prop.memory(mk, mv, qk)
    ops.update(mr_mk=_np(mk), mr_mv=_np(mv), mr_qk=_np(qk), mr_out=_np(out))
    print("memory_read  |oracle-ref|", float((O.memory_read(mk, mv, qk, 20) - out).abs().max()))

    # aggregate (aggregate.py:22-37)
    p = torch.rand(3, 1, 16, 20)
    p[:, :, :2] = 0.0
    p[:, :, 2:4] = 1.0
    ops.update(ag_in=_np(p), ag_soft=_np(ref["aggregate"].aggregate_wbg(p, keep_bg=True)),
               ag_hard=_np(ref["aggregate"].aggregate_wbg(p, keep_bg=True, hard=True)),
               ag_sbg=_np(ref["aggregate"].aggregate_sbg(p, keep_bg=True)))
    print("aggregate    |oracle-ref|", float((O.aggregate_wbg(p, True) - torch.from_numpy(ops["ag_soft"])).abs().max()))

    # get_attention (prop_net.py:187-200), 64x80 frame
    mk16, qk16 = f32(1, 128, 1, 4, 5), f32(1, 128, 4, 5)
    pos, neg = (torch.rand(1, 1, 64, 80) > 0.7).float(), (torch.rand(1, 1, 64, 80) > 0.8).float()
    att = prop.get_attention(mk16, pos, neg, qk16)
    ops.update(at_mk=_np(mk16), at_qk=_np(qk16), at_pos=_np(pos), at_neg=_np(neg), at_out=_np(att))
    print("get_attention|oracle-ref|", float((O.get_attention(mk16, pos, neg, qk16) - att).abs().max()))

    # FusionNet (fusion_net.py:32-50), 48x64
    im, s1, s2, at, tm = f32(1, 3, 48, 64), torch.rand(1, 1, 48, 64), torch.rand(1, 1, 48, 64), torch.rand(1, 2, 48, 64), torch.tensor([[0.25, 0.75]])
    fo = fuse(im, s1, s2, at, tm)
    ops.update(fu_im=_np(im), fu_s1=_np(s1), fu_s2=_np(s2), fu_at=_np(at), fu_tm=_np(tm), fu_out=_np(fo))
    print("fusion_net   |oracle-ref|", float((O.fusion_net(fsd, im, s1, s2, at, tm) - fo).abs().max()))

    # encoders + decoder on one 64x96 frame, 2 objects (prop_net.py:144-181)
    frame = f32(1, 3, 64, 96)
    masks = (torch.rand(2, 1, 64, 96) > 0.6).float()
    k16m, v16m = prop.memorize(frame, masks)
    q = prop.get_query_values(frame)
    ops.update(en_frame=_np(frame), en_masks=_np(masks), en_mk=_np(k16m), en_mv=_np(v16m),
               en_f16=_np(q[0]), en_f8=_np(q[1]), en_f4=_np(q[2]), en_qk=_np(q[3]), en_qv=_np(q[4]))
    ok, ov = O.memorize(sd, frame, masks)
    oq = O.get_query_values(sd, frame)
    print("memorize     |oracle-ref|", float((ok - k16m).abs().max()), float((ov - v16m).abs().max()))
    print("query        |oracle-ref|", [float((a - b).abs().max()) for a, b in zip(oq, q)])
    m4 = f32(2, 1024, 4, 6)
    dl = prop.decoder(m4, q[1], q[2])
    ops.update(de_m4=_np(m4), de_out=_np(dl))
    print("decoder      |oracle-ref|", float((O.decoder(sd, m4, q[1], q[2]) - dl).abs().max()), "range", float(dl.min()), float(dl.max()))
    np.savez_compressed(os.path.join(G, "ops_small.npz"), **ops)

    # 4. end-to-end with the reference InferenceCore ----------------------------------
    # pick the clip whose top-k selections have the widest margin and where the fp32 and fp64
    # runs of the algorithm agree: an end-to-end fixture must not hinge on rounding-level ties
    c = dict(E2E)
    best = None
    for seed in range(3, 11):
        images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], seed)
        O.TOPK_GAP = []
        a = O.OracleCore(sd, fsd, images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"])
        b = O.OracleCore(sd, fsd, images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"], dtype=torch.float64)
        for idx in c["interactions"]:
            ma, mb = a.interact(gt[idx], idx), b.interact(gt[idx], idx)
        gap = min(O.TOPK_GAP)
        O.TOPK_GAP = None
        dp = float((a.prob.double() - b.prob).abs().max())
        print(f"seed {seed}: min top-k margin {gap:.2e}, fp32-vs-fp64 |dprob| {dp:.2e}, mask mismatch {int((ma != mb).sum())}")
        score = (dp < 2e-4, gap)
        if best is None or score > best[0]:
            best = (score, seed)
    c["seed"] = best[1]
    print("chosen seed", c["seed"])
    images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
    trace = []

    def wrap(obj, name, tag):
        fn = getattr(obj, name)

        def inner(*a, **kw):
            trace.append(tag(*a))
            return fn(*a, **kw)
        setattr(obj, name, inner)

    wrap(prop, "memorize", lambda *a: "M")
    wrap(prop, "get_query_values", lambda *a: "Q")
    wrap(prop, "segment_with_query", lambda keys, *a: f"S{keys.shape[2]}")
    fuse_fwd = fuse.forward
    fuse.forward = lambda im, s1, s2, at, tm: (trace.append(f"F({float(tm[0,0]):.2f},{float(tm[0,1]):.2f})"), fuse_fwd(im, s1, s2, at, tm))[1]

    core = ref["inference_core"].InferenceCore(prop, fuse, images, c["k"], mem_profile=0, mem_freq=c["mem_freq"], device="cpu")
    ocore = O.OracleCore(sd, fsd, images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"])
    e2e = {"config": json.dumps(c)}
    for n, idx in enumerate(c["interactions"]):
        out = core.interact(gt[idx], idx)
        oout = ocore.interact(gt[idx], idx)
        e2e[f"masks_{n}"] = out.copy()
        e2e[f"prob_{n}"] = _np(core.prob).astype(np.float32)
        print(f"interact({idx}): mask mismatch px {int((out != oout).sum())}, |prob diff| {float((core.prob - ocore.prob).abs().max()):.3e}")
    e2e["trace"] = " ".join(trace)
    print("trace equal:", trace == ocore.trace)
    print(" ".join(trace))
    np.savez_compressed(os.path.join(G, "e2e_small.npz"), **e2e)
    make_attn_golden()
    make_s2m_golden()
    for f in sorted(os.listdir(G)):
        print(f, os.path.getsize(os.path.join(G, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
