"""The reference's entry script end to end ("drops into eval_interactive_davis.py unchanged").

Golden: `tests/golden/eval_davis/` = what `/root/reference/eval_interactive_davis.py`, UNCHANGED, wrote on PyTorch-CPU against the
unmodified reference modules (oracle/run_reference_eval.py): the mini-DAVIS tree, the scripted `davisinteractive` stand-in
(4 samples x 8 interactions = 12 propagating interactions with fusion + 20 update_mask_only calls, S2M on every one), the
synthetic `saves/*.pth`.  The S2M checkpoint of this fixture is a hand-set "scribble follower" (run_reference_eval.eval_s2m_state):
with the random S2M of the other tests the session is chaotic - its masks are noise, and the unmodified reference run with 1 CPU
thread instead of 8 ends at IoU 0.39 against ITSELF - whereas with object-shaped masks the reference's two runs agree to IoU
0.99987 (15 pixels over all 32 submissions; test_reference_fixture_is_well_conditioned), so a second implementation can be held
to the north star's bar.  Saved: the palette PNGs of three samples (the script never writes the last one), `summary.json`, and
every array the script handed to `sess.submit_masks` (interaction_log.npz).

On the GPU the same session runs on the engine, twice over where possible:
  * through the UNCHANGED script (`python -m mivos_amd.dropin <reference>/eval_interactive_davis.py ...`) when the reference tree
    is present (it is not on the gpurun box: a Python reference cannot travel);
  * through `mivos_amd.eval_davis.run_interactive_davis`, the engine's own statement of the script's loop - everywhere.
Bar: every submitted mask array within IoU >= 0.999 (mean over objects) of the reference's, identical PNG file set."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "oracle", "ref_shim")
GOLD = os.path.join(ROOT, "tests", "golden", "eval_davis")
MINI = os.path.join(ROOT, "tests", "golden", "mini_davis")
REFERENCE = os.environ.get("MIVOS_REFERENCE_ROOT", "/root/reference")


def _iou(a, b, k):
    return (np.logical_and(a == k, b == k).sum() + 1e-6) / (np.logical_or(a == k, b == k).sum() + 1e-6)


def _compare_logs(log_path, tag):
    with np.load(os.path.join(GOLD, "interaction_log.npz")) as z:
        gold = {k: z[k] for k in z.files}
    with np.load(log_path) as z:
        got = {k: z[k] for k in z.files}
    assert sorted(got) == sorted(gold) and len(gold) == 32                 # same samples, interaction counters and FRAMES (the keys carry them)
    worst, rows = 1.0, []
    for key in sorted(gold):
        a, b = got[key], gold[key]
        assert a.shape == b.shape and a.dtype == np.uint8
        ks = [int(v) for v in np.unique(b) if v != 0]
        iou = float(np.mean([_iou(a, b, k) for k in ks])) if ks else 1.0
        rows.append((key, iou, int((a != b).sum())))
        worst = min(worst, iou)
    bad = [r for r in rows if r[1] < 0.999]
    print(f"{tag}: 32 submissions, worst mean-object IoU {worst:.6f}, mismatching px total {sum(r[2] for r in rows)} of {sum(v.size for v in gold.values())}")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "entry_script_parity.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=tag, worst_iou=worst, rows=rows)) + "\n")
    except OSError:
        pass
    assert not bad, bad


def _compare_pngs(out_dir):
    from PIL import Image
    want = sorted(os.path.relpath(os.path.join(d, f), GOLD) for d, _, fs in os.walk(GOLD) for f in fs if f.endswith(".png"))
    have = sorted(os.path.relpath(os.path.join(d, f), out_dir) for d, _, fs in os.walk(out_dir) for f in fs if f.endswith(".png"))
    assert have == want and len(want) == 14                                 # 0/blackswan, 1/blackswan, 0/seqb - never 1/seqb (the script's rule)
    for rel in want:
        a, b = Image.open(os.path.join(out_dir, rel)), Image.open(os.path.join(GOLD, rel))
        assert a.mode == b.mode == "P" and a.getpalette()[:768] == b.getpalette()[:768]
        x, y = np.array(a), np.array(b)
        ks = [int(v) for v in np.unique(y) if v != 0]
        assert x.shape == y.shape and all(_iou(x, y, k) >= 0.999 for k in ks), rel
    s, g = json.load(open(os.path.join(out_dir, "summary.json"))), json.load(open(os.path.join(GOLD, "summary.json")))
    assert s["submissions"] == g["submissions"] == 32 and abs(s["mean_J"] - g["mean_J"]) < 1e-3


def _saves(tmp):
    from oracle.run_reference_eval import write_saves
    return write_saves(os.path.join(str(tmp), "saves"))


def test_golden_is_complete_and_the_stand_ins_load():
    """CPU: the committed golden has the script's file layout; the scripted session replays the same (sample, interaction, frame)
    sequence the golden's log keys carry, whatever masks are submitted."""
    sys.path.insert(0, SHIM)
    try:
        from davisinteractive.session.session import DavisInteractiveSession
        from davisinteractive.utils.scribbles import scribbles2mask
        with np.load(os.path.join(GOLD, "interaction_log.npz")) as z:
            keys = sorted(z.files)
        seen = []
        with DavisInteractiveSession(davis_root=os.path.join(MINI, "trainval"), max_nb_interactions=8) as sess:
            n = 0
            while sess.next():
                seq, scr, new = sess.get_scribbles(only_last=True)
                f = [i for i, s in enumerate(scr["scribbles"]) if s][0]
                m = scribbles2mask(dict(scr, scribbles=[scr["scribbles"][f]]), (128, 160))[0]
                assert m.shape == (128, 160) and set(np.unique(m)) <= {-1, 0, 1, 2} and (m >= 0).sum() > 50
                user, i = sess.samples[sess.sample_i][1], sess.inter_i
                seen.append(f"sub_{n:03d}_{seq}_u{user}_n{i}_f{f}")
                gt = sess._annotations(seq)
                sess.submit_masks(np.zeros_like(gt), [f] if i not in (2, 5, 7) else None)      # DAVISProcessor's schedule [2, 5, 7]
                n += 1
        assert seen == keys
    finally:
        sys.path.remove(SHIM)
    assert os.path.isfile(os.path.join(GOLD, "summary.json")) and not os.path.isdir(os.path.join(GOLD, "1", "seqb"))


@pytest.mark.gpu
def test_engine_eval_loop_matches_the_reference_script_output(tmp_path):
    """mivos_amd.eval_davis.run_interactive_davis (the script's loop on the engine, clips ingested on the GPU) against the golden."""
    sys.path.insert(0, SHIM)
    os.environ["MIVOS_STUB_LOG"] = str(tmp_path / "log.npz")
    try:
        from mivos_amd.eval_davis import run_interactive_davis
        saves, out = _saves(tmp_path), str(tmp_path / "out")
        run_interactive_davis(MINI, out, os.path.join(saves, "propagation_model.pth"), os.path.join(saves, "fusion.pth"), os.path.join(saves, "s2m.pth"),
                              save_mask=True, device="cuda:0", log=lambda *a: None)
    finally:
        sys.path.remove(SHIM)
        os.environ.pop("MIVOS_STUB_LOG", None)
    _compare_logs(str(tmp_path / "log.npz"), "eval_davis_loop")
    _compare_pngs(out)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "eval_interactive_davis.py")), reason="the reference tree is not on this machine")
def test_unchanged_reference_script_runs_on_the_engine(tmp_path):
    """`python -m mivos_amd.dropin /root/reference/eval_interactive_davis.py ...`: the reference's file, byte for byte, with every
    hot-path import resolving to the engine (needs the reference tree AND a GPU)."""
    saves, out = _saves(tmp_path), str(tmp_path / "out")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SHIM, ROOT, os.environ.get("PYTHONPATH", "")]), MIVOS_STUB_LOG=str(tmp_path / "log.npz"))
    cmd = [sys.executable, "-m", "mivos_amd.dropin", os.path.join(REFERENCE, "eval_interactive_davis.py"), "--prop_model", os.path.join(saves, "propagation_model.pth"),
           "--fusion_model", os.path.join(saves, "fusion.pth"), "--s2m_model", os.path.join(saves, "s2m.pth"), "--davis", MINI, "--output", out, "--save_mask"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    _compare_logs(str(tmp_path / "log.npz"), "unchanged_script_via_dropin")
    _compare_pngs(out)


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "eval_interactive_davis.py")), reason="the reference tree is not on this machine")
@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only plumbing check (the GPU run is test_unchanged_reference_script_runs_on_the_engine)")
def test_unchanged_reference_script_reaches_the_engine_without_a_gpu(tmp_path):
    """Where the reference tree exists but no GPU does (this container): the unchanged script, through dropin, imports the engine's
    classes, loads both mini-DAVIS sequences through the engine's DAVISTestDataset inside its DataLoader(num_workers=2), loads the
    checkpoints - and stops at `PropagationNetwork().cuda()` (eval_interactive_davis.py:58), because the engine has no CPU path."""
    saves = _saves(tmp_path)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SHIM, ROOT, os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "mivos_amd.dropin", os.path.join(REFERENCE, "eval_interactive_davis.py"), "--prop_model", os.path.join(saves, "propagation_model.pth"),
           "--fusion_model", os.path.join(saves, "fusion.pth"), "--s2m_model", os.path.join(saves, "s2m.pth"), "--davis", MINI, "--output", str(tmp_path / "out"), "--save_mask"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert "Finished loading 2 sequences." in r.stdout and r.returncode != 0
    tail = r.stderr.strip().splitlines()[-1]
    assert "eval_interactive_davis.py" in r.stderr and ("cuda" in r.stderr.lower() or "hip" in r.stderr.lower()), tail


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "eval_interactive_davis.py")), reason="the reference tree is not on this machine")
def test_reference_fixture_is_well_conditioned(tmp_path):
    """The golden is only a fair target if the reference reproduces it under a rounding-level perturbation of ITSELF: the unchanged
    script on the unmodified reference with 1 CPU thread (different summation order in every convolution) against the committed
    8-thread golden - the same bar the engine is held to."""
    env = dict(os.environ, MIVOS_EVAL_THREADS="1", MIVOS_EVAL_PROBE_DIR=str(tmp_path / "probe"))
    r = subprocess.run([sys.executable, "-m", "oracle.run_reference_eval"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _compare_logs(str(tmp_path / "probe" / "interaction_log.npz"), "reference_1_thread_vs_golden")
    _compare_pngs(str(tmp_path / "probe"))
