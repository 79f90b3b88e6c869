"""Run the reference's OWN entry script, `/root/reference/eval_interactive_davis.py`, UNCHANGED on PyTorch-CPU against the
unmodified reference modules, on the committed mini-DAVIS tree with the scripted `davisinteractive` stand-in, and store what it
wrote as the golden of the end-to-end entry-script test (tests/test_entry_script.py).

TEST INFRASTRUCTURE ONLY (this container: needs /root/reference).

    python -m oracle.run_reference_eval            # -> tests/golden/eval_davis/{<user>/<seq>/0000N.png, interaction_log.npz, summary.json}

Environment shims (none of them touches reference source):
  * oracle/ref_shim on sys.path: torchvision (models + transforms), cv2 (dilate), davisinteractive (scripted session) - packages the
    reference imports and this image lacks;
  * `torch.utils.model_zoo.load_url -> {}` (no network), `numpy.bool = bool` (davis_processor.py:55 uses the alias numpy 1.24
    removed);
  * no GPU here: `nn.Module.cuda` is a no-op and `DAVISProcessor.__init__`'s default `device='cuda:0'` becomes 'cpu' (the script
    hard-codes `.cuda()`, eval_interactive_davis.py:58-68, and passes no device, :82);
  * `saves/*.pth`: the seeded synthetic state dicts (the released checkpoints cannot be downloaded)."""
import os
import runpy
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import weights as Wt  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "eval_davis")
MINI = os.path.join(ROOT, "tests", "golden", "mini_davis")


def eval_s2m_state(gain=6.0, bias=-3.0):
    """The scribble-to-mask weights of the entry-script fixture (shared with tests/test_entry_script.py): a "scribble follower" laid
    out in the reference's 368-key DeepLabV3+ state_dict.

    Why not the seeded random S2M of the other tests: its masks are noise (J = 0.09 against the annotations), and propagating noise
    masks through 8 interactions is chaotic - the UNMODIFIED reference run with 1 instead of 8 CPU threads already disagrees with
    itself (IoU 0.39 by the last interaction, `MIVOS_EVAL_THREADS=1 MIVOS_EVAL_PROBE_DIR=...`), so no second implementation could
    match it.  A trained S2M returns object-shaped masks around the strokes; this state does the same with one hand-set path and
    zeros elsewhere (the network's own arithmetic is pinned on random weights by tests/test_gpu_s2m.py):
      conv1 (7x7 / 2)   ch 0 = box(current mask) * 0.5 + box(positive scribble),  ch 1 = box(negative scribble)        (box = mean over 7 x 7)
      layer1            every residual branch is zero (conv3 = 0): blocks pass relu(identity); the first block's 1x1 projection keeps ch 0, 1
      low-level head    classifier.project keeps ch 0, 1;  ASPP branch silenced (aspp.project = 0)
      classifier.0      3x3 box on ch 0 and ch 1;  classifier.3: logit = gain * (ch 0 - ch 1) + bias
    All BatchNorms are the identity (mean 0, var 1, weight 1, bias 0)."""
    from mivos_amd.util.synthetic import s2m_spec
    sd = {}
    for name, shape in s2m_spec().items():
        if name.endswith("running_var") or (name.endswith(".weight") and len(shape) == 1):
            sd[name] = torch.ones(shape)
        elif name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros(shape, dtype=torch.long)
        else:
            sd[name] = torch.zeros(shape)
    w = sd["backbone.conv1.weight"]
    w[0, 3] = 0.5 / 49.0
    w[0, 4] = 1.0 / 49.0 * 12.0          # a 3-pixel-wide stroke covers ~3/7 of the box: bring it to the scale of a filled mask
    w[1, 5] = 1.0 / 49.0 * 12.0
    for c in (0, 1):
        sd["backbone.layer1.0.downsample.0.weight"][c, c, 0, 0] = 1.0
        sd["classifier.project.0.weight"][c, c, 0, 0] = 1.0
        sd["classifier.classifier.0.weight"][c, c] = 1.0 / 9.0
    sd["classifier.classifier.3.weight"][0, 0, 0, 0] = gain
    sd["classifier.classifier.3.weight"][0, 1, 0, 0] = -gain
    sd["classifier.classifier.3.bias"][0] = bias
    return sd


def write_saves(d):
    os.makedirs(d, exist_ok=True)
    torch.save(Wt.make_prop_state(0), os.path.join(d, "propagation_model.pth"))
    torch.save(Wt.make_fuse_state(0), os.path.join(d, "fusion.pth"))
    torch.save(eval_s2m_state(), os.path.join(d, "s2m.pth"))
    return d


def script_argv(script, saves, out):
    return [script, "--prop_model", os.path.join(saves, "propagation_model.pth"), "--fusion_model", os.path.join(saves, "fusion.pth"),
            "--s2m_model", os.path.join(saves, "s2m.pth"), "--davis", MINI, "--output", out, "--save_mask"]


def main():
    ref_loader.load_reference()                      # sys.path (shim first, reference root), model_zoo patch
    if not hasattr(np, "bool"):
        np.bool = bool
    probe = os.environ.get("MIVOS_EVAL_PROBE_DIR")   # conditioning probe: run with another thread count, keep the golden untouched
    torch.set_num_threads(int(os.environ.get("MIVOS_EVAL_THREADS", "8")))
    torch.nn.Module.cuda = lambda self, device=None: self
    import davis_processor                           # the reference's
    assert davis_processor.__file__.startswith(ref_loader.REFERENCE_ROOT)
    d = list(davis_processor.DAVISProcessor.__init__.__defaults__)
    assert d == ["cuda:0"], d
    davis_processor.DAVISProcessor.__init__.__defaults__ = ("cpu",)
    work = tempfile.mkdtemp(prefix="mivos_ref_eval_")
    out = os.path.join(work, "out")
    saves = write_saves(os.path.join(work, "saves"))
    os.environ["MIVOS_STUB_LOG"] = os.path.join(work, "interaction_log.npz")
    script = os.path.join(ref_loader.REFERENCE_ROOT, "eval_interactive_davis.py")
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = script_argv(script, saves, out)
    os.chdir(work)
    try:
        import contextlib
        import io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):       # hundreds of 'Not OK' lines from mod_resnet.py:29
            runpy.run_path(script, run_name="__main__")
        print("\n".join(l for l in buf.getvalue().splitlines() if "OK" not in l))
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    dest = probe or GOLDEN
    shutil.rmtree(dest, ignore_errors=True)
    shutil.copytree(out, dest)
    shutil.copy(os.environ["MIVOS_STUB_LOG"], os.path.join(dest, "interaction_log.npz"))
    n = 0
    for dp, _, fs in os.walk(dest):
        for f in fs:
            n += os.path.getsize(os.path.join(dp, f))
            pass
    print("golden bytes", n)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
