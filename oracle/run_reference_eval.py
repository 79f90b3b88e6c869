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


def write_saves(d):
    os.makedirs(d, exist_ok=True)
    torch.save(Wt.make_prop_state(0), os.path.join(d, "propagation_model.pth"))
    torch.save(Wt.make_fuse_state(0), os.path.join(d, "fusion.pth"))
    torch.save(Wt.make_s2m_state(0), os.path.join(d, "s2m.pth"))
    return d


def script_argv(script, saves, out):
    return [script, "--prop_model", os.path.join(saves, "propagation_model.pth"), "--fusion_model", os.path.join(saves, "fusion.pth"),
            "--s2m_model", os.path.join(saves, "s2m.pth"), "--davis", MINI, "--output", out, "--save_mask"]


def main():
    ref_loader.load_reference()                      # sys.path (shim first, reference root), model_zoo patch
    if not hasattr(np, "bool"):
        np.bool = bool
    torch.set_num_threads(8)
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
    shutil.rmtree(GOLDEN, ignore_errors=True)
    shutil.copytree(out, GOLDEN)
    shutil.copy(os.environ["MIVOS_STUB_LOG"], os.path.join(GOLDEN, "interaction_log.npz"))
    n = 0
    for dp, _, fs in os.walk(GOLDEN):
        for f in fs:
            n += os.path.getsize(os.path.join(dp, f))
            print(os.path.relpath(os.path.join(dp, f), GOLDEN))
    print("golden bytes", n)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
