"""Import the UNMODIFIED reference (`/root/reference`) on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/make_golden.py`` (this container) to pin
the CPU restatement in ``oracle/stm_oracle.py`` against the real reference code and to
generate the committed fixtures under ``tests/golden/``.  `/root/reference` does not
exist on the GPU box, so nothing that runs there may import this module's result.

Three gaps are shimmed:
  1. torchvision is absent  -> ``oracle/ref_shim/torchvision`` (ResNet-50 architecture only)
  2. cv2 is absent          -> ``oracle/ref_shim/cv2.py`` (imported, never used by the path)
  3. no network             -> ``torch.utils.model_zoo.load_url`` returns ``{}``
"""
import contextlib
import importlib
import io
import os
import sys

REFERENCE_ROOT = os.environ.get("MIVOS_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shim")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "inference_core.py"))


def load_reference():
    """Returns a dict of the reference modules on the hot path."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for p in (REFERENCE_ROOT, _SHIM):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)          # shim first, then the reference root
    import torch.utils.model_zoo as model_zoo
    model_zoo.load_url = lambda *a, **k: {}
    names = dict(prop_net="model.propagation.prop_net", modules="model.propagation.modules",
                 fusion_net="model.fusion_net", aggregate="model.aggregate",
                 attn_network="model.attn_network", tensor_util="util.tensor_util",
                 inference_core="inference_core")
    return {k: importlib.import_module(v) for k, v in names.items()}


def build_reference_networks(top_k=50):
    """Instantiate PropagationNetwork / FusionNet quietly (the constructor prints
    hundreds of 'Not OK' lines because the model zoo is patched out)."""
    ref = load_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        prop = ref["prop_net"].PropagationNetwork(top_k=top_k).eval()
        fuse = ref["fusion_net"].FusionNet().eval()
    return ref, prop, fuse
