"""Run the reference's own entry scripts on top of the MI355X engine, unchanged.

    python -m mivos_amd.dropin /path/to/MiVOS/eval_interactive_davis.py --davis ... --output ...
    python -m mivos_amd.dropin /path/to/MiVOS/interactive_gui.py --images ...

``install()`` registers this package's modules under the import names the reference's scripts use
(`eval_interactive_davis.py:11-15`, `interactive_gui.py:29-35`, `davis_processor.py:7-9`):

    inference_core, davis_processor, model.propagation.prop_net, model.propagation.modules, model.fusion_net,
    model.aggregate, model.attn_network, model.s2m.s2m_network, util.tensor_util,
    generation.fusion_generator (generate_fusion.py:16), model.fusion_model (train.py:14),
    dataset.davis_test_dataset / dataset.yv_test_dataset / dataset.range_transform / dataset.onehot_util (the test-time loaders)

Everything else of the reference (``interact``, ``dataset``, ``model.s2s`` ...) keeps resolving to the reference
tree, which is appended to the package search paths of ``model`` / ``util``.
"""
import importlib
import os
import runpy
import sys

ALIASES = {
    "inference_core": "mivos_amd.inference_core",
    "model.propagation.prop_net": "mivos_amd.model.propagation.prop_net",
    "model.propagation.modules": "mivos_amd.model.propagation.modules",
    "model.fusion_net": "mivos_amd.model.fusion_net",
    "model.aggregate": "mivos_amd.model.aggregate",
    "model.attn_network": "mivos_amd.model.attn_network",
    "model.s2m.s2m_network": "mivos_amd.model.s2m.s2m_network",
    "davis_processor": "mivos_amd.davis_processor",
    "util.tensor_util": "mivos_amd.util.tensor_util",
    "generation.fusion_generator": "mivos_amd.generation.fusion_generator",      # generate_fusion.py:16
    "model.fusion_model": "mivos_amd.model.fusion_model",                        # train.py:14
    "dataset.davis_test_dataset": "mivos_amd.dataset.davis_test_dataset",        # eval_interactive_davis.py:14, generate_fusion.py:16
    "dataset.yv_test_dataset": "mivos_amd.dataset.yv_test_dataset",
    "dataset.range_transform": "mivos_amd.dataset.range_transform",              # interact/interactive_utils.py:15 (no torchvision needed)
    "dataset.onehot_util": "mivos_amd.dataset.onehot_util",
}
PACKAGES = {"model": "mivos_amd.model", "model.propagation": "mivos_amd.model.propagation", "model.s2m": "mivos_amd.model.s2m",
            "util": "mivos_amd.util", "generation": "mivos_amd.generation", "dataset": "mivos_amd.dataset"}


def install(reference_root=None):
    """Alias the engine's modules to the reference's import names.  With ``reference_root`` the
    reference's remaining sub-modules (model/s2m, util/palette ...) stay importable."""
    for ref_name, ours in PACKAGES.items():
        pkg = importlib.import_module(ours)
        if reference_root:
            extra = os.path.join(reference_root, *ref_name.split("."))
            if os.path.isdir(extra) and extra not in pkg.__path__:
                pkg.__path__.append(extra)
        sys.modules[ref_name] = pkg
    for ref_name, ours in ALIASES.items():
        mod = importlib.import_module(ours)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, mod)
    if reference_root and reference_root not in sys.path:
        sys.path.append(reference_root)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    install(os.path.dirname(script))
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
