"""mivos_amd — MI355X-native space-time-memory mask propagation + difference-aware fusion engine.

Drop-in for the hot path of hkchengrex/MiVOS (see INTEGRATION.md / ``mivos_amd.dropin``):

    from mivos_amd.inference_core import InferenceCore
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    from mivos_amd.model.fusion_net import FusionNet

All arithmetic runs in libmivos_hip.so (hand-written HIP for gfx950, C ABI in include/mivos_hip.h);
there is no CPU or stock-PyTorch fallback.
"""
__version__ = "0.1.0"
