"""Multi-object soft aggregation on MI355X — interface of the reference's `model/aggregate.py`."""
from .. import ops


def aggregate_wbg(prob, keep_bg=False, hard=False):
    """prob [K,1,H,W] -> [K(+1),1,H,W]; background = prod(1 - p) (aggregate.py:22-37)."""
    with ops.on_device(prob):
        return ops.aggregate(prob, keep_bg=keep_bg, hard=hard, soft_bg=True)


def aggregate_sbg(prob, keep_bg=False, hard=False):
    """Same with the background probability fixed at 0.5 (aggregate.py:4-20)."""
    with ops.on_device(prob):
        return ops.aggregate(prob, keep_bg=keep_bg, hard=hard, soft_bg=False)


def aggregate_wbg_channel(prob, keep_bg=False, hard=False):
    """prob [B,K,H,W] -> (logits [B,K+1,H,W], softmax [B,K(+1),H,W]) (aggregate.py:39-53; FusionNet training helper):
    one launch of the same HIP aggregate kernel, which also writes the logits."""
    with ops.on_device(prob):
        return ops.aggregate_channel(prob, keep_bg=keep_bg, hard=hard)
