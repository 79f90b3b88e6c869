"""Multi-object soft aggregation on MI355X — interface of the reference's `model/aggregate.py`."""
import torch

from .. import ops


def aggregate_wbg(prob, keep_bg=False, hard=False):
    """prob [K,1,H,W] -> [K(+1),1,H,W]; background = prod(1 - p) (aggregate.py:22-37)."""
    return ops.aggregate(prob, keep_bg=keep_bg, hard=hard, soft_bg=True)


def aggregate_sbg(prob, keep_bg=False, hard=False):
    """Same with the background probability fixed at 0.5 (aggregate.py:4-20)."""
    return ops.aggregate(prob, keep_bg=keep_bg, hard=hard, soft_bg=False)


def aggregate_wbg_channel(prob, keep_bg=False, hard=False):
    """prob [B,K,H,W] -> (logits [B,K+1,H,W], softmax) (aggregate.py:39-53; training-time helper).
    The softmax runs on the HIP aggregate kernel per batch element; the logits are recovered from
    it (softmax of logits is invariant to the shift, the caller's cross-entropy too)."""
    B = prob.shape[0]
    soft = torch.stack([ops.aggregate(prob[b].unsqueeze(1), keep_bg=True, hard=hard).squeeze(1) for b in range(B)], 0)
    new_prob = torch.cat([torch.prod(1 - prob, dim=1, keepdim=True), prob], 1).clamp(1e-7, 1 - 1e-7)
    logits = torch.log(new_prob / (1 - new_prob))
    if hard:
        logits = logits * 1000
    return (logits, soft) if keep_bg else (logits, soft[:, 1:])
