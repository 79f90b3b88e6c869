"""`dataset/onehot_util.py` of the reference: label maps uint8 [T,H,W] -> one-hot uint8 [K,T,H,W]."""
import numpy as np


def all_to_onehot(masks, labels):
    out = np.zeros((len(labels),) + tuple(masks.shape), dtype=np.uint8)
    for k, l in enumerate(labels):
        out[k] = masks == l
    return out
