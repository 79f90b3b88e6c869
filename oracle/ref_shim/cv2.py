"""Test-infrastructure stand-in for OpenCV (not installed in this image; NOT shipped, NOT used by the product path).

`/root/reference/inference_core.py:9` and `eval_interactive_davis.py:9` import cv2 without using it; `davis_processor.py:54-60`
calls ``cv2.dilate(uint8 image, np.ones((3, 3)))`` on the scribble planes.  ``dilate`` below restates that one call: grey-level
dilation with a rectangular kernel of ones anchored at its centre, border pixels see only the inside (OpenCV's default border
value for dilation is -inf, i.e. the border never wins)."""
import numpy as np


def dilate(src, kernel, iterations=1):
    k = np.asarray(kernel)
    assert k.ndim == 2 and k.shape[0] % 2 == 1 and k.shape[1] % 2 == 1 and bool((k != 0).all()), "stub: odd rectangular kernels of ones only"
    ry, rx = k.shape[0] // 2, k.shape[1] // 2
    out = np.asarray(src)
    for _ in range(iterations):
        h, w = out.shape[:2]
        pad = np.full((h + 2 * ry, w + 2 * rx) + out.shape[2:], np.iinfo(out.dtype).min if out.dtype.kind in "iu" else -np.inf, dtype=out.dtype)
        pad[ry:ry + h, rx:rx + w] = out
        acc = out.copy()
        for dy in range(2 * ry + 1):
            for dx in range(2 * rx + 1):
                acc = np.maximum(acc, pad[dy:dy + h, dx:dx + w])
        out = acc
    return out
