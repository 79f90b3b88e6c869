"""scribbles2mask of the stand-in `davisinteractive` (see the package docstring).  Interface of the framework's function:
scribbles dict {'sequence': str, 'scribbles': [per frame: [{'path': [[x, y], ...] in [0, 1], 'object_id': int, ...}]]},
output_resolution (H, W) -> int array [n_frames, H, W], `default_value` (-1) where no scribble passes, the object id along the
scribble lines (points scaled to pixel coordinates by (W - 1, H - 1) and TRUNCATED to integers like the framework's
`path.astype(int)`, consecutive points joined by Bresenham segments).  A stand-in written from the framework's documented behaviour, not
a byte-exact copy: the entry-script goldens pin the reference's code given THIS rasteriser on both sides."""
import numpy as np


def _bresenham(x0, y0, x1, y1):
    dx, dy = abs(x1 - x0), -abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    err = dx + dy
    pts = []
    while True:
        pts.append((x0, y0))
        if x0 == x1 and y0 == y1:
            return pts
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x0 += sx
        if e2 <= dx:
            err += dx
            y0 += sy


def scribbles2mask(scribbles, output_resolution, bresenham=True, default_value=-1):
    h, w = output_resolution
    frames = scribbles["scribbles"]
    masks = np.full((len(frames), h, w), default_value, dtype=np.int64)
    for f, lines in enumerate(frames):
        for line in lines:
            path = np.asarray(line["path"], dtype=np.float64)
            px = np.clip((path * np.array([w - 1, h - 1])).astype(np.int64), 0, [w - 1, h - 1])
            pts = [tuple(px[0])]
            for a, b in zip(px[:-1], px[1:]):
                pts += _bresenham(int(a[0]), int(a[1]), int(b[0]), int(b[1]))[1:] if bresenham else [tuple(b)]
            for x, y in pts:
                masks[f, y, x] = line["object_id"]
    return masks
