"""Write the committed mini-datasets the loader / entry-script fixtures run on (TEST INFRASTRUCTURE ONLY).

    python -m oracle.make_mini_dataset

tests/golden/mini_davis/trainval/   a DAVIS-2017-shaped tree (dataset/davis_test_dataset.py:19-47, eval_interactive_davis.py:37-43):
    ImageSets/2017/val.txt                       two sequences
    JPEGImages/480p/<seq>/0000N.jpg              synthetic frames (mivos_amd.util.synthetic.synthetic_clip de-normalised), JPEG q=92
    Annotations/480p/<seq>/0000N.png             palette-index label maps (K = 2 and 1 objects; the VOC colour map)
  frames are 128 x 157 (padded to 128 x 160 = 80 key positions >= top_k = 50: the smallest frame the default network accepts);
  'blackswan' (the name eval_interactive_davis.py:38 reads its palette from) has 5 frames / 2 objects, 'seqb' 4 frames / 1 object.
tests/golden/mini_yv/vos/           a YouTube-VOS-shaped tree (dataset/yv_test_dataset.py:17-35):
    all_frames/valid/JPEGImages/<vid>/0000N.jpg  3 frames of 60 x 96 (loader resizes to 480 x 768)
    valid/Annotations/<vid>/0000N.png            frames 0 and 2 annotated; object 3 appears in frame 2 only
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mivos_amd.util.synthetic import synthetic_clip  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
MEAN = np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]
STD = np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]


def voc_palette():
    """The PASCAL-VOC colour map DAVIS annotations carry (bit-reversal of the label index)."""
    pal = []
    for i in range(256):
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal += [r, g, b]
    return pal


def write_clip(img_dir, ann_dir, t, h, w, k, seed, annotated=None, relabel=None):
    images, gt = synthetic_clip(t, h, w, k, seed)
    os.makedirs(img_dir, exist_ok=True)
    os.makedirs(ann_dir, exist_ok=True)
    for i in range(t):
        rgb = np.clip(np.round((images[0, i].numpy() * STD + MEAN) * 255), 0, 255).astype(np.uint8).transpose(1, 2, 0)
        Image.fromarray(rgb).save(os.path.join(img_dir, f"{i:05d}.jpg"), quality=92)
        if annotated is None or i in annotated:
            lab = gt[i, :, 0].argmax(0).numpy().astype(np.uint8)
            if relabel is not None:
                lab = relabel(i, lab)
            im = Image.fromarray(lab, mode="P")
            im.putpalette(voc_palette())
            im.save(os.path.join(ann_dir, f"{i:05d}.png"))


def main():
    dv = os.path.join(G, "mini_davis", "trainval")
    os.makedirs(os.path.join(dv, "ImageSets", "2017"), exist_ok=True)
    with open(os.path.join(dv, "ImageSets", "2017", "val.txt"), "w") as f:
        f.write("blackswan\nseqb\n")
    write_clip(os.path.join(dv, "JPEGImages", "480p", "blackswan"), os.path.join(dv, "Annotations", "480p", "blackswan"), 5, 128, 157, 2, seed=501)
    write_clip(os.path.join(dv, "JPEGImages", "480p", "seqb"), os.path.join(dv, "Annotations", "480p", "seqb"), 4, 128, 157, 1, seed=502)
    yv = os.path.join(G, "mini_yv", "vos")

    def relabel(i, lab):                     # YouTube-VOS ids are arbitrary: objects 1, 2 -> 2, 5; a third object (id 3) only in frame 2
        out = np.zeros_like(lab)
        out[lab == 1], out[lab == 2] = 2, 5
        if i == 2:
            out[4:14, 6:20] = 3
        return out
    write_clip(os.path.join(yv, "all_frames", "valid", "JPEGImages", "vid0"), os.path.join(yv, "valid", "Annotations", "vid0"), 3, 60, 96, 2, seed=503,
               annotated=(0, 2), relabel=relabel)
    n = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(G, "mini_davis")) for f in fs)
    n2 = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(G, "mini_yv")) for f in fs)
    print("mini_davis", n, "bytes; mini_yv", n2, "bytes")


if __name__ == "__main__":
    main()
