"""Golden vectors for the input side generated with Pillow itself -- the library the reference's loader calls
(vo/dataset/common.py:38-44: `img.resize((W, H), Image.BILINEAR)`):

    python tests/golden/make_golden_pipeline.py

Small random uint8 images and their PIL bilinear resizes (shrinking = antialiased triangle filter, enlarging, one axis only)."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [((48, 64), (30, 40)), ((30, 40), (48, 64)), ((50, 70), (50, 35)), ((33, 47), (96, 47)), ((120, 160), (48, 64)),
         ((37, 53), (19, 101))]

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    rec = {"pil_version": np.array(Image.__version__)}
    for i, ((h, w), (oh, ow)) in enumerate(CASES):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        out = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
        rec["in%d" % i], rec["out%d" % i] = img, out
    np.savez_compressed(os.path.join(HERE, "pil_resize_bilinear.npz"), **rec)
    print("pil_resize_bilinear.npz", os.path.getsize(os.path.join(HERE, "pil_resize_bilinear.npz")) // 1024, "KiB, Pillow", Image.__version__)
