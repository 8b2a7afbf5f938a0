#!/usr/bin/env python3
"""tests/golden/ref_masks.npz: the three 640x480 images the reference ships (masks/mask.png, mask2.png, tum_mask.png; loaded by
Tracking.cc:83-84 as the line extractor's mask), as bit-packed arrays -- the GPU box has neither /root/reference nor a PNG
decoder.  Run in the build container:  python tools/gen_golden_masks.py"""
import os
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for name in ("mask", "mask2", "tum_mask"):
    a = np.array(Image.open("/root/reference/masks/%s.png" % name).convert("L"))
    assert a.shape == (480, 640) and set(np.unique(a)) <= {0, 255}
    out[name] = np.packbits(a > 0)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_masks.npz"), rows=480, cols=640, **out)
print({k: int(np.unpackbits(v).sum()) for k, v in out.items()})
