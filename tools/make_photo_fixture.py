#!/usr/bin/env python3
"""CONTAINER-ONLY: decodes one pair of the reference's own sample images (/root/reference/images/amir1.jpg, amir2.jpg: 720 x 405 photographs,
the pair its README morphs) and stores the pixels as a fixture, tests/golden/photo_pair_720x405.npz (BGR u8, the channel order cv::imread
hands poppy::morph).  Data only — inputs for timing the path on non-synthetic content (bench.py: content_sensitivity) and for one library-vs-oracle
frame test; bench.py upscales it to 1080p with poppy_amd/synth.py: upscale_bgr (integer bilinear, defined there so the GPU box reproduces it).
Decoder: Pillow 12.2 (libjpeg-turbo); the decoded bytes' sha256 go into the file so that a different decoder is noticed."""
import hashlib, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/images"
out = {}
for key, name in (("a", "amir1.jpg"), ("b", "amir2.jpg")):
    rgb = np.asarray(Image.open(os.path.join(src, name)).convert("RGB"), dtype=np.uint8)
    out[key] = np.ascontiguousarray(rgb[:, :, ::-1])
    out[key + "_sha256"] = np.frombuffer(hashlib.sha256(out[key].tobytes()).digest(), dtype=np.uint8)
out["provenance"] = np.array("kallaballa/Poppy images/amir1.jpg + amir2.jpg decoded with Pillow %s, RGB -> BGR" % Image.__version__)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "photo_pair_720x405.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
