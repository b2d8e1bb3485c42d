#!/usr/bin/env python3
"""CONTAINER-ONLY: decodes two of the reference's own demo pairs whose widths are NOT multiples of 4 (make_demos.sh:15 `cars`: kindpng1s.png / kindpng2s.png,
749 x 480; make_demos.sh:31 `numbers`: 1st.png / 2nd.png, 639 x 480) and stores the pixels as a fixture, tests/golden/demo_pairs.npz (BGR u8, the channel order
cv::imread hands poppy::morph; the car images' alpha channel is dropped, as imread's default IMREAD_COLOR does).  Data only — inputs of the two reference-run whole-morph
fixtures a_639x480_numbers / a_749x480_cars and of bench.py's odd_width leg.  Decoder: Pillow (libpng); the decoded bytes' sha256 go into the file."""
import hashlib, os
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/images"
out = {}
for key, name in (("numbers_a", "1st.png"), ("numbers_b", "2nd.png"), ("cars_a", "kindpng1s.png"), ("cars_b", "kindpng2s.png")):
    rgb = np.asarray(Image.open(os.path.join(src, name)).convert("RGB"), dtype=np.uint8)
    out[key] = np.ascontiguousarray(rgb[:, :, ::-1])
    out[key + "_sha256"] = np.frombuffer(hashlib.sha256(out[key].tobytes()).digest(), dtype=np.uint8)
out["provenance"] = np.array("kallaballa/Poppy images/{1st,2nd,kindpng1s,kindpng2s}.png decoded with Pillow %s, RGB(A) -> BGR" % Image.__version__)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "demo_pairs.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") and v.ndim == 3 else "") for k, v in out.items()})
