#!/usr/bin/env python3
"""Regenerates orb_pattern.inc (the 256 published rBRIEF test pairs) from the reference's table.
Container-only helper: reads OCV/features2d/src/orb.cpp:380-638, writes the numbers, nothing else."""
import re
import sys

src = open("/root/reference/third/opencv-4.6.0/modules/features2d/src/orb.cpp").read()
i = src.index("static int bit_pattern_31_[256*4]")
body = re.sub(r"/\*.*?\*/", "", src[src.index("{", i) + 1:src.index("};", i)], flags=re.S)
nums = [int(x) for x in re.findall(r"-?\d+", body)]
assert len(nums) == 1024
for path in sys.argv[1:]:
    head = open(path).read().split("\n")[:3]
    with open(path, "w") as f:
        f.write("\n".join(head) + "\n")
        for r in range(0, 1024, 32):
            f.write(",".join(str(v) for v in nums[r:r + 32]) + ",\n")
