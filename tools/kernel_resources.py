#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage output (build/resource_usage.txt)."""
import re
import sys

txt = open(sys.argv[1] if len(sys.argv) > 1 else "build/resource_usage.txt").read()
blocks = re.split(r"Function Name: ", txt)
keys = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"TotalSGPRs"),
        ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("lds", r"LDS Size \[bytes/block\]")]
for b in blocks[1:]:
    name = b.split()[0]
    vals = []
    for label, k in keys:
        m = re.search(r" " + k + r": (\d+)", b)
        vals.append("%s=%s" % (label, m.group(1) if m else "?"))
    print("%-95s %s" % (name[:95], " ".join(vals)))
