#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (make -C reinmav-gym_amd asm)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "reinmav-gym_amd/build/resource_usage.txt"
txt = open(path).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]


def g(b, k):
    m = re.search(k + r": (\d+)", b)
    return m.group(1) if m else "?"


for b in blocks:
    name = b.split()[0]
    print("%-58s VGPR %3s AGPR %2s SGPR %3s scratch %3s occ %s LDS %s" % (
        name[:58], g(b, " VGPRs"), g(b, "AGPRs"), g(b, "TotalSGPRs"), g(b, r"ScratchSize \[bytes/lane\]"),
        g(b, r"Occupancy \[waves/SIMD\]"), g(b, r"LDS Size \[bytes/block\]")))
