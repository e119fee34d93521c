#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (build/asm/*.usage.txt)."""
import re, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    for blk in re.split(r'Function Name: ', txt)[1:]:
        name = blk.split()[0]
        def f(k):
            m = re.search(k + r': (\d+)', blk)
            return m.group(1) if m else '?'
        print("%-44s VGPR %3s AGPR %3s SGPR %3s scratch %3s occ %s LDS %s" % (
            name[:44], f('VGPRs'), f('AGPRs'), f('TotalSGPRs'), f(r'ScratchSize \[bytes/lane\]'),
            f(r'Occupancy \[waves/SIMD\]'), f(r'LDS Size \[bytes/block\]')))
