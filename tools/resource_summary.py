#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log) per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
K = {"v": r"VGPRs", "a": r"AGPRs", "s": r"SGPRs", "scr": r"ScratchSize \[bytes/lane\]", "occ": r"Occupancy \[waves/SIMD\]", "lds": r"LDS Size \[bytes/block\]"}
for b in blocks:
    name = b.split()[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn)[:60]
    vals = {}
    for k, pat in K.items():
        m = re.search(pat + r': (\d+)', b)
        vals[k] = m.group(1) if m else '?'
    print("%-60s VGPR %4s AGPR %3s SGPR %3s scratch %5s occ %s LDS %s" % (dn, vals['v'], vals['a'], vals['s'], vals['scr'], vals['occ'], vals['lds']))
