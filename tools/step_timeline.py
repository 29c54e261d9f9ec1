#!/usr/bin/env python3
"""Kernel timeline of the LAST step (one MSM + one NTT) in a rocprofv3 --kernel-trace csv of `bench.py --inflight 1`."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_msm_hist' in r['Kernel_Name']]
start = idx[-1] - 1
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f %8.1f  %-45s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'].split('(')[0][-45:], r['Grid_Size_X']))
