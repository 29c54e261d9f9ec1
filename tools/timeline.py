#!/usr/bin/env python3
"""Kernel timeline of the LAST create_proof in a rocprofv3 --kernel-trace csv (start offset, duration, kernel, queue, grid).
usage: tools/timeline.py kernel_trace.csv [msm_calls_per_proof=6]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_msm_accum0' in r['Kernel_Name']]
start = max(0, idx[-per] - 12)
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f %8.1f  %-45s q%s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'].split('(')[0][-45:], r.get('Queue_Id'), r['Grid_Size_X']))
