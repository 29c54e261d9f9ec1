#!/usr/bin/env python3
"""Concurrency of the MSM + NTT step bench from a rocprofv3 --kernel-trace csv of `bench.py --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0`:
window = from the start of the (skip+1)-th k_msm_accum0 to the end of the last one.   usage: tools/step_concurrency.py kernel_trace.csv [skip=4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc = [r for r in rows if 'k_msm_accum0' in r['Kernel_Name']]
t0, t1 = int(acc[skip]['Start_Timestamp']), int(acc[-1]['End_Timestamp'])
win = [r for r in rows if int(r['Start_Timestamp']) >= t0 and int(r['End_Timestamp']) <= t1]
steps = len(acc) - skip
W = t1 - t0
print("window %.3f ms, %d steps: %.3f ms per step" % (W / 1e6, steps, W / 1e6 / steps))
ev = []
for r in win:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    ev.append((int(r['Start_Timestamp']), 1, nm)); ev.append((int(r['End_Timestamp']), -1, nm))
ev.sort()
active, alone, level, tot = {}, {}, {}, {}
prev = None
for t, d, nm in ev:
    if prev is not None and t > prev:
        n_act = sum(active.values())
        level[n_act] = level.get(n_act, 0) + (t - prev)
        for k, v in active.items():
            if v: tot[k] = tot.get(k, 0) + (t - prev) * v
        if n_act == 1:
            k = next(k for k, v in active.items() if v)
            alone[k] = alone.get(k, 0) + (t - prev)
    active[nm] = active.get(nm, 0) + d
    prev = t
print("time by number of kernels running: " + ", ".join("%d: %.1f%%" % (k, 100.0 * v / W) for k, v in sorted(level.items())))
print("%-42s %10s %10s   (ms per step)" % ("kernel", "total", "alone"))
for nm, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-40s %10.3f %10.3f" % (nm, v / 1e6 / steps, alone.get(nm, 0) / 1e6 / steps))
