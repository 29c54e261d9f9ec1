#!/usr/bin/env python3
"""k_msm_accum0's launch durations from a rocprofv3 kernel trace of `bench.py --inflight 1` (preheat, warm-up and timed steps are the same launches one at a
time): the average over all launches (what --stats reports), over the LAST `steps` launches (the timed region, what bench.py's HIP events bracket), and by
blocks of 100 launches (the clock state over the run).   python tools/accum0_launches.py <kernel_trace.csv> [steps=20]
Any other kernel:  python tools/accum0_launches.py <kernel_trace.csv> <name substring> [last=20]"""
import csv, sys
name = "k_msm_accum0"
rest = sys.argv[2:]
if rest and not rest[0].isdigit():
    name, rest = rest[0], rest[1:]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if name in r["Kernel_Name"]]
steps = int(rest[0]) if rest else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(name + ": %d launches, average %.4f ms (min %.4f, max %.4f)" % (len(d), sum(d) / len(d), min(d), max(d)))
print("the last %d launches (the timed region): average %.4f ms" % (steps, sum(d[-steps:]) / steps))
for i in range(0, len(d), 100):
    blk = d[i:i + 100]
    print("launches %4d .. %4d: average %.4f ms" % (i, i + len(blk) - 1, sum(blk) / len(blk)))
