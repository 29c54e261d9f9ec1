#!/usr/bin/env python3
"""SQ counters per kernel from rocprofv3 --pmc passes (counter_collection.csv + kernel_trace.csv in DIR...): per launch the duration under collection, the
vector-ALU time (SQ_ACTIVE_INST_VALU x 4 cycles / (2.4 GHz x 1024 SIMDs)) and how a wave's cycles split.  The eight counters do not fit one pass: give
two directories (one --pmc pass each); a kernel's launches are matched by order.
usage: tools/sq_summary.py DIR [DIR ...] [--group-by-launch N] [--skip-launches M]   (--group-by-launch 3: launches i, i + 3, ... of k_ntt_pass are reported separately: its passes)"""
import csv, glob, os, sys
from collections import defaultdict

args = sys.argv[1:]
group = 1
if "--group-by-launch" in args:
    i = args.index("--group-by-launch")
    group = int(args[i + 1])
    del args[i:i + 2]
skip = 0
if "--skip-launches" in args:      # drop the first N launches of every kernel (warm-up at ramping clocks)
    i = args.index("--skip-launches")
    skip = int(args[i + 1])
    del args[i:i + 2]
dirs = args
per_kernel = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [value per launch, in launch order]
dur = defaultdict(list)


def short(name):
    return name.split("(")[0].replace("void ", "")


for d in dirs:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        disp, names = defaultdict(float), {}
        for r in csv.DictReader(open(f)):
            did = int(r["Dispatch_Id"])
            names[did] = short(r["Kernel_Name"])
            disp[(did, r["Counter_Name"])] += float(r["Counter_Value"])
        counters = sorted({c for (_, c) in disp})
        for did in sorted(names):
            for c in counters:
                per_kernel[names[did]][c].append(disp.get((did, c), 0.0))
    if not dur:
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[:1]:
            for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

SIMDS, GHZ = 1024, 2.4


def mean(v):
    return sum(v) / max(1, len(v))


rows = []
for k, cs in per_kernel.items():
    if "SQ_ACTIVE_INST_VALU" not in cs:
        continue
    step = group if ("ntt" in k and group > 1) else 1
    for g in range(step):
        sel = lambda v: v[skip:][g::step]
        n = len(sel(cs["SQ_ACTIVE_INST_VALU"]))
        if n == 0:
            continue
        valu = mean(sel(cs["SQ_ACTIVE_INST_VALU"])) * 4 / (GHZ * 1e3 * SIMDS)      # us per launch
        wc = mean(sel(cs.get("SQ_WAVE_CYCLES", [0.0])))
        d = mean(sel(dur.get(k, [0.0])))
        parts = []
        if wc:
            pct = lambda name: 100.0 * mean(sel(cs.get(name, [0.0]))) / wc
            parts.append("of a wave's cycles: issuing VALU %.0f %%" % pct("SQ_ACTIVE_INST_VALU"))
            if "SQ_ACTIVE_INST_LDS" in cs:
                parts.append("LDS %.0f %%" % pct("SQ_ACTIVE_INST_LDS"))
            if "SQ_WAIT_ANY" in cs:
                parts.append("parked at a wait / barrier %.0f %%" % pct("SQ_WAIT_ANY"))
            if "SQ_ACTIVE_INST_ANY" in cs and "SQ_WAIT_ANY" in cs:
                parts.append("ready but not issued %.0f %%" % (100.0 - pct("SQ_ACTIVE_INST_ANY") - pct("SQ_WAIT_ANY")))
        label = k[:44] + (" pass %d" % g if step > 1 else "")
        rows.append((valu * n, "%-52s %3d launches %8.1f us each under collection | VALU time %7.1f us each (%3.0f %% busy) | %s" %
                     (label, n, d, valu, 100.0 * valu / d if d else 0.0, ", ".join(parts))))
print("SQ counters per kernel (rocprofv3 --kernel-trace --pmc, two passes); VALU time = SQ_ACTIVE_INST_VALU x 4 cycles / (2.4 GHz x 1024 SIMDs); durations under counter collection")
for _, line in sorted(rows, reverse=True):
    print(line)
