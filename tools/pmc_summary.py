#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).
usage: tools/pmc_summary.py gpurun_out/pmc_*/  [--json profiles/pmc_traffic.json --log-n 20 --curve pallas]"""
import csv, glob, json, os, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
agg = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, cname), v in per_dispatch.items():
            agg[names[disp]][cname].append(v)
out = {}
for k in sorted(agg, key=lambda k: -sum(agg[k].get("SQ_WAVE_CYCLES", [0]))):
    short = k.split("(")[0][:48]
    vals = {c: sum(v) / len(v) for c, v in agg[k].items()}
    out[short] = vals
    print("%-48s " % short + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(vals.items())))
if "--json" in sys.argv:
    path = sys.argv[sys.argv.index("--json") + 1]
    log_n = int(sys.argv[sys.argv.index("--log-n") + 1]); curve = sys.argv[sys.argv.index("--curve") + 1]
    acc = next((v for k, v in out.items() if "k_msm_accum0" in k), {})
    # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB... rocprofv3 reports them in kilobytes; on gfx950
    # FETCH_SIZE under-reports wide coalesced streaming reads by 2x; this kernel's reads are 64-B gathers
    # (4 x dwordx4 per lane) plus 4-B index reads, so the raw value is kept and the caveat recorded.
    rec = {"log_n": log_n, "curve": curve, "kernel": "k_msm_accum0",
           "FETCH_SIZE_KB": acc.get("FETCH_SIZE"), "WRITE_SIZE_KB": acc.get("WRITE_SIZE"),
           "msm_accumulate_hbm_bytes_per_launch": None if acc.get("FETCH_SIZE") is None or acc.get("WRITE_SIZE") is None else int((acc["FETCH_SIZE"] + acc["WRITE_SIZE"]) * 1024),
           "note": "sum of FETCH_SIZE and WRITE_SIZE (separate --pmc passes), x1024 B; gfx950 FETCH_SIZE may under-count 16-B-per-lane streaming reads by up to 2x (uncalibrated for this 64-B gather pattern)"}
    json.dump(rec, open(path, "w"), indent=1)
    print("wrote", path)
