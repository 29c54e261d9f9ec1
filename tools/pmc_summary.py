#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).
usage: tools/pmc_summary.py DIR...  [--json profiles/pmc_traffic.json --log-n 20 --curve pallas --window-bits 16 --kernel-rev r02a --calib DIR]

HBM traffic of k_msm_accum0 as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE from SEPARATE --pmc passes
(they do not fit one pass), in KiB; WRITE_SIZE is exact for 16-B-per-lane stores; FETCH_SIZE is exactly 1/2 for wide coalesced
streaming reads and UNCALIBRATED for other widths -- so the gather pattern of this kernel (one 64-B record per lane through four
dwordx4 loads) is calibrated on tools/pmc_calib's k_gather64, whose byte count is known: factor = FETCH_SIZE * 1024 / known bytes,
and the kernel's fetch traffic is reported as FETCH_SIZE * 1024 / factor."""
import csv, glob, json, os, sys
from collections import defaultdict

def collect(dirs):
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
    for k in agg:
        out[k.split("(")[0][:60]] = {c: sum(v) / len(v) for c, v in agg[k].items()}
    return out

def arg(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default

opts = {"--json", "--log-n", "--curve", "--window-bits", "--kernel-rev", "--calib"}
dirs, skip = [], False
for a in sys.argv[1:]:
    if skip: skip = False; continue
    if a in opts: skip = True; continue
    dirs.append(a)
out = collect(dirs)
for k, vals in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    print("%-60s " % k + "  ".join("%s=%.6g" % (c, v) for c, v in sorted(vals.items())))
if "--json" in sys.argv:
    acc = next((v for k, v in out.items() if "k_msm_accum0" in k), {})
    calib = collect([arg("--calib")]) if arg("--calib") else {}
    g = next((v for k, v in calib.items() if "k_gather64" in k), {})
    st = next((v for k, v in calib.items() if "k_stream16" in k), {})
    n_rec = 1 << 24
    gather_factor = g["FETCH_SIZE"] * 1024 / (n_rec * 68) if "FETCH_SIZE" in g else None
    stream_factor = st["FETCH_SIZE"] * 1024 / (n_rec * 64) if "FETCH_SIZE" in st else None
    fetch_raw = acc.get("FETCH_SIZE"); write_raw = acc.get("WRITE_SIZE")
    fetch_bytes = None if fetch_raw is None else fetch_raw * 1024 / (gather_factor or 1.0)
    rec = {"log_n": int(arg("--log-n", 20)), "curve": arg("--curve", "pallas"), "window_bits": int(arg("--window-bits", 16)), "kernel_rev": arg("--kernel-rev"),
           "kernel": "k_msm_accum0", "FETCH_SIZE_KiB_raw": fetch_raw, "WRITE_SIZE_KiB_raw": write_raw,
           "calibration": {"k_gather64_FETCH_SIZE_over_known_bytes": gather_factor, "k_stream16_FETCH_SIZE_over_known_bytes": stream_factor,
                           "note": "tools/pmc_calib: 2^24 records of 64 B gathered once each from a 1 GiB table (known 68 B read per record incl. the index), and the same table streamed 16 B per lane (the guide's calibrated case: 0.5)"},
           "msm_accumulate_hbm_bytes_per_launch": None if fetch_bytes is None or write_raw is None else int(fetch_bytes + write_raw * 1024),
           "note": "FETCH_SIZE / WRITE_SIZE from separate --pmc passes (KiB); fetch corrected by the gather calibration factor, write exact"}
    json.dump(rec, open(arg("--json"), "w"), indent=1)
    print("wrote", arg("--json"), rec["msm_accumulate_hbm_bytes_per_launch"])
