#!/usr/bin/env python3
"""pairs tools/accum_eff.py's printed shapes with the k_msm_accum0 launches of the rocprofv3 kernel trace taken over it.
   python tools/accum_eff_read.py SHAPES.jsonl KERNEL_TRACE.csv"""
import csv, json, sys
shapes = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
names = ["k_msm_hist", "k_msm_part", "k_msm_bucket", "k_msm_accum0", "k_msm_merge2", "k_msm_bred", "k_msm_merge_classify2"]
per = {nm: [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if nm in r["Kernel_Name"]] for nm in names}
span = []                                   # first kernel of an MSM (k_msm_hist) to the end of its k_msm_bred
hs = [int(r["Start_Timestamp"]) for r in rows if "k_msm_hist" in r["Kernel_Name"]]
be = [int(r["End_Timestamp"]) for r in rows if "k_msm_bred" in r["Kernel_Name"]]
span = [(e - s) / 1e3 for s, e in zip(hs, be)]
i = 0
print("k batch | sorted pairs, per lane | k_msm_accum0 us (pairs per us) | hist part bucket classify merge2 bred us | whole MSM us (first sort kernel to the end of the reduction)")
for sh in shapes:
    n = sh["launches"]
    sl = slice(i + 1, i + n)               # the first launch of a shape is its warm-up
    m = lambda v: min(v[sl]) if v[sl] else 0.0
    a = m(per["k_msm_accum0"])
    print("%2d %2d | %9d %3d | %7.1f (%6.0f) | %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f | %7.1f" % (sh["k"], sh["batch"], sh["pairs"], sh["points_per_lane"], a, sh["pairs"] / a if a else 0,
          m(per["k_msm_hist"]), m(per["k_msm_part"]), m(per["k_msm_bucket"]), m(per["k_msm_merge_classify2"]), m(per["k_msm_merge2"]), m(per["k_msm_bred"]), m(span)))
    i += n
