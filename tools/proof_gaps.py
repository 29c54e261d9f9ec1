#!/usr/bin/env python3
"""Device idle time inside the LAST create_proof of a rocprofv3 --kernel-trace csv: the union of kernel intervals over all queues against the proof's span,
the gaps longer than 5 us (what ran before / after), and per queue the busy time.   usage: tools/proof_gaps.py kernel_trace.csv [msm_calls_per_proof=6]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_msm_accum0' in r['Kernel_Name']]
start = max(0, idx[-per] - 12)
rs = rows[start:]
t0 = int(rs[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rs)
name = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), name(r), r.get('Queue_Id')) for r in rs)
busy = 0; cur_s, cur_e, last = iv[0][0], iv[0][1], iv[0]
gaps = []
for s, e, nm, q in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e - t0, s - cur_e, last[2], nm))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: last = (s, e, nm, q)
busy += cur_e - cur_s
print("span %.3f ms, device busy (union over queues) %.3f ms = %.1f %%, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(rs)))
print("idle gaps > 5 us (at ms, length us, after -> before):")
for at, ln, a, b in gaps:
    if ln > 5000: print("  %7.3f  %6.1f   %s -> %s" % (at / 1e6, ln / 1e3, a, b))
print("sum of gaps <= 5 us: %.1f us in %d gaps" % (sum(g[1] for g in gaps if g[1] <= 5000) / 1e3, sum(1 for g in gaps if g[1] <= 5000)))
per_q = {}
for s, e, nm, q in iv: per_q[q] = per_q.get(q, 0) + (e - s)
print("kernel time per queue (ms):", {q: round(v / 1e6, 3) for q, v in per_q.items()})
