#!/usr/bin/env python3
"""Device busy fraction and concurrency of a batch-mode window in a rocprofv3 --kernel-trace csv (tools/batch_trace.py): the window runs
between the two marker fills; busy = union of kernel intervals / window; per-kernel totals inside the window.
usage: tools/busy_fraction.py kernel_trace.csv [proofs]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'FillFunctor' in r['Kernel_Name'] and r['Grid_Size_X'] in ('12544', '12352', '12345', '3136', '6272')]
if len(marks) < 2:
    marks = [i for i, r in enumerate(rows) if 'FillFunctor' in r['Kernel_Name']][-2:]
a, b = marks[-2], marks[-1]
win = rows[a + 1:b]
t0, t1 = int(rows[a]['End_Timestamp']), int(rows[b]['Start_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in win)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
total = sum(e - s for s, e in iv)
W = t1 - t0
print("window %.3f ms, %d kernels, %d proofs: %.3f ms per proof" % (W / 1e6, len(win), proofs, W / 1e6 / proofs))
print("device busy (union of kernel intervals) %.1f %%; sum of kernel durations %.3f ms per proof; mean concurrency while busy %.2f" % (100.0 * busy / W, total / 1e6 / proofs, total / max(busy, 1)))
agg = {}
for r in win:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    c = agg.setdefault(nm, [0, 0]); c[0] += 1; c[1] += d
for nm, (cnt, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-50s %6.1f launches / proof  %7.3f ms / proof  avg %8.1f us" % (nm, cnt / proofs, d / 1e6 / proofs, d / 1e3 / cnt))
# time during which exactly ONE kernel was running, by kernel (who runs alone?), and total time by concurrency level
ev = []
for r in win:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    ev.append((int(r['Start_Timestamp']), 1, nm)); ev.append((int(r['End_Timestamp']), -1, nm))
ev.sort()
active, alone, level = {}, {}, {}
prev = None
for t, d, nm in ev:
    if prev is not None and t > prev:
        n_act = sum(active.values())
        level[n_act] = level.get(n_act, 0) + (t - prev)
        if n_act == 1:
            k = next(k for k, v in active.items() if v)
            alone[k] = alone.get(k, 0) + (t - prev)
    active[nm] = active.get(nm, 0) + d
    prev = t
print("time by number of kernels running: " + ", ".join("%d: %.1f%%" % (k, 100.0 * v / W) for k, v in sorted(level.items())))
print("running ALONE (ms per proof):")
for nm, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-42s %7.3f" % (nm, v / 1e6 / proofs))
