#!/usr/bin/env python3
"""One-shot (unregistered) MSM: window size sweep of the single-row path -- table build (host points in), device MSM, and the
host-buffer dehalo_best_multiexp as shipped.  python tools/sweep_single_row.py [curve]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
curve = pkg.fields.CURVES[sys.argv[1] if len(sys.argv) > 1 else "pallas"]
for log_n in (14, 17, 20):
    n = 1 << log_n
    bases = co.synth_bases(curve.id, n)
    sc = co.fill_scalars(curve.scalar.id, "uniform", n, 5)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda(); d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    row = []
    for c in (9, 10, 11, 12, 13, 14, 15, 16):
        ts = []
        for _ in range(3):
            t = time.perf_counter(); h = ctx.register_bases(curve.id, bases, c, False); ts.append(time.perf_counter() - t)
            if _ < 2: h.release()
        for _ in range(2): ctx.msm_device(h, d_sc.data_ptr(), n, 1, d_out.data_ptr(), 0)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ctx.msm_device(h, d_sc.data_ptr(), n, 1, d_out.data_ptr(), 0)
        ctx.synchronize(); dt = (time.perf_counter() - t0) / 5
        h.release()
        row.append("c%d reg %.2f msm %.3f" % (c, 1e3 * min(ts), dt * 1e3))
    tb = []
    for _ in range(4):
        t = time.perf_counter(); ctx.best_multiexp(curve.id, sc, bases); tb.append(time.perf_counter() - t)
    print("2^%d: " % log_n + " | ".join(row) + " || best_multiexp (host buffers) %.2f ms" % (1e3 * min(tb[1:])))
