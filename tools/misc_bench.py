#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md: other curves / sizes / entry points (not the headline)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)

def t_msm_device(curve, log_n, dist="uniform", precompute=True, reps=10):
    n = 1 << log_n
    bases = co.synth_bases(curve.id, n); sc = co.fill_scalars(curve.scalar.id, dist, n, 5)
    t0 = time.time(); h = ctx.register_bases(curve.id, bases, 0, precompute); t_reg = time.time() - t0
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda(); d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    for _ in range(2): ctx.msm_device(h, d_sc.data_ptr(), n, 1, d_out.data_ptr(), 0)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ctx.msm_device(h, d_sc.data_ptr(), n, 1, d_out.data_ptr(), 0)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / reps
    h.release()
    return dt * 1e3, t_reg * 1e3

for cname in ("pallas", "bn254"):
    curve = pkg.fields.CURVES[cname]
    for log_n in (14, 17, 20, 22):
        for dist in ("uniform", "witness"):
            ms, reg = t_msm_device(curve, log_n, dist)
            print("msm %-6s 2^%-2d %-8s precomputed rows: %8.3f ms  (%7.1f Mpoints/s)  register %.0f ms" % (cname, log_n, dist, ms, (1 << log_n) / ms / 1e3, reg))
    ms, reg = t_msm_device(curve, 20, "uniform", precompute=False, reps=5)
    print("msm %-6s 2^20 uniform  single row (W bucket groups): %8.3f ms  register %.0f ms" % (cname, ms, reg))
    n = 1 << 20
    bases = co.synth_bases(curve.id, n); sc = co.fill_scalars(curve.scalar.id, "uniform", n, 5)
    ctx.best_multiexp(curve.id, sc, bases)
    t0 = time.perf_counter(); ctx.best_multiexp(curve.id, sc, bases); dt = time.perf_counter() - t0
    print("best_multiexp %-6s 2^20 host buffers, unregistered bases (upload + convert + MSM): %.2f ms" % (cname, dt * 1e3))
for fname in ("pasta_fp", "bn254_fr"):
    f = pkg.fields.FIELDS[fname]
    for log_n in (14, 17, 20, 22, 24):
        a = co.fill_scalars(f.id, "uniform", 1 << log_n, 3)
        d = torch.from_numpy(a.view(np.int64)).cuda()
        om = f.encode(po.FIELDS[fname].omega(log_n))
        for _ in range(2): ctx.ntt_device(f.id, d.data_ptr(), log_n, om, 1, 0)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ctx.ntt_device(f.id, d.data_ptr(), log_n, om, 1, 0)
        ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("ntt %-8s 2^%-2d %8.3f ms  %7.1f GB/s algorithmic (64 B/elem)" % (fname, log_n, dt * 1e3, 64 * (1 << log_n) / dt / 1e9))
