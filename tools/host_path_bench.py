import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
for cname in ("bn254", "pallas"):
    curve = pkg.fields.CURVES[cname]
    for log_n in (14, 17, 20):
        n = 1 << log_n
        bases = co.synth_bases(curve.id, n); sc = co.fill_scalars(curve.scalar.id, "uniform", n, 5)
        h = ctx.register_bases(curve.id, bases, 0, True)
        for _ in range(2): ctx.msm(h, sc)
        t0 = time.perf_counter()
        for _ in range(10): ctx.msm(h, sc)
        dt = (time.perf_counter() - t0) / 10
        cols = [co.fill_scalars(curve.scalar.id, "witness", n, 9 + i) for i in range(5)]
        ctx.msm_batch(h, cols)
        t0 = time.perf_counter()
        for _ in range(5): ctx.msm_batch(h, cols)
        db = (time.perf_counter() - t0) / 5
        print("%-6s 2^%-2d host-buffer msm: %7.3f ms (%6.1f Mpoints/s, PCIe-inclusive) | msm_batch x5: %7.3f ms" % (cname, log_n, dt * 1e3, n / dt / 1e6, db * 1e3))
        h.release()
f = pkg.fields.BN254_FR
for log_n in (17, 19, 20):
    a = co.fill_scalars(f.id, "uniform", 1 << log_n, 3); om = f.encode(po.FIELDS[f.name].omega(log_n))
    for _ in range(2): ctx.ntt(f.id, a, log_n, om)
    t0 = time.perf_counter()
    for _ in range(5): ctx.ntt(f.id, a, log_n, om)
    dt = (time.perf_counter() - t0) / 5
    print("ntt bn254_fr 2^%d host-buffer (H2D + NTT + D2H): %.3f ms" % (log_n, dt * 1e3))
