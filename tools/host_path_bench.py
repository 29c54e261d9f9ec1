"""Host-buffer (PCIe-inclusive) entry points: the literal `&[F]` drop-in forms.  python tools/host_path_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
from dehalo2_amd import _lib
ctx = pkg.Context(0)
def best(fn, reps=6):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts[1:])
for cname in ("bn254", "pallas"):
    curve = pkg.fields.CURVES[cname]
    for log_n in (14, 17, 20):
        n = 1 << log_n
        bases = co.synth_bases(curve.id, n); sc = co.fill_scalars(curve.scalar.id, "uniform", n, 5)
        h = ctx.register_bases(curve.id, bases, 0, True)
        dt = best(lambda: ctx.msm(h, sc))
        cols = [co.fill_scalars(curve.scalar.id, "witness", n, 9 + i) for i in range(5)]
        db = best(lambda: ctx.msm_batch(h, cols), 4)
        print("%-6s 2^%-2d host-buffer msm: %7.3f ms (%6.1f Mpoints/s, PCIe-inclusive) | msm_batch x5: %7.3f ms" % (cname, log_n, dt * 1e3, n / dt / 1e6, db * 1e3))
        h.release()
f = pkg.fields.BN254_FR
for log_n in (17, 19, 20, 22):
    a = co.fill_scalars(f.id, "uniform", 1 << log_n, 3); om = f.encode(po.FIELDS[f.name].omega(log_n))
    buf = a.copy()
    dt = best(lambda: ctx._check(ctx.lib.dehalo_ntt(ctx.handle, f.id, buf.ctypes.data, log_n, om.ctypes.data)))   # in place on one caller buffer, as best_fft(&mut a) is
    print("ntt bn254_fr 2^%d host-buffer (H2D + NTT + D2H, in place): %.3f ms" % (log_n, dt * 1e3))
for fname in ("pasta_fp", "bn254_fr"):
    f = pkg.fields.FIELDS[fname]
    for log_n in (14, 17, 20):
        a = co.fill_scalars(f.id, "uniform", 1 << log_n, 3); x = f.encode(12345)
        dt = best(lambda: ctx.eval_polynomial(f.id, a, x))
        print("%s 2^%d eval_polynomial host-buffer: %.3f ms" % (fname, log_n, dt * 1e3))
