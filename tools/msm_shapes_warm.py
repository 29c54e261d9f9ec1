"""Warm device times (HIP events on the context's stream, 60 launches after a 0.3 s preheat) of one BN254 MSM launch by shape: n = 2^k uniform columns x batch over the
library's precomputed table -- sort / accumulate / merge + reduce, sorted pairs per microsecond of the accumulation, points per lane.   python tools/msm_shapes_warm.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
from dehalo2_amd import _lib
ctx = pkg.Context(0)
curve = pkg.fields.BN254
for k, batches in ((17, (1, 2, 3, 4, 5, 7, 8, 10)), (19, (1, 4)), (20, (1, 4))):
    n = 1 << k
    bases = co.synth_bases(curve.id, n)
    h = ctx.register_bases(curve.id, bases, 0, True)
    for b in batches:
        sc = np.concatenate([co.fill_scalars(curve.scalar.id, "uniform", n, 100 + j) for j in range(b)])
        d = ctx.upload(sc)
        out = torch.zeros((b, 12), dtype=torch.int64, device="cuda")
        run = lambda: ctx.msm_device(h, d.data_ptr(), n, b, out.data_ptr(), 0)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3: run()
        ctx.synchronize()
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(60): run()
        ctx.synchronize(); ctx.timing_enable(False)
        s, a, r = (ctx.timing_get(kid)[0] / 60 for kid in (_lib.K_MSM_SORT, _lib.K_MSM_ACCUMULATE, _lib.K_MSM_REDUCE))
        sh = ctx.msm_last_shape()
        print("k %2d batch %2d window %2d: sort %7.1f us, accumulate %7.1f us (%6.0f pairs / us, %2d points per lane), merge + reduce %6.1f us" %
              (k, b, h.window_bits, 1e3 * s, 1e3 * a, sh["pairs"] / (1e3 * a), sh["points_per_lane"], 1e3 * r), flush=True)
    h.release()
