import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
curve = pkg.fields.CURVES[sys.argv[1] if len(sys.argv) > 1 else "bn254"]
for log_n in (12, 13, 14, 15, 16, 17, 18, 19):
    n = 1 << log_n
    bases = co.synth_bases(curve.id, n)
    row = []
    for batch, dist in ((1, "uniform"), (5, "witness")):
        sc = np.stack([co.fill_scalars(curve.scalar.id, dist, n, 5 + b) for b in range(batch)])
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda(); d_out = torch.zeros((batch, 12), dtype=torch.int64, device="cuda")
        for c in (10, 11, 12, 13, 14, 15, 16):
            if c > log_n + 2: continue
            h = ctx.register_bases(curve.id, bases, c, True)
            for _ in range(2): ctx.msm_device(h, d_sc.data_ptr(), n, batch, d_out.data_ptr(), 0)
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(8): ctx.msm_device(h, d_sc.data_ptr(), n, batch, d_out.data_ptr(), 0)
            ctx.synchronize(); dt = (time.perf_counter() - t0) / 8
            h.release()
            row.append("b%d c%d %.3f" % (batch, c, dt * 1e3))
    print("2^%d: " % log_n + " | ".join(row))
