"""One MSM at the size of a proof's commitments (2^17 points, bn256, the registered window tables), uniform scalars: the per-kernel
times of the twelve-kernel pipeline when it runs alone.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
curve = pkg.fields.CURVES["bn254"]
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 1 << log_n
bases = co.synth_bases(curve.id, n)
sc = np.stack([co.fill_scalars(curve.scalar.id, dist, n, 5 + b) for b in range(batch)])
d_sc = torch.from_numpy(sc.view(np.int64)).cuda(); d_out = torch.zeros((batch, 12), dtype=torch.int64, device="cuda")
h = ctx.register_bases(curve.id, bases, 0, True)
for _ in range(3): ctx.msm_device(h, d_sc.data_ptr(), n, batch, d_out.data_ptr(), 0)
ctx.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); ctx.msm_device(h, d_sc.data_ptr(), n, batch, d_out.data_ptr(), 0); ctx.synchronize(); ts.append(time.perf_counter() - t0)
print("2^%d x %d %s: min %.3f ms median %.3f ms" % (log_n, batch, dist, 1e3 * min(ts), 1e3 * sorted(ts)[10]))
