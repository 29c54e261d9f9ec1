import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
fid = pkg.fields.BN254_FR.id
x = co.fill_scalars(fid, "uniform", 1, 3)[0]
for log_n in (20, 17, 20, 22, 20, 19, 21, 20):
    n = 1 << log_n
    a = co.fill_scalars(fid, "uniform", n, 1)
    da = torch.from_numpy(a.view(np.int64).copy()).cuda(); dout = torch.zeros((1, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        t = time.perf_counter(); ctx.eval_polynomial_device(fid, da.data_ptr(), n, n, 1, x, dout.data_ptr(), 0); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    print("2^%d:" % log_n, " ".join("%.3f" % t for t in ts))
