import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
for fname, log_n, batch in (("pasta_fp", 20, 1), ("bn254_fr", 19, 23), ("bn254_fr", 17, 24)):
    f = pkg.fields.FIELDS[fname]
    a = np.stack([co.fill_scalars(f.id, "uniform", 1 << log_n, 3)] * batch)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    om = f.encode(po.FIELDS[fname].omega(log_n))
    for _ in range(3): ctx.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%s 2^%d x%d: %.4f ms" % (fname, log_n, batch, dt * 1e3), end="  |  ")
print()
