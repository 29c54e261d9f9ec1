import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
for fname, log_n, batch in (("pasta_fp", 20, 1), ("bn254_fr", 19, 23), ("bn254_fr", 17, 22)):
    f = pkg.fields.FIELDS[fname]
    a = co.fill_scalars(f.id, "uniform", batch << log_n, 3)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    om = f.encode(po.FIELDS[fname].omega(log_n))
    torch.cuda.synchronize()
    for _ in range(3): ctx.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0)
    ctx.synchronize()
    ts = []
    for _ in range(10):
        t = time.perf_counter(); ctx.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0); ctx.synchronize(); ts.append(time.perf_counter() - t)
    print("%s 2^%d x %d: %.4f ms" % (fname, log_n, batch, 1e3 * min(ts)))
