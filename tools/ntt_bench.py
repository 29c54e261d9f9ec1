"""NTT timings at the sizes of the bench line and of a k = 17 proof (device-resident, min of 10):
plain transforms with the full and the half twiddle table (context knob "ntt_full_table_log"), and the proof's padded coset
transform coeff_to_extended 2^17 -> 2^19 x 23 (the first pass skips the two copy stages of the zero rows)."""
import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
full = pkg.Context(0)
full.set_tuning("ntt_full_table_log", 24)

def best(fn, c, reps=30):
    for _ in range(10): fn()
    c.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); c.synchronize(); ts.append(time.perf_counter() - t)
    return 1e3 * min(ts)

# clocks up before anything is compared
_w = torch.zeros(1 << 22, 4, dtype=torch.int64, device="cuda")
_f = pkg.fields.FIELDS["bn254_fr"]
_om = _f.encode(po.FIELDS["bn254_fr"].omega(19))
for _ in range(150): ctx.ntt_device(_f.id, _w.data_ptr(), 19, _om, 8, 0)
ctx.synchronize()

for fname, log_n, batch in (("pasta_fp", 20, 1), ("bn254_fr", 19, 23), ("bn254_fr", 17, 22)):
    f = pkg.fields.FIELDS[fname]
    a = co.fill_scalars(f.id, "uniform", batch << log_n, 3)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    om = f.encode(po.FIELDS[fname].omega(log_n))
    torch.cuda.synchronize()
    row = []
    for name, c in (("half table", ctx), ("full table", full), ("half again", ctx), ("full again", full)):
        row.append("%s %.4f ms" % (name, best(lambda: c.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0), c)))
    print("%s 2^%d x %d: %s" % (fname, log_n, batch, " | ".join(row)))

# coeff_to_extended as the prover runs it: 23 coefficient vectors of 2^17, zero-padded to 2^19, x zeta^(i mod 3), forward transform
spec = pkg.fields.FIELDS["bn254_fr"]
dom = pkg.EvaluationDomain(ctx, spec, 5, 17)
batch = 23
coeffs = torch.from_numpy(co.fill_scalars(spec.id, "uniform", batch << 17, 9).view(np.int64)).cuda()
ext = torch.zeros(batch << 19, 4, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
e = spec.encode
t = best(lambda: ctx.coset_ntt_device(spec.id, coeffs.data_ptr(), 17, ext.data_ptr(), dom.extended_k, e(dom.extended_omega), e(dom.g_coset), batch, 0), ctx)
print("bn254_fr coeff_to_extended 2^17 -> 2^%d x %d: %.4f ms" % (dom.extended_k, batch, t))
