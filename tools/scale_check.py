#!/usr/bin/env python3
"""Large-size checks that are too slow / too big for the test suite: 2^23-point MSM (linearity and a sampled
sub-MSM against the oracle), 2^26-point NTT round trip, a 64-column batch.  Prints PASS lines."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)

curve = pkg.fields.CURVES["pallas"]; fid = curve.scalar.id
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n = 1 << log_n
t0 = time.time(); bases = co.synth_bases(curve.id, n); print("bases %.1fs" % (time.time() - t0), flush=True)
t0 = time.time(); h = ctx.register_bases(curve.id, bases, 0, True); print("register 2^%d: %.2fs" % (log_n, time.time() - t0), flush=True)
a = co.fill_scalars(fid, "uniform", n, 1); b = co.fill_scalars(fid, "witness", n, 2)
apb = ctx.field_op(fid, "add", a, b)
t0 = time.time(); ra, rb, rab = ctx.msm(h, a), ctx.msm(h, b), ctx.msm(h, apb); print("3 MSMs (host buffers) %.2fs" % (time.time() - t0), flush=True)
aff = ctx.to_affine(curve.id, np.stack([ra, rb, rab]))
# ra + rb == rab through the oracle's group law
f = po.PALLAS
dec = lambda xy: None if not xy.any() else (pkg.fields.PALLAS.base.decode(xy[:4]), pkg.fields.PALLAS.base.decode(xy[4:]))
s = po.ec_add(f, dec(aff[0]), dec(aff[1]))
assert s == dec(aff[2]), "MSM linearity failed"
print("PASS msm 2^%d linearity: MSM(a) + MSM(b) == MSM(a + b)" % log_n, flush=True)
# a sparse selection: only 4096 scalars non-zero -> equals the oracle's MSM over those points
sel = np.zeros_like(a); idx = np.random.default_rng(3).choice(n, 4096, replace=False); sel[idx] = a[idx]
got = ctx.to_affine(curve.id, ctx.msm(h, sel).reshape(1, 12))[0]
want = co.to_affine(curve.id, co.best_multiexp(curve.id, a[idx], bases[idx], 8))
assert np.array_equal(got, want), "sparse MSM differs from the oracle"
print("PASS msm 2^%d with 4096 non-zero scalars == oracle MSM of those terms" % log_n, flush=True)
h.release(); del bases

# NTT 2^26 round trip on the device
fld = pkg.fields.FIELDS["bn254_fr"]; pf = po.FIELDS["bn254_fr"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 26
N = 1 << k
x = co.fill_scalars(fld.id, "uniform", N, 5)
d = torch.from_numpy(x.view(np.int64).copy()).cuda()
w, wi, ninv = fld.encode(pf.omega(k)), fld.encode(pf.inv(pf.omega(k))), fld.encode(pf.inv(N))
t0 = time.time(); ctx.ntt_device(fld.id, d.data_ptr(), k, w, 1, 0); ctx.synchronize(); t1 = time.time()
mid = d[:4].cpu().numpy().view(np.uint64).copy()
ctx.intt_scaled_device(fld.id, d.data_ptr(), k, wi, ninv, 1, 0); ctx.synchronize()
assert np.array_equal(d.cpu().numpy().view(np.uint64), x), "NTT round trip failed"
# first outputs against the definition: a'[i] = sum_j a[j] w^(ij) for i = 0 (plain sum) via eval_polynomial at 1 and at w
one = fld.encode(1)
assert np.array_equal(mid[0], ctx.eval_polynomial(fld.id, x, one)) and np.array_equal(mid[1], ctx.eval_polynomial(fld.id, x, w))
print("PASS ntt 2^%d round trip (forward %.1f ms) and outputs 0, 1 == eval_polynomial(a, 1), eval_polynomial(a, w)" % (k, (t1 - t0) * 1e3), flush=True)
