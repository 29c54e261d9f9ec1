"""Workload for SQ counter passes over k_ntt_pass: five plain 23 x 2^19 bn256::Fr transforms, then five coeff_to_extended 2^17 -> 2^19 x 23.
rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/ntt_pmc.py"""
import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
spec = pkg.fields.FIELDS["bn254_fr"]
batch = 23
d = torch.from_numpy(co.fill_scalars(spec.id, "uniform", batch << 19, 3).view(np.int64)).cuda()
om = spec.encode(po.FIELDS["bn254_fr"].omega(19))
for _ in range(5): ctx.ntt_device(spec.id, d.data_ptr(), 19, om, batch, 0)
ctx.synchronize()
dom = pkg.EvaluationDomain(ctx, spec, 5, 17)
coeffs = torch.from_numpy(co.fill_scalars(spec.id, "uniform", batch << 17, 9).view(np.int64)).cuda()
ext = torch.zeros(batch << 19, 4, dtype=torch.int64, device="cuda")
e = spec.encode
for _ in range(5): ctx.coset_ntt_device(spec.id, coeffs.data_ptr(), 17, ext.data_ptr(), dom.extended_k, e(dom.extended_omega), e(dom.g_coset), batch, 0)
ctx.synchronize()
