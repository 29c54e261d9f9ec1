"""Workload for SQ counter passes over k_ntt_pass: WARM (default 60) plain 23 x 2^19 bn256::Fr transforms to bring the clocks up -- round 4's version measured its ten
launches right after idle, at ramping clocks, and its "busy" percentages divide nominal-clock cycles by cold-clock durations --, then five more plain ones and five
coeff_to_extended 2^17 -> 2^19 x 23: tools/sq_summary.py --skip-launches 3*WARM reads the last ten transforms only.
rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/ntt_pmc.py [WARM]"""
import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
spec = pkg.fields.FIELDS["bn254_fr"]
batch = 23
d = ctx.upload(co.fill_scalars(spec.id, "uniform", batch << 19, 3))
om = spec.encode(po.FIELDS["bn254_fr"].omega(19))
warm = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("NTT_PMC_WARM", "60"))      # (the same number tools/refresh_profiles_r05.sh hands to sq_summary.py --skip-launches)
for _ in range(warm + 5): ctx.ntt_device(spec.id, d.data_ptr(), 19, om, batch, 0)
ctx.synchronize()
dom = pkg.EvaluationDomain(ctx, spec, 5, 17)
coeffs = ctx.upload(co.fill_scalars(spec.id, "uniform", batch << 17, 9))
ext = torch.zeros(batch << 19, 4, dtype=torch.int64, device="cuda")
e = spec.encode
for _ in range(5): ctx.coset_ntt_device(spec.id, coeffs.data_ptr(), 17, ext.data_ptr(), dom.extended_k, e(dom.extended_omega), e(dom.g_coset), batch, 0)
ctx.synchronize()
