#!/usr/bin/env python3
"""Randomised MSM configurations against the C restatement (more trials than the test suite runs): curve, n, window bits, table
mode, batch, scalar distribution, prefix length, affine output.   python tools/stress_msm.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
curves = [pkg.fields.BN254, pkg.fields.PALLAS, pkg.fields.VESTA]
dists = ["uniform", "witness", "lookup"]
ctx = pkg.Context(0)
for trial in range(trials):
    spec = curves[trial % 3]
    n = int(rng.integers(1, 3000)) if trial % 3 else int(rng.integers(3000, 70000))
    c = int(rng.choice([0, 4, 5, 7, 9, 10, 12, 13, 14, 15, 16, 17]))
    precompute = bool(rng.integers(0, 2)) or c == 17      # (17 bits: precomputed rows only)
    if trial % 7 == 3:                                     # every seventh trial with the sort in 512-thread workgroups
        ctx.set_tuning("msm_sort_block", 512)
    elif trial % 7 == 4:
        ctx.set_tuning("msm_sort_block", 1024)
    batch = int(rng.choice([1, 1, 2, 3, 6, 9]))
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, c, precompute)
    cols = []
    for j in range(batch):
        kind = int(rng.integers(0, 5))
        if kind < 3:
            col = co.fill_scalars(spec.scalar.id, dists[kind], n, 9000 + 10 * trial + j)
        elif kind == 3:                                   # a few values repeated many times (heavy buckets)
            vals = co.fill_scalars(spec.scalar.id, "uniform", 4, 5000 + trial)
            col = vals[rng.integers(0, 4, size=n)]
        else:                                             # zero runs
            col = co.fill_scalars(spec.scalar.id, "uniform", n, 7000 + trial)
            col[: int(rng.integers(0, n + 1))] = 0
        cols.append(np.ascontiguousarray(col))
    m = int(rng.integers(1, n + 1))
    with ctx.torch_stream():
        d = torch.from_numpy(np.stack([c_[:m] for c_ in cols]).view(np.int64).copy()).cuda()
        aff = torch.zeros((batch, 8), dtype=torch.int64, device="cuda")
        ctx.msm_device_affine(h, d.data_ptr(), m, batch, 0, aff.data_ptr(), 0)
        ctx.synchronize()
        got = aff.cpu().numpy().view(np.uint64)
    got2 = ctx.to_affine(spec.id, ctx.msm_batch(h, [c_[:m] for c_ in cols]))
    h.release()
    for j, col in enumerate(cols):
        want = co.to_affine(spec.id, co.best_multiexp(spec.id, col[:m], bases[:m], 8))
        assert np.array_equal(got[j], want) and np.array_equal(got2[j], want), (trial, spec.name, n, m, c, precompute, batch, j)
print("%d randomised MSM configurations: all equal to the C restatement" % trials)
