#!/usr/bin/env python3
"""Runs the prover-shaped schedule a few times (for rocprofv3): python tools/profile_prover.py [k] [curve]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
from dehalo2_amd import prover_shape as ps
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
curve = pkg.fields.CURVES[sys.argv[2] if len(sys.argv) > 2 else "bn254"]
ctx = pkg.Context(0)
n = 1 << k
g = co.synth_bases(curve.id, n); gl = g[::-1].copy()
cols = ps.synthetic_columns(lambda fid, dist, m, seed: co.fill_scalars(fid, dist, m, seed), curve.scalar.id, k, 7)
wb = int(os.environ.get('WINDOW_BITS', '0'))
bg, bgl = ctx.register_bases(curve.id, g, wb, True), ctx.register_bases(curve.id, gl, wb, True)
fill = lambda fid, dist, m, seed: co.fill_scalars(fid, dist, m, seed)
cols.update(ps.synthetic_proving_key(fill, curve.scalar, k, k + 2, 55))
shape = ps.ProverShape(ctx, curve, k, bgl, bg, cols, with_quotient=True)
shape.run()
for _ in range(3):
    r = shape.run()
    print("total %.3f ms  msm %.3f  ntt %.3f  evaluate_h %.3f  arguments %.3f  openings %.3f" % (r.ms_total, r.ms_msm, r.ms_ntt, r.ms_eval_h, r.ms_arguments, r.ms_openings))
