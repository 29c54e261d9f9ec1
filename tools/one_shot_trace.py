#!/usr/bin/env python3
"""Three device MSMs over a single-row (unregistered-style) table: run under rocprofv3 --kernel-trace, then tools/step_timeline.py.  [log_n] [curve]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
curve = pkg.fields.CURVES[sys.argv[2] if len(sys.argv) > 2 else "pallas"]
n = 1 << log_n
bases = co.synth_bases(curve.id, n)
sc = co.fill_scalars(curve.scalar.id, "uniform", n, 5)
d_sc = torch.from_numpy(sc.view(np.int64)).cuda(); d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
h = ctx.register_bases(curve.id, bases, 0, False)
for _ in range(3):
    ctx.msm_device(h, d_sc.data_ptr(), n, 1, d_out.data_ptr(), 0)
    ctx.synchronize()
