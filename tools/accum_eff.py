#!/usr/bin/env python3
"""k_msm_accum0 by MSM shape: sorted pairs per microsecond for uniform scalars at n = 2^k, `batch` columns, the library's own window for that size
(precomputed tables).  Run under rocprofv3 --kernel-trace; tools/accum_eff_read.py pairs the launches with the shapes printed here.
   python tools/accum_eff.py K BATCH[,BATCH...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package()
from dehalo2_amd import native
k = int(sys.argv[1]); batches = [int(b) for b in sys.argv[2].split(",")]
n = 1 << k
ctx = pkg.Context(0)
rng = np.random.default_rng(5)
# bases: the SRS of a fixed secret, made by the library on the device (dehalo_params_setup) and registered with the library's window for this size
params = native.ParamsKZG.setup(ctx, pkg.fields.BN254, k, 12345)
g = np.frombuffer(params.write()[4:4 + 64 * n], dtype=np.uint64).reshape(n, 8)
h = ctx.register_bases(pkg.fields.BN254.id, g, 0, True)
out = []
for b in batches:
    cols = rng.integers(0, 1 << 62, size=(b, n, 4), dtype=np.int64)
    cols[:, :, 3] &= (1 << 60) - 1                                  # < the modulus; Montgomery form of SOME canonical value: digits are uniform either way
    with ctx.torch_stream():
        d = ctx.upload(cols.reshape(-1))
        res = torch.zeros((b, 12), dtype=torch.int64, device="cuda")
        for rep in range(4):
            ctx.msm_device(h, d.data_ptr(), n, b, res.data_ptr(), 0)
        ctx.synchronize()
    sh = ctx.msm_last_shape()
    out.append({"k": k, "batch": b, "launches": 4, **{a: int(v) for a, v in sh.items()}})
    print(json.dumps(out[-1]), flush=True)
