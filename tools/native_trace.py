#!/usr/bin/env python3
"""One traced native proof (DEHALO_PROVER_TRACE=1: host timestamps inside dehalo_create_proof go to stderr).  python tools/native_trace.py [k] [circuit]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, native
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
circuit = sys.argv[2] if len(sys.argv) > 2 else "delay_enc"
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, circuit)
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
ctx, side = pkg.Context(0), pkg.Context(0)
with ctx.torch_stream():
    adv = keygen.to_device(circ.advice)
    ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
ctx.synchronize()
nparams = native.ParamsKZG.create(ctx, curve, k, srs["g"], srs["g_lagrange"])
npk = native.ProvingKey.keygen(ctx, nparams, circ.cs, circ.fixed, circ.assembly, circ.selectors)
N = native.Prover(nparams, npk, ctx, side)
for _ in range(5): N.create_proof(adv, [[]], prover.SeededRng(7))
os.environ["DEHALO_PROVER_TRACE"] = "1"
for _ in range(2):
    sys.stderr.write("---- proof\n")
    N.create_proof(adv, [[]], prover.SeededRng(7))
