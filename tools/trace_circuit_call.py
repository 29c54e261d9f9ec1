#!/usr/bin/env python3
"""Host timeline (DEHALO_PROVER_TRACE) of one dehalo_create_proof_circuit call at k = 17 after warm-up.   python tools/trace_circuit_call.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, native
import bench
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, 17, "delay_enc")
srs = PO.setup_srs(po.BN254, 17, 0x1234567890abcdef, 16)
ctx, side = pkg.Context(0, priority=1), pkg.Context(0, priority=-1)
params = native.ParamsKZG.create(ctx, curve, 17, srs["g"], srs["g_lagrange"])
pk = native.ProvingKey.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
N = native.Prover(params, pk, ctx, side)
spec = circ.native_spec
kw = {a: b for a, b in spec.items() if a not in ("circuit", "k")}
for _ in range(30): N.create_proof_circuit(spec["circuit"], [[]], prover.SeededRng(7), **kw)
os.environ["DEHALO_PROVER_TRACE"] = "1"
t = time.perf_counter(); N.create_proof_circuit(spec["circuit"], [[]], prover.SeededRng(7), **kw); print("call: %.3f ms" % (1e3 * (time.perf_counter() - t)))
