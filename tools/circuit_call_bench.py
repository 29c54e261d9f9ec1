#!/usr/bin/env python3
"""dehalo_create_proof_circuit at k = 17 (delay_enc): min / median over `reps` calls, beside dehalo_create_proof from the resident witness.
python tools/circuit_call_bench.py [reps=40]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, native, keygen
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, 17, "delay_enc")
srs = PO.setup_srs(po.BN254, 17, 0x1234567890abcdef, 16)
ctx, side = pkg.Context(0, priority=1), pkg.Context(0, priority=-1)
params = native.ParamsKZG.create(ctx, curve, 17, srs["g"], srs["g_lagrange"])
pk = native.ProvingKey.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
N = native.Prover(params, pk, ctx, side)
spec = circ.native_spec
kw = {a: b for a, b in spec.items() if a not in ("circuit", "k")}
with ctx.torch_stream():
    adv = keygen.to_device(circ.advice)
    ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
ctx.synchronize()
want = N.create_proof(adv, [[]], prover.SeededRng(7)).finalize()
def fused(): return N.create_proof_circuit(spec["circuit"], [[]], prover.SeededRng(7), **kw)[0].finalize()
def resident(): return N.create_proof(adv, [[]], prover.SeededRng(7)).finalize()
for name, fn in (("resident witness", resident), ("circuit-level call", fused), ("resident witness", resident), ("circuit-level call", fused)):
    for _ in range(10): assert fn() == want
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t))
    print("%-20s min %.3f ms, median %.3f ms  [DEHALO_SYNTH_STREAM=%s]" % (name, min(ts), sorted(ts)[len(ts) // 2], os.environ.get("DEHALO_SYNTH_STREAM", "unset")))
