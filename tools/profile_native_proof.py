#!/usr/bin/env python3
"""Steady-state dehalo_create_proof (the whole-call C ABI) with a side context: per-phase host wall times (dehalo_prover_last_timings), then two
traced proofs (DEHALO_PROVER_TRACE=1: the host's timeline inside the phases, on stderr).   python tools/profile_native_proof.py [k] [circuit] [proofs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, native
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
circuit = sys.argv[2] if len(sys.argv) > 2 else "delay_enc"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, circuit)
print(desc)
ctx, side = pkg.Context(0, priority=1), pkg.Context(0)
with ctx.torch_stream():
    adv = keygen.to_device(circ.advice)
    ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
ctx.synchronize()
params = native.ParamsKZG.setup(ctx, curve, k, 0x1234567890abcdef)      # ParamsKZG::setup on the device (dehalo_params_setup)
pk = native.ProvingKey.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
N = native.Prover(params, pk, ctx, side)
for _ in range(5): N.create_proof(adv, [[]], prover.SeededRng(7))
ts, best = [], None
for _ in range(reps):
    t = time.perf_counter(); N.create_proof(adv, [[]], prover.SeededRng(7)); el = 1e3 * (time.perf_counter() - t)
    ts.append(el)
    if best is None or el <= min(ts): best = N.last_timings()
print("k = %d %s: min %.3f ms, median %.3f ms over %d proofs" % (k, circuit, min(ts), sorted(ts)[len(ts) // 2], reps))
print("phases of the fastest proof (host wall clock, ms):", {a: round(b, 3) for a, b in best.items()})
os.environ["DEHALO_PROVER_TRACE"] = "1"
for _ in range(2):
    sys.stderr.write("---- proof\n")
    N.create_proof(adv, [[]], prover.SeededRng(7))
