#!/usr/bin/env python3
"""create_proof driven from C++ (dehalo_create_proof) beside the Python-driven prover, and batch throughput (dehalo_create_proofs) for
several numbers of provers in flight.   python tools/native_bench.py [k=17] [circuit=delay_enc] [batch=64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import fine_grained_prover as fgp
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, transcript, native
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
circuit = sys.argv[2] if len(sys.argv) > 2 else "delay_enc"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, circuit)
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
ctx, side = pkg.Context(0, priority=1), pkg.Context(0)
import torch
with ctx.torch_stream():
    adv = keygen.to_device(circ.advice)
    ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
ctx.synchronize()
# Python-driven
params = keygen.ParamsKZG(ctx, curve, k, srs["g"], srs["g_lagrange"])
pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
P = fgp.Prover(params, pk, ctx, side)
def py_prove(seed):
    tr = transcript.Blake2bWrite(curve); P.create_proof(adv, [[]], prover.SeededRng(seed), tr); return tr.finalize()
# native
nparams = native.ParamsKZG.create(ctx, curve, k, srs["g"], srs["g_lagrange"])
npk = native.ProvingKey.keygen(ctx, nparams, circ.cs, circ.fixed, circ.assembly, circ.selectors)
npk.transcript_repr = pk.vk.transcript_repr
ctx2, side2 = pkg.Context(0, priority=1), pkg.Context(0, priority=-1)      # streams of one priority share few hardware queues (tools/stream_concurrency.hip): the
# two provers of this comparison must not queue behind each other's idle streams
N = native.Prover(nparams, npk, ctx2, side2)
def nat_prove(seed):
    return N.create_proof(adv, [[]], prover.SeededRng(seed)).finalize()
want = py_prove(7)
assert nat_prove(7) == want
for name, fn in (("python-driven", py_prove), ("native (dehalo_create_proof)", nat_prove)):
    for _ in range(3): fn(7)
    ts = []
    for _ in range(15):
        t = time.perf_counter(); fn(7); ts.append(1e3 * (time.perf_counter() - t))
    print("%-30s k = %d %s: min %.3f ms, median %.3f ms" % (name, k, circuit, min(ts), sorted(ts)[len(ts) // 2]))
print("native phases (ms):", {a: round(b, 3) for a, b in N.last_timings().items()})
# OS entropy instead of the seeded stream
for _ in range(3): N.create_proof(adv, [[]])
ts = []
for _ in range(10):
    t = time.perf_counter(); N.create_proof(adv, [[]]); ts.append(1e3 * (time.perf_counter() - t))
print("native, OS-entropy rng: min %.3f ms, median %.3f ms" % (min(ts), sorted(ts)[len(ts) // 2]))
# batch throughput
for nprov, with_side in ((1, True), (2, False), (4, False), (4, True)):
    ctxs = [pkg.Context(0, priority=(1, 0, -1)[i % 3]) for i in range(nprov)]
    sides = [pkg.Context(0) for _ in range(nprov)] if with_side else [None] * nprov
    provers = [native.Prover(nparams, npk, c, s) for c, s in zip(ctxs, sides)]
    native.create_proofs(provers, adv, [prover.SeededRng(1000 + i) for i in range(2 * nprov)])
    best = None
    for rep in range(3):
        rngs = [prover.SeededRng(2000 + i) for i in range(batch)]
        t = time.perf_counter(); out = native.create_proofs(provers, adv, rngs); el = time.perf_counter() - t
        best = el if best is None or el < best else best
    assert len(set(out)) == batch
    print("batch of %d on %d provers%s: %.1f proofs/s (%.3f ms per proof)" % (batch, nprov, " + side contexts" if with_side else "", batch / best, 1e3 * best / batch))
    for p in provers: p.release()
    for c in ctxs + [s for s in sides if s is not None]: c.close()
