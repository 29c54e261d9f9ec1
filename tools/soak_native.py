#!/usr/bin/env python3
"""Soak of the whole-call prover: many dehalo_create_proof calls on one prover with a side context (helper thread, its own stream, deferred lookup status, out-of-place
transforms), seeds cycled, every proof byte-compared with the first one made from its seed; then rounds of dehalo_create_proofs on four provers compared with the same.
A race between the contexts / threads shows up as a different proof.   python tools/soak_native.py [k] [circuit] [proofs] [batch rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, native
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
circuit = sys.argv[2] if len(sys.argv) > 2 else "delay_enc"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 10
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, circuit)
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
ctx, side = pkg.Context(0, priority=1), pkg.Context(0)
with ctx.torch_stream():
    adv = keygen.to_device(circ.advice)
    ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
ctx.synchronize()
params = native.ParamsKZG.create(ctx, curve, k, srs["g"], srs["g_lagrange"])
pk = native.ProvingKey.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
N = native.Prover(params, pk, ctx, side)
seeds = list(range(50, 58))
first = {}
t0 = time.perf_counter()
for i in range(count):
    sd = seeds[i % len(seeds)]
    got = N.create_proof(adv, [[]], prover.SeededRng(sd)).finalize()
    if sd not in first: first[sd] = got
    assert got == first[sd], "proof %d (seed %d) differs from the first proof of that seed" % (i, sd)
    if i % 250 == 249: print("  %d proofs" % (i + 1), flush=True)
print("k = %d %s: %d proofs on one prover + side context, every one identical to the first of its seed; %.2f ms per proof" % (k, circuit, count, 1e3 * (time.perf_counter() - t0) / count), flush=True)
ctxs = [pkg.Context(0, priority=(1, 0, -1)[i % 3]) for i in range(4)]
provers = [native.Prover(params, pk, c) for c in ctxs]
for r in range(rounds):
    proofs = native.create_proofs(provers, adv, [prover.SeededRng(seeds[i % len(seeds)]) for i in range(32)])
    for i, pf in enumerate(proofs):
        assert pf == first[seeds[i % len(seeds)]], "batch round %d proof %d differs" % (r, i)
print("%d rounds of 32 proofs on four provers (one library thread each): all identical to the lone prover's" % rounds)
# the circuit-level calls (witness synthesized inside, by the synthesis pool) and a host witness shared by four library threads (HostPin's shared registration)
spec = getattr(circ, "native_spec", None)
if spec is not None:
    import numpy as np
    kw = {a: b for a, b in spec.items() if a not in ("circuit", "k")}
    n_c = max(50, count // 5)
    for i in range(n_c):
        sd = seeds[i % len(seeds)]
        assert N.create_proof_circuit(spec["circuit"], [[]], prover.SeededRng(sd), **kw)[0].finalize() == first[sd], "create_proof_circuit %d differs" % i
    print("%d dehalo_create_proof_circuit calls: identical" % n_c, flush=True)
    host = native.synthesize(spec["circuit"], spec["k"], **kw)["advice"]
    host_m = np.stack([co.field_op(curve.scalar.id, "to_mont", host[i]) for i in range(host.shape[0])])
    for r in range(rounds):
        proofs = native.create_proofs(provers, host_m, [prover.SeededRng(seeds[i % len(seeds)]) for i in range(32)])
        assert all(pf == first[seeds[i % len(seeds)]] for i, pf in enumerate(proofs)), "host-witness batch round %d differs" % r
        proofs = native.create_proofs_circuit(provers, spec["circuit"], [kw] * 32, [prover.SeededRng(seeds[i % len(seeds)]) for i in range(32)])
        assert all(pf == first[seeds[i % len(seeds)]] for i, pf in enumerate(proofs)), "circuit batch round %d differs" % r
    print("%d rounds of 32 proofs from one shared HOST witness array and %d rounds from circuits synthesized inside the calls, four provers: all identical" % (rounds, rounds))
