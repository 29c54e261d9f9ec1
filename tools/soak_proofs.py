#!/usr/bin/env python3
"""Repeats create_proof with a side context and checks every proof's bytes against the first one (races between the two
contexts or with the helper thread would show up as a different proof).   python tools/soak_proofs.py [k] [range_lookups] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import fine_grained_prover as fgp
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, transcript
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
rl = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, "delay_enc" if rl else "pose_enc")
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
with pkg.Context(0) as ctx, pkg.Context(0) as side:
    params = keygen.ParamsKZG(ctx, curve, k, srs["g"], srs["g_lagrange"])
    pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    P = fgp.Prover(params, pk, side_ctx=side)
    with ctx.torch_stream():
        adv = keygen.to_device(circ.advice)
        ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
    ctx.synchronize()
    want = None
    t0 = time.perf_counter()
    for i in range(count):
        tr = transcript.Blake2bWrite(curve)
        P.create_proof(adv, [[]], prover.SeededRng(7), tr)
        got = tr.finalize()
        if want is None:
            want = got
        assert got == want, "proof %d differs from proof 0" % i
    print("k = %d: %d proofs, all %d bytes identical; %.2f ms per proof back to back" % (k, count, len(want), 1e3 * (time.perf_counter() - t0) / count))
    # batch mode: four provers, each on its own context and host thread, different seeds -- every proof against the one made alone
    import threading
    seeds = list(range(100, 100 + min(count, 48)))
    alone = {}
    for sd in seeds:
        tr = transcript.Blake2bWrite(curve); P.create_proof(adv, [[]], prover.SeededRng(sd), tr); alone[sd] = tr.finalize()
    ctxs = [pkg.Context(0) for _ in range(4)]
    provers = [fgp.Prover(params, pk, ctx=c) for c in ctxs]
    bad = []

    def work(j):
        for sd in seeds[j::4]:
            tr = transcript.Blake2bWrite(curve)
            provers[j].create_proof(adv, [[]], prover.SeededRng(sd), tr)
            if tr.finalize() != alone[sd]:
                bad.append(sd)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(j,)) for j in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not bad, "batch-mode proofs differ for seeds %s" % bad
    print("   batch mode: %d proofs on 4 contexts / threads, all identical to the ones made alone; %.2f ms per proof" % (len(seeds), 1e3 * (time.perf_counter() - t0) / len(seeds)))
    for c in ctxs:
        c.close()
    params.release()
