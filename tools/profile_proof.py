#!/usr/bin/env python3
"""Per-phase timings of the device create_proof (steady state).   python tools/profile_proof.py [k] [range_lookups] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import fine_grained_prover as fgp

pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO, pairing as pr
from dehalo2_amd import circuits, prover, keygen, transcript

k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
rl = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
curve = pkg.fields.BN254
import bench                                            # the bench's witness: the real DelayEncryptCircuit / PoseidonEncCircuit values
circ, desc, _ = bench.real_witness(curve.scalar.p, k, "delay_enc" if rl else "pose_enc")
print(desc)
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
with pkg.Context(0) as ctx:
    params = keygen.ParamsKZG(ctx, curve, k, srs["g"], srs["g_lagrange"])
    pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    P = fgp.Prover(params, pk)
    with ctx.torch_stream():
        adv = keygen.to_device(circ.advice)
        ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
    ctx.synchronize()
    for _ in range(2):
        P.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve))
    best, best_t = 1e9, None
    for _ in range(reps):
        tm = fgp.ProofTimings()
        P.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve), tm)
        if tm.total_ms < best:
            best, best_t = tm.total_ms, tm
    print("with per-phase syncs: total %.2f ms" % best, {a: round(b, 2) for a, b in best_t.phases_ms.items()})
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); P.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve)); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    print("no extra syncs: best %.2f ms, median %.2f ms" % (min(ts), sorted(ts)[len(ts) // 2]))

    class PlainRng:                                            # the same stream without fork(): no helper thread
        def __init__(self, seed): self.r = prover.SeededRng(seed)
        def scalars(self, c): return self.r.scalars(c)
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); P.create_proof(adv, [[]], PlainRng(7), transcript.Blake2bWrite(curve)); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    print("without the random-polynomial helper thread: best %.2f ms, median %.2f ms" % (min(ts), sorted(ts)[len(ts) // 2]))
    side = pkg.Context(0, priority=int(os.environ.get("SIDE_PRIO", "0")))
    P2 = fgp.Prover(params, pk, ctx, side)
    want = transcript.Blake2bWrite(curve); P.create_proof(adv, [[]], prover.SeededRng(7), want)
    ts = []
    for _ in range(reps + 2):
        tr = transcript.Blake2bWrite(curve)
        t = time.perf_counter(); P2.create_proof(adv, [[]], prover.SeededRng(7), tr); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
        assert tr.finalize() == want.finalize(), "side-context proof differs"
    ts = ts[2:]
    print("with a side context (NTTs and the random commitment beside the commitment phases): best %.2f ms, median %.2f ms" % (min(ts), sorted(ts)[len(ts) // 2]))
    tm = fgp.ProofTimings(); P2.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve), tm)
    print("   phases", {a: round(b, 2) for a, b in tm.phases_ms.items()})
    best = None
    for _ in range(5):                                         # host timestamps inside the phases, no extra syncs
        tm = fgp.ProofTimings(fine=True)
        t0 = time.perf_counter(); P2.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve), tm); ctx.synchronize(); t1 = time.perf_counter()
        if best is None or t1 - t0 < best[0]:
            best = (t1 - t0, t0, tm.ticks)
            shapes = tm.msm_shapes
    prev = best[1]
    print("   host timeline of the best of 5 (%.2f ms): label, ms since start, ms since previous" % (1e3 * best[0]))
    for label, t in best[2]:
        print("      %-28s %8.3f %8.3f" % (label, 1e3 * (t - best[1]), 1e3 * (t - prev)))
        prev = t
    print("   commitment MSMs (columns, sorted pairs, points per lane, buckets merged by one lane or quad / 32 lanes / one wave / a block):")
    for sh in shapes:
        print("      %2d columns  %9d pairs  %3d per lane   %7d %6d %5d %4d" % (sh["columns"], sh["pairs"], sh["points_per_lane"], sh["merge_light"], sh["merge_32"],
                                                                                sh["merge_wave"], sh["merge_block"]))
    side.close()
    params.release()
