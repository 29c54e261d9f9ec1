#!/usr/bin/env python3
"""create_proof latency against the window size of the resident SRS tables.   python tools/sweep_proof_window.py k range_lookups c0 c1"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import fine_grained_prover as fgp
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, transcript
import bench
k, rl, c0, c1 = int(sys.argv[1]), bool(int(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, "delay_enc" if rl else "pose_enc")
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
for c in [0] + list(range(c0, c1 + 1)):
    with pkg.Context(0) as ctx, pkg.Context(0) as side:
        params = keygen.ParamsKZG(ctx, curve, k, srs["g"], srs["g_lagrange"], window_bits=c)
        pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
        P = fgp.Prover(params, pk, side_ctx=side)
        with ctx.torch_stream():
            adv = keygen.to_device(circ.advice)
            ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
        ctx.synchronize()
        ts = []
        for _ in range(12):
            t = time.perf_counter(); P.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve)); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
        ts = ts[2:]
        print("k %d window_bits %2d (%d windows): best %.3f ms  median %.3f ms" % (k, params.bases_g.window_bits, params.bases_g.windows, min(ts), sorted(ts)[len(ts) // 2]))
        params.release()
