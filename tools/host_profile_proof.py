#!/usr/bin/env python3
"""cProfile of the host side of create_proof (k = 17 delay_enc witness, side context): where the Python time goes."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import fine_grained_prover as fgp
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, transcript
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, "delay_enc")
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
with pkg.Context(0) as ctx, pkg.Context(0) as side:
    params = keygen.ParamsKZG(ctx, curve, k, srs["g"], srs["g_lagrange"])
    pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    P = fgp.Prover(params, pk, side_ctx=side)
    with ctx.torch_stream():
        adv = keygen.to_device(circ.advice)
        ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
    ctx.synchronize()
    for _ in range(3):
        P.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve))
    pr = cProfile.Profile()
    reps = 10
    t = time.perf_counter()
    pr.enable()
    for _ in range(reps):
        P.create_proof(adv, [[]], prover.SeededRng(7), transcript.Blake2bWrite(curve))
    pr.disable()
    print("%.2f ms per proof under cProfile (x%d)" % (1e3 * (time.perf_counter() - t) / reps, reps))
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
