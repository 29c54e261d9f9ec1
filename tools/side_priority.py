#!/usr/bin/env python3
"""Single-proof latency with the side context at the lowest / default / highest stream priority, with and without other (idle) contexts alive in the
process (stream -> hardware-queue collisions).   python tools/side_priority.py [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, native
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, "delay_enc")
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
ctx = pkg.Context(0)
with ctx.torch_stream():
    adv = keygen.to_device(circ.advice)
    ctx.field_op_device(curve.scalar.id, "to_mont", adv.data_ptr(), 0, adv.data_ptr(), adv.numel() // 4, 0)
ctx.synchronize()
nparams = native.ParamsKZG.create(ctx, curve, k, srs["g"], srs["g_lagrange"])
npk = native.ProvingKey.keygen(ctx, nparams, circ.cs, circ.fixed, circ.assembly, circ.selectors)
def run(label, main_prio, side_prio, extra):
    others = [pkg.Context(0) for _ in range(extra)]
    m, s = pkg.Context(0, priority=main_prio), pkg.Context(0, priority=side_prio)
    N = native.Prover(nparams, npk, m, s)
    for _ in range(4): N.create_proof(adv, [[]], prover.SeededRng(7))
    ts = []
    for _ in range(12):
        t = time.perf_counter(); N.create_proof(adv, [[]], prover.SeededRng(7)); ts.append(1e3 * (time.perf_counter() - t))
    print("%-46s extra idle contexts %d: min %.3f median %.3f ms" % (label, extra, min(ts), sorted(ts)[len(ts) // 2]))
    N.release(); m.close(); s.close()
    for o in others: o.close()
for extra in (0, 1, 2, 3, 5):
    run("main default, side default", 0, 0, extra)
    run("main default, side lowest", 0, -1, extra)
    run("main default, side highest", 0, 1, extra)
    run("main highest, side default", 1, 0, extra)
