#!/usr/bin/env python3
"""Host-side cost of one phase boundary of create_proof (after the device has finished): D2H of the commitments, decode, transcript,
challenge, and the per-call cost of the launches that follow.  python tools/host_gap_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
from dehalo2_amd import keygen, transcript, prover
curve = pkg.fields.BN254; f = curve.scalar
ctx = pkg.Context(0)
def T(name, fn, reps=300):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    print("%-46s %7.1f us" % (name, 1e6 * (time.perf_counter() - t) / reps))
with ctx.torch_stream():
    aff = torch.zeros(8, 8, dtype=torch.int64, device="cuda")
    pts = co.synth_bases(curve.id, 8)
    aff.copy_(torch.from_numpy(pts.view(np.int64).reshape(8, 8)))
    ctx.synchronize()
    T("ctx.synchronize (idle)", lambda: ctx.synchronize())
    T("to_host(aff[:5])", lambda: keygen.to_host(aff[:5]))
    h = keygen.to_host(aff[:5])
    T("decode_points(5)", lambda: keygen.decode_points(curve, h))
    P = keygen.decode_points(curve, h)
    tr = transcript.Blake2bWrite(curve)
    T("write_point x5", lambda: [tr.write_point(p) for p in P])
    T("squeeze_challenge_scalar", lambda: tr.squeeze_challenge_scalar())
    T("f.encode", lambda: f.encode(123456789 << 200))
    T("f.encode_many(10)", lambda: f.encode_many([123456789 << 200] * 10))
    a = torch.zeros(4, 1 << 17, 4, dtype=torch.int64, device="cuda")
    T("field_op_device launch", lambda: ctx.field_op_device(f.id, "add", a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), 1 << 10, 0))
    ctx.synchronize()
    T("scale_device launch", lambda: ctx.scale_device(f.id, a[0].data_ptr(), 1 << 10, None, a[1].data_ptr(), 0))
    ctx.synchronize()
    T("lincomb_device(8) launch", lambda: ctx.lincomb_device(f.id, [a[i % 4].data_ptr() for i in range(8)], f.encode_many([3] * 8), 1 << 10, a[3].data_ptr(), None, 0))
    ctx.synchronize()
    rng = prover.SeededRng(5)
    T("rng.scalars(5*6)", lambda: rng.scalars(30))
    T("rng.scalars(5*21400) (advice blinding at k=17)", lambda: rng.scalars(5 * 21400), 20)
    v = rng.scalars(5 * 21400).reshape(5, 21400, 4)
    T("to_device(5 x 21400 rows)", lambda: keygen.to_device(v), 50)
    T("torch.cuda.Event record+wait", lambda: (lambda e: (e.record(ctx.torch_stream_obj()), ctx.torch_stream_obj().wait_event(e)))(torch.cuda.Event()))
    T("tensor slice copy_ (5 x 2^17 x 32 B) launch", lambda: a[:3].copy_(a[1:4]), 50)
    ctx.synchronize()
