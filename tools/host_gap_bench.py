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
    _p8 = [a[i % 4].data_ptr() for i in range(8)]; _c8 = f.encode_many([3] * 8)
    T("lincomb_device(8) launch, prepared args", lambda: ctx.lincomb_device(f.id, _p8, _c8, 1 << 10, a[3].data_ptr(), None, 0))
    ctx.synchronize()
    T("lincomb_device(8) launch", lambda: ctx.lincomb_device(f.id, [a[i % 4].data_ptr() for i in range(8)], f.encode_many([3] * 8), 1 << 10, a[3].data_ptr(), None, 0))
    ctx.synchronize()
    rng = prover.SeededRng(5)
    T("rng.scalars(5*6)", lambda: rng.scalars(30))
    T("rng.scalars(5*21400) (advice blinding at k=17)", lambda: rng.scalars(5 * 21400), 20)
    v = rng.scalars(5 * 21400).reshape(5, 21400, 4)
    T("to_device(5 x 21400 rows)", lambda: keygen.to_device(v), 50)
    T("torch.cuda.Event record+wait", lambda: (lambda e: (e.record(ctx.torch_stream_obj()), ctx.torch_stream_obj().wait_event(e)))(torch.cuda.Event()))
    bases = co.synth_bases(curve.id, 1 << 11)
    h = ctx.register_bases(curve.id, bases, 0, True)
    sc = torch.zeros(5, 1 << 11, 4, dtype=torch.int64, device="cuda"); out = torch.zeros(5, 12, dtype=torch.int64, device="cuda"); aff = torch.zeros(5, 8, dtype=torch.int64, device="cuda")
    def msm_and_wait():
        ctx.msm_device_affine(h, sc.data_ptr(), 1 << 11, 5, 0, aff.data_ptr(), 0)
    import time as _t
    ts = []
    for _ in range(50):
        ctx.synchronize(); t = _t.perf_counter(); msm_and_wait(); ts.append(_t.perf_counter() - t); ctx.synchronize()
    print("%-46s %7.1f us (host time of the launches, device idle before)" % ("msm_device_affine(5 x 2^11) launch", 1e6 * sorted(ts)[len(ts) // 2]))
    ts = []
    for _ in range(50):
        ctx.synchronize(); t = _t.perf_counter(); msm_and_wait(); ctx.synchronize(); ts.append(_t.perf_counter() - t)
    print("%-46s %7.1f us (launch + device + wait)" % ("msm_device_affine(5 x 2^11) complete", 1e6 * sorted(ts)[len(ts) // 2]))
    om = f.encode(po.BN254.scalar.omega(11)) if hasattr(po.BN254, "scalar") else None
    if om is not None:
        ts = []
        for _ in range(50):
            ctx.synchronize(); t = _t.perf_counter(); ctx.ntt_device(f.id, a[0].data_ptr(), 11, om, 5, 0); ts.append(_t.perf_counter() - t); ctx.synchronize()
        print("%-46s %7.1f us (host time of the launches)" % ("ntt_device(5 x 2^11) launch", 1e6 * sorted(ts)[len(ts) // 2]))
