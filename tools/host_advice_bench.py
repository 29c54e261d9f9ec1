#!/usr/bin/env python3
"""create_proof from a witness in HOST memory: a fresh pageable array per proof (what a caller that allocates per proof hands over), one pageable
array kept across proofs, and a page-locked buffer; keygen and pk_read from host arrays beside it.   python tools/host_advice_bench.py [k=17] [reps=30]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, native
import bench, torch
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
curve = pkg.fields.BN254
circ, desc, _ = bench.real_witness(curve.scalar.p, k, "delay_enc")
srs = PO.setup_srs(po.BN254, k, 0x1234567890abcdef, 16)
ctx, side = pkg.Context(0, priority=1), pkg.Context(0, priority=-1)
params = native.ParamsKZG.create(ctx, curve, k, srs["g"], srs["g_lagrange"])
t = time.perf_counter(); pk = native.ProvingKey.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors); t_keygen = time.perf_counter() - t
t = time.perf_counter(); pk = native.ProvingKey.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors); t_keygen2 = time.perf_counter() - t
print("keygen from host arrays: %.1f ms first, %.1f ms second" % (1e3 * t_keygen, 1e3 * t_keygen2))
N = native.Prover(params, pk, ctx, side)
host = np.ascontiguousarray(circ.advice, dtype=np.uint64)           # canonical values, (5, n, 4): converted on the device inside the call
dev = torch.from_numpy(host.view(np.int64)).cuda()
want = N.create_proof(dev, [[]], prover.SeededRng(7), canonical=True).finalize()
pinned = torch.empty(host.shape, dtype=torch.int64).pin_memory()
pinned.numpy()[...] = host.view(np.int64)
def fresh():
    a = host.copy()                                                  # a new allocation every proof
    return N.create_proof(a, [[]], prover.SeededRng(7), canonical=True).finalize()
def kept():
    return N.create_proof(host, [[]], prover.SeededRng(7), canonical=True).finalize()
def locked():
    return N.create_proof(pinned.numpy().view(np.uint64), [[]], prover.SeededRng(7), canonical=True).finalize()
def resident():
    return N.create_proof(dev, [[]], prover.SeededRng(7), canonical=True).finalize()
t = time.perf_counter(); [host.copy() for _ in range(10)]; t_copy = (time.perf_counter() - t) / 10
for rnd in (1, 2):
    for name, fn in (("witness resident in HBM", resident), ("page-locked host buffer", locked), ("one pageable array, kept", kept), ("fresh pageable array per proof", fresh)):
        for _ in range(5): assert fn() == want
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t))
        print("round %d  %-32s min %.3f ms, median %.3f ms%s" % (rnd, name, min(ts), sorted(ts)[len(ts) // 2], "  (includes the %.2f ms of the array copy)" % (1e3 * t_copy) if fn is fresh else ""))
