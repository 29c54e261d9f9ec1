#!/usr/bin/env python3
"""One lone transform (default pasta::Fp 2^20) and the prover's batch (23 x 2^19 bn256::Fr): min of 30 timings, then -- with DEHALO_NTT_STAMPS=1 in the
environment -- one more call whose workgroups stamp their phases (csrc/ntt.cuh: ntt_report_stamps prints them on stderr).
   python tools/ntt_phases.py [field log_n batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package()
ctx = pkg.Context(0)
cases = [("pasta_fp", 20, 1), ("bn254_fr", 19, 23), ("bn254_fr", 17, 22)]
if len(sys.argv) > 3: cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))]
rng = np.random.default_rng(1)
for fname, log_n, batch in cases:
    f = pkg.fields.FIELDS[fname]
    a = rng.integers(0, 1 << 62, size=(batch << log_n, 4), dtype=np.int64); a[:, 3] &= (1 << 60) - 1
    d = ctx.upload(a.reshape(-1))
    om = f.encode(pow(f.root_of_unity, 1 << (f.two_adicity - log_n), f.p))
    if os.environ.get("DEHALO_NTT_STAMPS"):
        ctx.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0); ctx.synchronize()
        continue
    for _ in range(60): ctx.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0)
    ctx.synchronize()
    ts = []
    for _ in range(30):
        t = time.perf_counter(); ctx.ntt_device(f.id, d.data_ptr(), log_n, om, batch, 0); ctx.synchronize(); ts.append(time.perf_counter() - t)
    print("%s 2^%d x %d: min %.4f ms, median %.4f ms" % (fname, log_n, batch, 1e3 * min(ts), 1e3 * sorted(ts)[15]), flush=True)
