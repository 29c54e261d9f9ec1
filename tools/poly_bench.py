#!/usr/bin/env python3
"""Device-resident timings of the field-vector primitives beside the C
oracle on the host cores.  Prints one line per (op, size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
ctx = pkg.Context(0)
threads = min(64, os.cpu_count() or 1)

def timeit(fn, reps=20):
    """median of per-call times, each call waited for (a loop average once showed 1.9 ms for one size whose calls take 0.08 ms:
    torch's allocator was still freeing the previous size's buffers behind the first calls -- tools/eval_outlier.py)"""
    import torch
    for _ in range(3): fn()
    ctx.synchronize(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3

def cpu(fn, reps=3):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3

for fname in ("bn254_fr", "pasta_fp"):
    fid = pkg.fields.FIELDS[fname].id
    for log_n in (14, 17, 20, 22):
        n = 1 << log_n
        a = co.fill_scalars(fid, "uniform", n, 1); b = co.fill_scalars(fid, "uniform", n, 2)
        x = co.fill_scalars(fid, "uniform", 1, 3)[0]
        da = torch.from_numpy(a.view(np.int64).copy()).cuda(); db = torch.from_numpy(b.view(np.int64).copy()).cuda()
        dz = torch.zeros_like(da); dout = torch.zeros((1, 4), dtype=torch.int64, device="cuda")
        g_eval = timeit(lambda: ctx.eval_polynomial_device(fid, da.data_ptr(), n, n, 1, x, dout.data_ptr(), 0))
        g_inv = timeit(lambda: ctx.batch_invert_device(fid, db.data_ptr(), n, 0))     # inverts back and forth: same cost
        g_gp = timeit(lambda: ctx.grand_product_device(fid, da.data_ptr(), db.data_ptr(), n, dz.data_ptr(), 0))
        do_cpu = log_n <= 20
        c_eval = cpu(lambda: co.eval_polynomial(fid, a, x, threads)) if do_cpu else float("nan")
        c_inv = cpu(lambda: co.batch_invert(fid, b)) if do_cpu else float("nan")
        c_gp = cpu(lambda: co.grand_product(fid, a, b)) if do_cpu else float("nan")
        # lookup permutation: a 2^12-entry table padded with its first value, inputs drawn from it
        tsize = min(n, 1 << 12)
        table = np.concatenate([a[:tsize], np.repeat(a[:1], n - tsize, axis=0)])
        inputs = a[:tsize][np.random.default_rng(1).integers(0, tsize, size=n)]
        dt, di = torch.from_numpy(table.view(np.int64).copy()).cuda(), torch.from_numpy(inputs.view(np.int64).copy()).cuda()
        dpi, dpt = torch.zeros_like(dt), torch.zeros_like(dt)
        g_lp = timeit(lambda: ctx.permute_expression_pair_device(fid, di.data_ptr(), dt.data_ptr(), n, dpi.data_ptr(), dpt.data_ptr(), 0), reps=10)
        c_lp = cpu(lambda: co.permute_expression_pair(fid, inputs, table, n)) if do_cpu else float("nan")
        gbs = lambda bytes_per, ms: bytes_per * n / ms / 1e6
        print("%-8s 2^%-2d eval_polynomial %7.3f ms (%6.1f GB/s alg 32 B/coef; cpu %d thr %8.2f ms) | batch_invert %7.3f ms (%6.1f GB/s alg 64 B; cpu 1 thr %8.2f ms) | "
              "grand_product %7.3f ms (%6.1f GB/s alg 96 B; cpu 1 thr %8.2f ms) | permute_expression_pair %7.3f ms (cpu 1 thr %8.2f ms)"
              % (fname, log_n, g_eval, gbs(32, g_eval), threads, c_eval, g_inv, gbs(64, g_inv), c_inv, g_gp, gbs(96, g_gp), c_gp, g_lp, c_lp), flush=True)
