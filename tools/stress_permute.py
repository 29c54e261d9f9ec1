#!/usr/bin/env python3
"""Randomised permute_expression_pair batches against the C restatement: theta-compressed style values (tag * theta + small),
plain integers of every size (the fast path's verification must send them to the full sort), uniform values, constant columns,
mixed in one batch.   python tools/stress_permute.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pkg.Context(0)
for trial in range(trials):
    f = [pkg.fields.BN254_FR, pkg.fields.PASTA_FP][trial % 2]
    p = f.p
    n = int(rng.integers(2, 5000))
    B = int(rng.choice([1, 2, 5]))
    stride = n + int(rng.integers(0, 40))
    ins, tabs = np.zeros((B, stride, 4), dtype=np.uint64), np.zeros((B, stride, 4), dtype=np.uint64)
    for y in range(B):
        tsize = int(rng.integers(1, min(n, 600) + 1))
        kind = int(rng.integers(0, 5))
        if kind == 0:      # tag * theta + value
            theta = int(rng.integers(1, 1 << 62)) * (1 << 190) % p
            vals = [(int(rng.integers(0, 4)) * theta + int(rng.integers(0, 1 << 16))) % p for _ in range(tsize)]
        elif kind == 1:    # plain integers up to 2^bits (bits anywhere between 8 and 250)
            bits = int(rng.integers(8, 250))
            vals = [int.from_bytes(rng.bytes(32), "little") % (1 << bits) for _ in range(tsize)]
        elif kind == 2:    # uniform
            vals = [int.from_bytes(rng.bytes(40), "little") % p for _ in range(tsize)]
        elif kind == 3:    # values equal in the low 24 and the leading bits, different in between
            hi, lo = int(rng.integers(1, 1 << 30)) << 220, int(rng.integers(0, 1 << 24))
            vals = [(hi + (int(rng.integers(0, 1 << 60)) << 64) + lo) % p for _ in range(tsize)]
        else:
            vals = [int(rng.integers(0, 3))] * tsize
        base = f.encode_many(vals)
        tabs[y, :n] = np.concatenate([base, np.repeat(base[:1], n - tsize, axis=0)])
        ins[y, :n] = base[rng.integers(0, tsize, size=n)]
    with ctx.torch_stream():
        di, dt = torch.from_numpy(ins.view(np.int64)).cuda(), torch.from_numpy(tabs.view(np.int64)).cuda()
        oi, ot = torch.zeros_like(di), torch.zeros_like(dt)
        ctx.permute_expression_pair_batch_device(f.id, di.data_ptr(), dt.data_ptr(), n, B, stride, oi.data_ptr(), ot.data_ptr(), 0)
        gi, gt = oi.cpu().numpy().view(np.uint64), ot.cpu().numpy().view(np.uint64)
    for y in range(B):
        want = co.permute_expression_pair(f.id, ins[y], tabs[y], n)
        assert want is not None
        assert np.array_equal(gi[y, :n], want[0]) and np.array_equal(gt[y, :n], want[1]), (trial, y, n, B)
print("%d randomised permute_expression_pair batches: all equal to the C restatement" % trials)
