#!/usr/bin/env python3
"""The workload of a batch-mode trace: `batch` native proofs (dehalo_create_proofs) on `provers` provers, preceded by a marker launch
so that the analysis can find the timed window.   rocprofv3 --kernel-trace ... -- python3 tools/batch_trace.py [k] [provers] [batch] [side]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO
from dehalo2_amd import prover, keygen, native
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
nprov = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
with_side = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
prio_mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # 1: contexts alternate between the highest / default / lowest stream priority
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
ctxs = [pkg.Context(0, priority=((1, 0, -1)[i % 3] if prio_mode else 0)) for i in range(nprov)]
sides = [pkg.Context(0) for _ in range(nprov)] if with_side else [None] * nprov
provers = [native.Prover(nparams, npk, c, s) for c, s in zip(ctxs, sides)]
native.create_proofs(provers, adv, [prover.SeededRng(1000 + i) for i in range(2 * nprov)])
import torch
torch.cuda.synchronize()
marker = torch.zeros(12345, dtype=torch.int64, device="cuda")      # a fill kernel of a recognisable grid marks the window's start
torch.cuda.synchronize()
t = time.perf_counter()
out = native.create_proofs(provers, adv, [prover.SeededRng(2000 + i) for i in range(batch)])
el = time.perf_counter() - t
torch.cuda.synchronize()
marker2 = torch.zeros(12345, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
print("batch of %d on %d provers%s%s [GPU_MAX_HW_QUEUES=%s]: %.1f proofs/s (%.3f ms per proof)" % (batch, nprov, " + side contexts" if with_side else "", " mixed priorities" if prio_mode else "", os.environ.get("GPU_MAX_HW_QUEUES", "default"), batch / el, 1e3 * el / batch))
