"""CPU suite: the multi-GPU layer with gloo, world_size 2.  Units (proofs /
columns) are dealt round-robin to ranks; the only collective is the all-gather of the
commitment vector."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
import torch, torch.distributed as dist
pkg = entry.load_package()
from dehalo2_amd import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
units = 7
mine = sharding.units_for_rank(units, rank, world)
assert mine == list(range(rank, units, world))
# fake 12-word "commitments": unit u -> row filled with u + 1
local = torch.zeros((len(mine), 12), dtype=torch.int64)
for i, u in enumerate(mine):
    local[i] = u + 1
full = sharding.all_gather_commitments(local, units, rank, world)
assert full.shape == (units, 12)
for u in range(units):
    assert int(full[u, 0]) == u + 1 and int(full[u, 11]) == u + 1
# batch proving: proof p made by rank p mod world, every rank ends with all commitment blobs in proof order
from dehalo2_amd import plonk, prover
cs = plonk.maingate_cs(True)
head, evals = prover.proof_layout(cs)
assert (head, evals) == (27, 58)
total = 5
fake = lambda p: bytes([p + 1]) * (32 * (head + evals + 4))         # a stand-in proof: 27 + 4 points, 58 scalars (2848 bytes, as the real one)
blobs = [prover.proof_commitments(cs, fake(p)) for p in sharding.units_for_rank(total, rank, world)]
assert all(len(b) == 32 * 31 for b in blobs)
got = sharding.gather_proof_commitments(blobs, total, rank, world)
assert got == [bytes([p + 1]) * (32 * 31) for p in range(total)]
# one proof sharded by commitment columns: the exchange dehalo_prover_set_shard's callback makes (every count the prover has: 1 .. 12 columns a phase)
import numpy as np
for count in (2, 3, 5, 10, 12, 1):
    first = [sharding.column_range_for_rank(count, r, world)[0] for r in range(world)]
    num = [sharding.column_range_for_rank(count, r, world)[1] for r in range(world)]
    assert sum(num) == count and first[0] == 0 and all(first[r] + num[r] == (first[r + 1] if r + 1 < world else count) for r in range(world))
    pts = np.zeros((count, 8), dtype=np.uint64)
    pts[first[rank]:first[rank] + num[rank]] = (np.arange(first[rank], first[rank] + num[rank], dtype=np.uint64)[:, None] + 1) * np.uint64(0x0101010101010101)
    sharding.gather_points(pts, first, num, rank, world)
    assert np.array_equal(pts, (np.arange(count, dtype=np.uint64)[:, None] + 1) * np.uint64(0x0101010101010101) * np.ones((1, 8), dtype=np.uint64)), (count, pts)
# max-over-ranks timing helper
t = sharding.max_over_ranks(float(rank + 1))
assert t == float(world)
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("rank %d ok\n" % rank)   # one write: two ranks share the pipe
sys.stdout.flush()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_units_for_rank(pkg):
    import sys as _s
    sharding = __import__("importlib").import_module("dehalo2_amd.sharding")
    assert sharding.units_for_rank(64, 3, 8) == list(range(3, 64, 8))
    assert sharding.units_for_rank(5, 7, 8) == []
    got = sorted(u for r in range(8) for u in sharding.units_for_rank(31, r, 8))
    assert got == list(range(31))


def test_point_ranges_cover_without_overlap(pkg):
    sharding = __import__("importlib").import_module("dehalo2_amd.sharding")
    for n, world in ((10, 4), (1 << 20, 8), (7, 8), (0, 3), (5003, 4)):
        ranges = [sharding.point_range_for_rank(n, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world", [2, 8])
def test_all_gather_over_gloo(tmp_path, world):
    """world 2, and the node's own size 8 (rendezvous, every collective of the layer and the column-shard exchange at eight ranks -- on the CPU: the pool allows
    six processes on a card, the launcher included, so an eight-rank rehearsal cannot touch the GPU)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for attempt in range(2):  # a probed-free port can be taken before the rendezvous binds it
        port = _free_port()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(script), ROOT]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=190)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout + r.stderr
    assert all("rank %d ok" % i in r.stdout for i in range(world))


# ---- RCCL on the box's one GPU: a one-rank `nccl` process group, HBM tensors through the real collectives ---------------------------
NCCL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import __graft_entry__ as entry
import torch, torch.distributed as dist
pkg = entry.load_package()
po, co = entry.load_oracle()
from dehalo2_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
ctx = pkg.Context(0)
# (1) the commitment vector of the step bench: (units, 12) int64 rows in HBM through all_gather_into_tensor
local = torch.arange(7 * 12, dtype=torch.int64, device="cuda").reshape(7, 12) + 5
full = sharding.all_gather_commitments(local, 7, 0, 1, force_collective=True)
torch.cuda.synchronize()
assert full.is_cuda and torch.equal(full, local)
# (2) batch mode's gather of compressed commitments (31 x 32 B per proof), device tensors, width agreed by an all_reduce
blobs = [bytes([p + 1]) * (32 * 31) for p in range(5)]
got = sharding.gather_proof_commitments(blobs, 5, 0, 1, "cuda", force_collective=True)
assert got == blobs
# (3) one MSM split by point range: the library's result rows are gathered by RCCL, then summed on the context's own stream
curve = pkg.fields.CURVES["bn254"]
n = 3001
g = co.synth_bases(curve.id, n)
sc = co.fill_scalars(curve.scalar.id, "witness", n, 31)
h = ctx.register_bases(curve.id, g, 0, True)
d = ctx.upload(sc)
part = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ctx.msm_device(h, d.data_ptr(), n, 1, part.data_ptr(), 0)
ctx.synchronize()
whole = sharding.combine_partial_msm(ctx, curve.id, part, 0, 1, force_collective=True)
want = co.to_affine(curve.id, co.best_multiexp(curve.id, sc, g, 4))
assert np.array_equal(ctx.to_affine(curve.id, ctx.download_tensor(whole))[0], want)
# (4) the timing helper's all_reduce(MAX) on a device scalar
assert sharding.max_over_ranks(3.25) == 3.25
h.release()
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
ctx.close()
sys.stdout.write("rccl one-rank ok: %s\\n" % torch.cuda.nccl.version().__repr__())
sys.stdout.flush()
'''


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_one_rank_all_gather_on_hbm_tensors(tmp_path):
    """RCCL executes: a one-rank `nccl` group on the box's GPU (a one-GPU box cannot host two nccl ranks), the world-1 shortcut of sharding.py
    switched off, device tensors through all_gather_into_tensor / all_reduce, and the stream ordering of combine_partial_msm (the gather on torch's
    stream, the addition on the context's).  A child process of its own: one rank = one process, as under torchrun."""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rccl one-rank ok" in r.stdout


# ---- ONE proof sharded by commitment columns over two processes on the box's GPU (SURVEY.md 8(e) single-proof mode; dehalo_prover_set_shard) ---------------
SHARD_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import __graft_entry__ as entry
import torch, torch.distributed as dist
pkg = entry.load_package()
from dehalo2_amd import native, plonk, prover, sharding, circuits
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
ctx, side = pkg.Context(0), pkg.Context(0)
curve = pkg.fields.BN254
out = {}
for k, kind in ((11, "synthetic"), (17, "delay_enc")):
    if kind == "synthetic":
        circ = circuits.synthesize(curve.scalar.p, k, True, seed=3)
        fixed, asm, selectors, advice, canonical = circ.fixed, circ.assembly, circ.selectors, circ.advice, True
    else:
        v = json.load(open(os.path.join(sys.argv[1], "tests", "golden", "rsa_vectors.json")))[1]
        nat = native.synthesize(native.CIRCUIT_DELAY_ENC, k, n_big=int(v["n"]), e=0b101101110010111, x=int(v["signature"]), exp_bits=15, message=[0, 0], keygen=True)
        assert nat["rows"] == 125214
        fixed, selectors, advice, canonical = nat["fixed"], nat["selectors"], nat["advice"], True
        asm = plonk.Assembly(6, 1 << k)
        asm.mapping = nat["mapping"].astype(np.int64)
    cs = plonk.maingate_cs(True)
    params = native.ParamsKZG.setup(ctx, curve, k, 0x5EED5EED5EED5EED)
    pk = native.ProvingKey.keygen(ctx, params, cs, fixed, asm, selectors)
    pk.transcript_repr = 12345
    P = native.Prover(params, pk, ctx, side)
    lone = P.create_proof(advice, [[]], prover.SeededRng(5), canonical=canonical).finalize()
    calls = []
    def gather(points, first, num):
        calls.append((points.shape[0], num[rank]))
        sharding.gather_points(points, first, num, rank, world)
    P.set_shard(rank, world, gather)
    sharded = P.create_proof(advice, [[]], prover.SeededRng(5), canonical=canonical).finalize()
    assert sharded == lone, "k = %d: the sharded proof differs from the lone prover's" % k
    # every multi-column phase went through the exchange: advice (5), permuted columns (10), products (2 sets + 5 lookups), quotient pieces, openings
    counts = [c for c, _ in calls]
    assert counts[:2] == [5, 10] and counts[2] in (7, 8) and len(counts) == 5 and all(c >= 2 for c in counts), counts
    assert sum(m for _, m in calls) < sum(counts)      # this rank ran a strict share of the MSM columns
    P.set_shard(0, 1)
    assert P.create_proof(advice, [[]], prover.SeededRng(5), canonical=canonical).finalize() == lone
    # all ranks hold the same bytes
    digest = torch.tensor(list(__import__("hashlib").sha256(sharded).digest()), dtype=torch.int64)
    allg = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(allg, digest)
    assert all(torch.equal(a, digest) for a in allg)
    out[k] = (len(sharded), counts)
    P.release(); pk.release(); params.release()
dist.barrier()
dist.destroy_process_group()
side.close(); ctx.close()
sys.stdout.write("rank %d sharded proofs ok %r\\n" % (rank, out))
sys.stdout.flush()
'''


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_one_proof_sharded_by_columns_over_two_processes(tmp_path):
    """SURVEY.md 8(e) / north star "per-column commitments": two processes (gloo; both on the box's one GPU) prove the SAME circuit; each runs the MSMs of its
    share of every multi-column commitment phase and exchanges at most ten affine points a phase through the callback -- the proof bytes on both equal the lone
    prover's, at k = 11 (synthetic MainGate + RangeChip circuit) and at the metric's own k = 17 delay_enc witness (125,214 rows).  A correctness path: no speed
    is claimed for it here (one GPU); DESIGN.md section 7 prices it for eight."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    for attempt in range(2):
        port = _free_port()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(script), ROOT]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
        if r.returncode == 0 or "sharded proof differs" in r.stdout + r.stderr:
            break
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0 sharded proofs ok" in r.stdout and "rank 1 sharded proofs ok" in r.stdout
