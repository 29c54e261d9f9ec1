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


@pytest.mark.timeout(300)
def test_all_gather_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for attempt in range(2):  # a probed-free port can be taken before the rendezvous binds it
        port = _free_port()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(script), ROOT]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=140)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


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
