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
