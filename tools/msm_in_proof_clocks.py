"""Is a lone proof's accumulation slower than the same launch in a loop because of the clocks?  One process, one box: a 3-column 2^17 BN254 MSM (the shape of a k = 17
proof's openings commitment) timed (a) in a warm loop of its own, (b) once after every k = 17 proof of a back-to-back series, (c) once after every proof with 3 ms of host
sleep in front of the proof.   python tools/msm_in_proof_clocks.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import bench
from dehalo2_amd import _lib
ctx = pkg.Context(0)
st = bench.ProofSetup(pkg, ctx, 17, "delay_enc", 16)
curve = pkg.fields.BN254
n, b = 1 << 17, 3
c2 = pkg.Context(0)
h = c2.register_bases(curve.id, co.synth_bases(curve.id, n), 0, True)
d = c2.upload(np.concatenate([co.fill_scalars(curve.scalar.id, "uniform", n, 100 + j) for j in range(b)]))
out = torch.zeros((b, 12), dtype=torch.int64, device="cuda")
run = lambda: c2.msm_device(h, d.data_ptr(), n, b, out.data_ptr(), 0)
def acc(f, reps):
    c2.timing_reset(); c2.timing_enable(True)
    for _ in range(reps): f()
    c2.synchronize(); c2.timing_enable(False)
    return 1e3 * c2.timing_get(_lib.K_MSM_ACCUMULATE)[0] / reps
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5: run()
c2.synchronize()
print("(a) warm loop of the MSM alone: accumulate %.1f us" % acc(run, 60))
for i in range(10): st.prove(7)
def proof_then_msm():
    st.prove(7); run(); c2.synchronize()
print("(b) one MSM after every proof of a back-to-back series: accumulate %.1f us" % acc(proof_then_msm, 30))
def sleepy():
    time.sleep(0.003); st.prove(7); run(); c2.synchronize()
print("(c) the same with 3 ms of sleep before every proof: accumulate %.1f us" % acc(sleepy, 30))
print("(a) again: accumulate %.1f us" % acc(run, 60))
