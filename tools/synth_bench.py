"""dehalo_synthesize (k = 17 delay_enc witness, 15-bit exponent) by DEHALO_SYNTH_THREADS: time and a digest of the advice columns."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from dehalo2_amd import native
import json
v = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'rsa_vectors.json') ))[1]
n, x = int(v["n"]), int(v["signature"])
e = 0b101101110010111
buf = np.empty((5, 1<<17, 4), dtype=np.uint64)
for _ in range(3): out = native.synthesize(native.CIRCUIT_DELAY_ENC if hasattr(native,'CIRCUIT_DELAY_ENC') else 0, 17, n_big=n, e=e, x=x, exp_bits=15, message=[0,0], key=[3,4], out=buf)
ts=[]
for _ in range(20):
    t=time.perf_counter(); out = native.synthesize(native.CIRCUIT_DELAY_ENC if hasattr(native,'CIRCUIT_DELAY_ENC') else 0, 17, n_big=n, e=e, x=x, exp_bits=15, message=[0,0], key=[3,4], out=buf); ts.append(time.perf_counter()-t)
print(os.environ.get("DEHALO_SYNTH_THREADS"), "min %.3f ms median %.3f ms" % (1e3*min(ts), 1e3*sorted(ts)[10]), out["rows"], out["rsa_result"] == pow(x, e, n))
import hashlib; print(hashlib.sha256(buf.tobytes()).hexdigest()[:16])
