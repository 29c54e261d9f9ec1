#!/usr/bin/env python3
"""Development check: the device create_proof against the CPU restatement (oracle/plonk_oracle.py) and the verifier.
    python tools/proof_check.py [k range_lookups(0|1)] ..."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import fine_grained_prover as fgp

pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO, verifier as V, pairing as pr
from dehalo2_amd import plonk, circuits, prover, keygen, transcript

def run(ctx, k, rl, verify=True, threads=16):
    curve, ocurve = pkg.fields.BN254, po.BN254
    circ = circuits.synthesize(curve.scalar.p, k, rl, seed=3)
    import shapes
    desc = shapes.maingate_description(bool(circ.cs.lookups))
    s = 0x1234567890abcdef1234567890abcdef
    t = time.time(); srs = PO.setup_srs(ocurve, k, s, threads); print("srs", round(time.time() - t, 2), flush=True)
    params = keygen.ParamsKZG(ctx, curve, k, srs["g"], srs["g_lagrange"], pr.g2_to_raw(pr.G2), pr.g2_to_raw(pr.g2_mul(s, pr.G2)))
    t = time.time(); pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors); print("gpu keygen", round(time.time() - t, 2), flush=True)
    t = time.time(); key = PO.keygen(ocurve, srs, desc, k, circ.fixed, circ.assembly.mapping, threads); print("cpu keygen", round(time.time() - t, 2), flush=True)
    rep = PO.transcript_repr(ocurve, key, circ.selectors)
    assert keygen.decode_points(curve, pk.vk.fixed_commitments) == key["fixed_commitments"], "fixed commitments differ"
    assert keygen.decode_points(curve, pk.vk.permutation_commitments) == key["perm_commitments"], "permutation commitments differ"
    buf = io.BytesIO(); pk.vk.write(buf)
    assert buf.getvalue() == PO.vk_bytes(ocurve, key, circ.selectors), "vk bytes differ"
    assert rep == pk.vk.transcript_repr, "vk transcript_repr differs"
    F = PO.Fld(ocurve.scalar)
    adv = np.stack([co.field_op(F.id, "to_mont", circ.advice[i]) for i in range(5)])
    P = fgp.Prover(params, pk)
    tr = transcript.Blake2bWrite(curve)
    tm = fgp.ProofTimings()
    P.create_proof(adv, [[]], prover.SeededRng(7), tr, tm)
    proof = tr.finalize()
    t = time.time(); want, trace = PO.create_proof(ocurve, srs, key, adv, [[]], PO.ScalarStream(7), rep, threads); cpu_s = time.time() - t
    print("k", k, "lookups", rl, "proof bytes", len(proof), "identical to oracle:", proof == want, "cpu_s", round(cpu_s, 2), flush=True)
    if proof != want:
        for i in range(0, min(len(proof), len(want)), 32):
            if proof[i:i + 32] != want[i:i + 32]:
                print("first difference at item", i // 32); break
    best = 1e9
    for _ in range(3):
        tr2 = transcript.Blake2bWrite(curve)
        t = time.perf_counter(); P.create_proof(adv, [[]], prover.SeededRng(7), tr2); ctx.synchronize(); best = min(best, time.perf_counter() - t)
        assert tr2.finalize() == proof
    print("gpu create_proof ms", round(1e3 * best, 2), {k_: round(v, 2) for k_, v in tm.phases_ms.items()}, flush=True)
    if verify:
        ok = V.verify_proof(ocurve, desc, k, key["fixed_commitments"], key["perm_commitments"], rep, (1, 2), pr.G2, pr.g2_mul(s, pr.G2), [[]], proof)
        print("verifier accepts:", ok, flush=True)
        assert ok
    assert proof == want
    params.release()

if __name__ == "__main__":
    args = sys.argv[1:] or ["6", "0", "9", "1"]
    with pkg.Context(0) as ctx:
        for i in range(0, len(args), 2):
            run(ctx, int(args[i]), bool(int(args[i + 1])))
