#!/usr/bin/env python3
"""Generates tests/golden/*.json with oracle/pyoracle.py (Python big integers).

The reference holds no MSM/NTT vectors and cannot run here, so these are the
build's own definition-level vectors: NTT outputs from the O(n^2) DFT definition, MSM outputs
from per-term double-and-add.  Values are canonical integers in hex; the tests re-encode
them into halo2curves' Montgomery limbs.  Re-run: python tests/golden/make_golden.py"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyoracle as po

def hx(x): return hex(x)

def ntt_vectors():
    out = []
    for fname in ("bn254_fr", "pasta_fp", "pasta_fq"):
        f = po.FIELDS[fname]
        rng = po.Xoshiro(po.SEED ^ po.FIELD_IDS[fname])
        for log_n in (0, 1, 2, 3, 5, 8):
            n = 1 << log_n
            a = po.scalars_uniform(f, n, rng)
            w = f.omega(log_n)
            fwd = po.dft_naive(f, a, w)
            assert fwd == po.best_fft(f, a, w, log_n)
            out.append({"field": fname, "log_n": log_n, "omega": hx(w), "input": [hx(x) for x in a], "output": [hx(x) for x in fwd]})
    return out

def domain_vectors():
    out = []
    for fname, j in (("bn254_fr", 5), ("bn254_fr", 3), ("pasta_fp", 5), ("pasta_fq", 4)):
        f = po.FIELDS[fname]
        rng = po.Xoshiro(po.SEED + 17 * po.FIELD_IDS[fname] + j)
        for k in (2, 4):
            d = po.Domain(f, k, j)
            a = po.scalars_uniform(f, d.n, rng)
            coeffs = d.lagrange_to_coeff(a)
            ext = d.coeff_to_extended(coeffs)
            back = d.extended_to_coeff(ext)
            assert back[: d.n] == coeffs and all(x == 0 for x in back[d.n:])
            # definition check: ext[i] = poly(zeta * ext_omega^i)
            z = po.zeta(f)
            for i in (0, 1, len(ext) - 1):
                x = z * pow(d.ext_omega, i, f.p) % f.p
                assert ext[i] == sum(c * pow(x, e, f.p) for e, c in enumerate(coeffs)) % f.p
            out.append({"field": fname, "k": k, "j": j, "extended_k": d.extended_k, "zeta": hx(z), "lagrange": [hx(x) for x in a],
                        "coeffs": [hx(x) for x in coeffs], "extended": [hx(x) for x in ext], "back": [hx(x) for x in back]})
    return out

def msm_vectors():
    out = []
    for cname in ("bn254", "pallas", "vesta"):
        c = po.CURVES[cname]
        fs = c.scalar
        bases = po.synth_bases(c, 256)
        rng = po.Xoshiro(po.SEED ^ (0x1000 + po.CURVE_IDS[cname]))
        for n, dist in ((1, "uniform"), (2, "uniform"), (3, "witness"), (31, "uniform"), (32, "lookup"), (33, "witness"), (256, "uniform")):
            gen = {"uniform": po.scalars_uniform, "witness": po.scalars_witness_like, "lookup": po.scalars_lookup_like}[dist]
            s = gen(fs, n, rng)
            res = po.msm_naive(c, s, bases[:n])
            assert res == po.msm_pippenger(c, s, bases[:n])
            out.append({"curve": cname, "n": n, "dist": dist, "scalars": [hx(x) for x in s],
                        "result": None if res is None else [hx(res[0]), hx(res[1])]})
        # edge cases: 0, 1, r-1, 2^253, all-equal scalars, duplicated points, identity points, cancelling pair
        r = fs.p
        pts = list(bases[:8])
        pts[3] = pts[2]                      # duplicate point
        pts[5] = None                        # identity
        pts[7] = po.ec_neg(c, pts[6])        # P, -P
        sc = [0, 1, r - 1, 1 << 253, 5, 12345, 77, 77]
        res = po.msm_naive(c, sc, pts)
        out.append({"curve": cname, "n": 8, "dist": "edge", "scalars": [hx(x) for x in sc],
                    "points": [None if P is None else [hx(P[0]), hx(P[1])] for P in pts],
                    "result": None if res is None else [hx(res[0]), hx(res[1])]})
        sc = [r - 1] * 16                    # all-equal scalars
        res = po.msm_naive(c, sc, bases[:16])
        out.append({"curve": cname, "n": 16, "dist": "all_equal", "scalars": [hx(x) for x in sc],
                    "result": None if res is None else [hx(res[0]), hx(res[1])]})
        sc = [3, r - 3]                      # result = identity
        res = po.msm_naive(c, sc, [bases[0], bases[0]])
        assert res is None
        out.append({"curve": cname, "n": 2, "dist": "cancel", "scalars": [hx(x) for x in sc],
                    "points": [[hx(bases[0][0]), hx(bases[0][1])]] * 2, "result": None})
    return out

def bases_vectors():
    return {cname: [[hx(P[0]), hx(P[1])] for P in po.synth_bases(po.CURVES[cname], 256)] for cname in ("bn254", "pallas", "vesta")}

def poly_vectors():
    """eval_polynomial / batch_invert / grand_product, canonical hex."""
    out = []
    for fname in ("bn254_fr", "pasta_fp", "pasta_fq"):
        f = po.FIELDS[fname]
        rng = po.Xoshiro(0xF2 + po.FIELD_IDS[fname])
        for n in (0, 1, 5, 64, 300):
            poly = [rng.below(f.p) for _ in range(n)]
            for x in (0, 1, f.p - 1, rng.below(f.p)):
                out.append({"op": "eval_polynomial", "field": fname, "poly": [hx(c) for c in poly], "point": hx(x), "result": hx(po.eval_polynomial(f, poly, x))})
        vals = [0, 1, f.p - 1, 2, 0, 0] + [rng.below(f.p) for _ in range(120)] + [0]
        out.append({"op": "batch_invert", "field": fname, "values": [hx(v) for v in vals], "result": [hx(v) for v in po.batch_invert(f, vals)]})
        num = [rng.below(f.p) for _ in range(130)]
        den = [rng.below(f.p - 1) + 1 for _ in range(130)]
        out.append({"op": "grand_product", "field": fname, "num": [hx(v) for v in num], "den": [hx(v) for v in den], "result": [hx(v) for v in po.grand_product(f, num, den)]})
        den[40] = 0   # upstream's batch_invert leaves a zero denominator zero
        out.append({"op": "grand_product", "field": fname, "num": [hx(v) for v in num], "den": [hx(v) for v in den], "result": [hx(v) for v in po.grand_product(f, num, den)]})
    return out

def evalh_vectors():
    """Quotient-numerator kernels and the lookup permutation on tiny domains, canonical hex.
    The programs come from the same seeded generator the tests use (tests/conftest.py random_graph)."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from conftest import random_graph
    out = []
    for fname in ("bn254_fr", "pasta_fp"):
        f = po.FIELDS[fname]
        rng = po.Xoshiro(0xE5A1 + po.FIELD_IDS[fname])
        k, ext_k = 3, 5
        rows, rot_scale = 1 << ext_k, 1 << (ext_k - k)
        col = lambda: [rng.below(f.p) for _ in range(rows)]
        env = {"fixed": [col() for _ in range(2)], "advice": [col() for _ in range(3)], "instance": [col()], "challenges": [rng.below(f.p)],
               "beta": rng.below(f.p), "gamma": rng.below(f.p), "theta": rng.below(f.p), "y": rng.below(f.p)}
        for n_calcs, long_lived in ((8, 0), (30, 16)):
            g = random_graph(po, f, rng, n_calcs, 2, 3, 1, 1, long_lived)
            prev = col()
            res = po.graph_evaluate(f, g, env, rows, rot_scale, prev)
            hxl = lambda v: [hx(x) for x in v]
            out.append({"op": "graph", "field": fname, "ext_k": ext_k, "rot_scale": rot_scale,
                        "graph": {"constants": hxl(g["constants"]), "rotations": g["rotations"], "num_intermediates": g["num_intermediates"],
                                  "calcs": [[op, list(a), list(b), [list(q) for q in parts], t] for op, a, b, parts, t in g["calcs"]]},
                        "env": {"fixed": [hxl(c) for c in env["fixed"]], "advice": [hxl(c) for c in env["advice"]], "instance": [hxl(c) for c in env["instance"]],
                                "challenges": hxl(env["challenges"]), "beta": hx(env["beta"]), "gamma": hx(env["gamma"]), "theta": hx(env["theta"]), "y": hx(env["y"])},
                        "previous": hxl(prev), "result": hxl(res)})
        z, cols, sigma = [col() for _ in range(2)], [col() for _ in range(4)], [col() for _ in range(4)]
        l0, l_last, l_active, values = col(), col(), col(), col()
        beta, gamma, y, delta = (rng.below(f.p) for _ in range(4))
        zeta, w = po.zeta(f), f.omega(ext_k)
        res = po.permutation_h(f, values, z, cols, sigma, 3, -2, l0, l_last, l_active, beta, gamma, y, delta, zeta, w, rot_scale)
        hxl = lambda v: [hx(x) for x in v]
        out.append({"op": "permutation", "field": fname, "ext_k": ext_k, "rot_scale": rot_scale, "chunk_len": 3, "last_rotation": -2,
                    "z": [hxl(c) for c in z], "columns": [hxl(c) for c in cols], "sigma": [hxl(c) for c in sigma], "l0": hxl(l0), "l_last": hxl(l_last),
                    "l_active": hxl(l_active), "values": hxl(values), "beta": hx(beta), "gamma": hx(gamma), "y": hx(y), "delta": hx(delta), "zeta": hx(zeta),
                    "extended_omega": hx(w), "result": hxl(res)})
        prod, a, s, tv = col(), col(), col(), col()
        res = po.lookup_h(f, values, prod, a, s, tv, l0, l_last, l_active, beta, gamma, y, rot_scale)
        out.append({"op": "lookup", "field": fname, "ext_k": ext_k, "rot_scale": rot_scale, "product": hxl(prod), "permuted_input": hxl(a), "permuted_table": hxl(s),
                    "table_value": hxl(tv), "l0": hxl(l0), "l_last": hxl(l_last), "l_active": hxl(l_active), "values": hxl(values), "beta": hx(beta), "gamma": hx(gamma),
                    "y": hx(y), "result": hxl(res)})
        table = [3, 1, 4, 1, 5, 9, 2, 6] + [3] * 8
        inputs = [1, 1, 9, 3, 3, 3, 2, 6, 5, 5, 4, 1, 9, 9, 2, 3]
        pi, pt = po.permute_expression_pair(f, inputs, table, 14)
        out.append({"op": "permute", "field": fname, "usable": 14, "input": hxl(inputs), "table": hxl(table), "permuted_input": hxl(pi), "permuted_table": hxl(pt)})
    return out

if __name__ == "__main__":
    po.self_check()
    for name, fn in (("ntt", ntt_vectors), ("domain", domain_vectors), ("msm", msm_vectors), ("bases", bases_vectors), ("poly", poly_vectors), ("evalh", evalh_vectors)):
        with open(os.path.join(HERE, name + ".json"), "w") as fh:
            json.dump(fn(), fh, separators=(",", ":"))
        print("wrote", name)
    with open(os.path.join(HERE, "poseidon_kat.json"), "w") as fh:
        # the reference's own known-answer vectors (src/poseidon/permutation.rs:154-158,190-196): data, not code
        json.dump([{"field": "bn254_fr", "t": t, "r_f": rf, "r_p": rp, "input": list(range(t)), "expected": [str(x) for x in e]} for t, rf, rp, e in po.POSEIDON_KATS], fh)
    print("wrote poseidon_kat")
