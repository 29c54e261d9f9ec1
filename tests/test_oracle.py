"""CPU suite: pins the oracle (the checker) itself.  pyoracle (Python integers) is pinned by the
reference's Poseidon KATs and by definition-level identities; the C restatement is pinned by
pyoracle and by the committed golden vectors."""
import numpy as np
import pytest

from conftest import dec_point, enc_points, golden


def test_pyoracle_self_check(po):
    po.self_check()  # primes, two-adicity, [order]G = O, zeta^3 = 1, both Poseidon KATs


def test_poseidon_kats_reference_vectors(po):
    # src/poseidon/permutation.rs:154-158 and :190-196 (data committed in tests/golden/poseidon_kat.json)
    for kat in golden("poseidon_kat"):
        f = po.FIELDS[kat["field"]]
        out = po.poseidon_permute_ref(f, kat["input"], kat["r_f"], kat["r_p"])
        assert out == [int(x) for x in kat["expected"]]


def test_poseidon_kats_through_c_oracle_field_ops(po, co):
    """Same permutation, but every mul/add goes through oracle.c's Montgomery arithmetic:
    pins the C restatement's bn256::Fr mul/add (and to/from Montgomery)."""
    f = po.BN254_FR
    fid = po.FIELD_IDS[f.name]

    def enc(a):
        return co.limbs(f.to_mont(a)).reshape(1, 4)

    def dec(l):
        return f.from_mont(po.from_limbs64(l.reshape(4)))

    mul = lambda a, b: dec(co.field_op(fid, "mul", enc(a), enc(b)))
    add = lambda a, b: dec(co.field_op(fid, "add", enc(a), enc(b)))
    kat = golden("poseidon_kat")[0]
    out = po.poseidon_permute_ref(f, kat["input"], kat["r_f"], kat["r_p"], mul=mul, add=add)
    assert out == [int(x) for x in kat["expected"]]


@pytest.mark.parametrize("fname", ["bn254_fr", "bn254_fq", "pasta_fp", "pasta_fq"])
def test_c_field_ops_vs_python(po, co, fname):
    f = po.FIELDS[fname]
    fid = po.FIELD_IDS[fname]
    p = f.p
    rng = po.Xoshiro(1234 + fid)
    vals = [0, 1, 2, p - 1, p - 2, (1 << 255) % p, f.R % p, (p - 1) // 2] + [rng.below(p) for _ in range(200)]
    vb = list(reversed(vals))
    a = np.stack([co.limbs(f.to_mont(x)) for x in vals])
    b = np.stack([co.limbs(f.to_mont(x)) for x in vb])
    dec = lambda arr: [f.from_mont(po.from_limbs64(r)) for r in arr]
    assert dec(co.field_op(fid, "add", a, b)) == [(x + y) % p for x, y in zip(vals, vb)]
    assert dec(co.field_op(fid, "sub", a, b)) == [(x - y) % p for x, y in zip(vals, vb)]
    assert dec(co.field_op(fid, "mul", a, b)) == [(x * y) % p for x, y in zip(vals, vb)]
    nz = [x for x in vals if x]
    an = np.stack([co.limbs(f.to_mont(x)) for x in nz])
    assert dec(co.field_op(fid, "inv", an)) == [pow(x, -1, p) for x in nz]
    canon = np.stack([co.limbs(x) for x in vals])
    assert np.array_equal(co.field_op(fid, "to_mont", canon), a)
    assert np.array_equal(co.field_op(fid, "from_mont", a), canon)


def test_c_field_constants(po, co):
    import ctypes as C
    for name, f in po.FIELDS.items():
        p = np.zeros(4, np.uint64); r = np.zeros(4, np.uint64); r2 = np.zeros(4, np.uint64); inv = C.c_uint64()
        assert co.lib().orc_field_info(po.FIELD_IDS[name], p.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), r2.ctypes.data_as(C.c_void_p), C.byref(inv)) == 0
        assert po.from_limbs64(p) == f.p and po.from_limbs64(r) == f.R and po.from_limbs64(r2) == f.R2 and inv.value == f.inv64


def test_golden_ntt_vs_c_oracle(po, co):
    for v in golden("ntt"):
        f = po.FIELDS[v["field"]]
        fid = po.FIELD_IDS[v["field"]]
        a = np.stack([co.limbs(f.to_mont(int(x, 16))) for x in v["input"]])
        w = co.limbs(f.to_mont(int(v["omega"], 16)))
        for threads in (1, 4):
            out = co.best_fft(fid, a, w, v["log_n"], threads)
            assert [f.from_mont(po.from_limbs64(r)) for r in out] == [int(x, 16) for x in v["output"]]


def test_golden_domain_vs_c_oracle(po, co):
    for v in golden("domain"):
        f = po.FIELDS[v["field"]]
        fid = po.FIELD_IDS[v["field"]]
        d = po.Domain(f, v["k"], v["j"])
        assert d.extended_k == v["extended_k"]
        enc = lambda xs: np.stack([co.limbs(f.to_mont(int(x, 16))) for x in xs])
        dec = lambda arr: [f.from_mont(po.from_limbs64(r)) for r in arr]
        m = lambda x: co.limbs(f.to_mont(x))
        coeffs = co.lagrange_to_coeff(fid, enc(v["lagrange"]), d.k, m(d.omega_inv), m(d.ifft_divisor))
        assert dec(coeffs) == [int(x, 16) for x in v["coeffs"]]
        ext = co.coeff_to_extended(fid, coeffs, d.k, d.extended_k, m(d.ext_omega), m(int(v["zeta"], 16)))
        assert dec(ext) == [int(x, 16) for x in v["extended"]]
        back = co.extended_to_coeff(fid, ext, d.extended_k, m(d.ext_omega_inv), m(d.ext_ifft_divisor), m(int(v["zeta"], 16)))
        assert dec(back)[: d.n * d.quotient_poly_degree] == [int(x, 16) for x in v["back"]]


def _golden_points(po, fb_enc, v, bases_json):
    if "points" in v:
        pts = [None if P is None else (int(P[0], 16), int(P[1], 16)) for P in v["points"]]
    else:
        pts = [(int(P[0], 16), int(P[1], 16)) for P in bases_json[v["curve"]][: v["n"]]]
    return pts


def test_golden_msm_vs_c_oracle(po, co, pkg):
    bases_json = golden("bases")
    for v in golden("msm"):
        c = po.CURVES[v["curve"]]
        cid = po.CURVE_IDS[v["curve"]]
        spec = pkg.fields.CURVES[v["curve"]]
        pts = _golden_points(po, None, v, bases_json)
        bases = enc_points(spec.base, pts)
        scalars = np.stack([co.limbs(c.scalar.to_mont(int(x, 16))) for x in v["scalars"]])
        want = None if v["result"] is None else (int(v["result"][0], 16), int(v["result"][1], 16))
        for threads in (1, 3):
            got = dec_point(spec.base, co.to_affine(cid, co.best_multiexp(cid, scalars, bases, threads)))
            assert got == want, (v["curve"], v["n"], v["dist"], threads)


def test_c_generators_match_python(po, co):
    for cname, cid in po.CURVE_IDS.items():
        c = po.CURVES[cname]
        b = co.synth_bases(cid, 20)
        pb = po.synth_bases(c, 20)
        for i in range(20):
            assert po.from_limbs64(b[i, :4]) == c.base.to_mont(pb[i][0]) and po.from_limbs64(b[i, 4:]) == c.base.to_mont(pb[i][1])
    for dist, gen in (("uniform", po.scalars_uniform), ("witness", po.scalars_witness_like), ("lookup", po.scalars_lookup_like)):
        f = po.PASTA_FQ
        s = co.fill_scalars(po.FIELD_IDS[f.name], dist, 500)
        assert [f.from_mont(po.from_limbs64(x)) for x in s] == gen(f, 500, po.Xoshiro())


def test_c_msm_random_vs_python_pippenger(po, co):
    c = po.PALLAS
    cid = po.CURVE_IDS["pallas"]
    n = 300
    b = co.synth_bases(cid, n)
    pb = po.synth_bases(c, n)
    for dist in ("uniform", "witness", "lookup"):
        s = co.fill_scalars(po.FIELD_IDS[c.scalar.name], dist, n, 99)
        ps = [c.scalar.from_mont(po.from_limbs64(x)) for x in s]
        want = po.msm_pippenger(c, ps, pb)
        for threads in (1, 7):
            a = co.to_affine(cid, co.best_multiexp(cid, s, b, threads))
            got = None if not a.any() else (c.base.from_mont(po.from_limbs64(a[:4])), c.base.from_mont(po.from_limbs64(a[4:])))
            assert got == want


def test_c_fft_roundtrip_and_threads(po, co):
    f = po.BN254_FR
    fid = po.FIELD_IDS[f.name]
    k = 12
    a = co.fill_scalars(fid, "uniform", 1 << k, 5)
    w = co.limbs(f.to_mont(f.omega(k)))
    wi = co.limbs(f.to_mont(f.inv(f.omega(k))))
    one = co.best_fft(fid, a, w, k, 1)
    for t in (2, 3, 8):
        assert np.array_equal(co.best_fft(fid, a, w, k, t), one)
    back = co.best_fft(fid, one, wi, k, 4)
    ninv = co.limbs(f.to_mont(f.inv(1 << k)))
    back = co.field_op(fid, "mul", back, np.tile(ninv, (1 << k, 1)))
    assert np.array_equal(back, a)


# ---------------------------------------------------------------- field-vector primitives
def _enc(po, f, vals):
    return np.array([[(f.to_mont(v) >> (64 * i)) & (2**64 - 1) for i in range(4)] for v in vals], dtype=np.uint64).reshape(-1, 4)


def _dec(po, f, arr):
    return [f.from_mont(po.from_limbs64(r)) for r in np.asarray(arr).reshape(-1, 4)]


def test_poly_golden_vs_both_oracles(po, co, golden_loader):
    for v in golden_loader("poly"):
        f = po.FIELDS[v["field"]]
        fid = po.FIELD_IDS[v["field"]]
        if v["op"] == "eval_polynomial":
            poly, x, want = [int(c, 16) for c in v["poly"]], int(v["point"], 16), int(v["result"], 16)
            assert po.eval_polynomial(f, poly, x) == want
            assert want == sum(c * pow(x, i, f.p) for i, c in enumerate(poly)) % f.p       # the definition
            for threads in (1, 3):
                assert _dec(po, f, co.eval_polynomial(fid, _enc(po, f, poly), _enc(po, f, [x])[0], threads)) == [want]
        elif v["op"] == "batch_invert":
            vals, want = [int(c, 16) for c in v["values"]], [int(c, 16) for c in v["result"]]
            assert po.batch_invert(f, vals) == want
            assert all((a * b) % f.p == (1 if a else 0) for a, b in zip(vals, want))
            assert _dec(po, f, co.batch_invert(fid, _enc(po, f, vals))) == want
        else:
            num, den, want = ([int(c, 16) for c in v[k]] for k in ("num", "den", "result"))
            assert po.grand_product(f, num, den) == want
            assert _dec(po, f, co.grand_product(fid, _enc(po, f, num), _enc(po, f, den))) == want


def test_c_eval_polynomial_threads_agree(po, co):
    f = po.PASTA_FP
    fid = po.FIELD_IDS[f.name]
    c = co.fill_scalars(fid, "uniform", 5000, 77)
    x = co.fill_scalars(fid, "uniform", 1, 78)[0]
    one = co.eval_polynomial(fid, c, x, 1)
    for t in (2, 7, 64):
        assert np.array_equal(co.eval_polynomial(fid, c, x, t), one)


# ---------------------------------------------------------------- quotient numerator
def test_evalh_c_oracle_vs_python(po, co):
    from conftest import random_graph
    f = po.BN254_FR
    fid = po.FIELD_IDS[f.name]
    rng = po.Xoshiro(0xE0A1)
    k, ext_k = 4, 6
    rows, rot_scale = 1 << ext_k, 1 << (ext_k - k)
    col = lambda: [rng.below(f.p) for _ in range(rows)]
    enc = lambda vals: _enc(po, f, vals)
    env = {"fixed": [col() for _ in range(3)], "advice": [col() for _ in range(4)], "instance": [col()], "challenges": [rng.below(f.p) for _ in range(2)],
           "beta": rng.below(f.p), "gamma": rng.below(f.p), "theta": rng.below(f.p), "y": rng.below(f.p)}
    previous = col()
    for n_calcs, long_lived in ((15, 0), (70, 25)):
        g = random_graph(po, f, rng, n_calcs, 3, 4, 1, 2, long_lived)
        want = po.graph_evaluate(f, g, env, rows, rot_scale, previous)
        for threads in (1, 5):
            got = co.graph_evaluate(fid, enc(g["constants"]), g["rotations"], g["calcs"], g["num_intermediates"], [enc(c) for c in env["fixed"]],
                                    [enc(c) for c in env["advice"]], [enc(c) for c in env["instance"]], enc(env["challenges"]), enc([env["beta"]])[0],
                                    enc([env["gamma"]])[0], enc([env["theta"]])[0], enc([env["y"]])[0], ext_k, rot_scale, enc(previous), threads)
            assert _dec(po, f, got) == want
    # permutation and lookup terms
    z, cols, sigma = [col() for _ in range(2)], [col() for _ in range(5)], [col() for _ in range(5)]
    l0, l_last, l_active, values = col(), col(), col(), col()
    beta, gamma, y, delta = (rng.below(f.p) for _ in range(4))
    zeta, w = po.zeta(f), f.omega(ext_k)
    want = po.permutation_h(f, values, z, cols, sigma, 3, -6, l0, l_last, l_active, beta, gamma, y, delta, zeta, w, rot_scale)
    e1 = lambda v: enc([v])[0]
    for threads in (1, 7):
        got = co.permutation_h(fid, enc(values), [enc(c) for c in z], [enc(c) for c in cols], [enc(c) for c in sigma], 3, -6, enc(l0), enc(l_last), enc(l_active),
                               e1(beta), e1(gamma), e1(y), e1(delta), e1(beta * zeta % f.p), e1(w), ext_k, rot_scale, threads)
        assert _dec(po, f, got) == want
    prod, a, s, tv = col(), col(), col(), col()
    want = po.lookup_h(f, values, prod, a, s, tv, l0, l_last, l_active, beta, gamma, y, rot_scale)
    got = co.lookup_h(fid, enc(values), enc(prod), enc(a), enc(s), enc(tv), enc(l0), enc(l_last), enc(l_active), e1(beta), e1(gamma), e1(y), ext_k, rot_scale, 3)
    assert _dec(po, f, got) == want


def _lookup_case(po, f, rng, n, table_size, spread):
    """A lookup-shaped pair: a table of `table_size` distinct values padded with repeats of its first
    entry (halo2 pads range tables with zeros), inputs drawn from the first `spread` table values."""
    base = [rng.below(f.p) if i % 3 else i for i in range(table_size)]
    table = base + [base[0]] * (n - table_size)
    inputs = [base[rng.below(spread)] for _ in range(n)]
    return inputs, table


def test_permute_expression_pair_oracles_agree(po, co):
    f = po.BN254_FR
    fid = po.FIELD_IDS[f.name]
    rng = po.Xoshiro(0x5077)
    for n, tsize, spread in ((1, 1, 1), (16, 16, 16), (200, 64, 64), (200, 64, 3), (257, 256, 100)):
        inputs, table = _lookup_case(po, f, rng, n, tsize, spread)
        want = po.permute_expression_pair(f, inputs, table, n)
        assert want is not None
        pi, pt = want
        assert sorted(pi) == sorted(inputs) and sorted(pt) == sorted(table)                  # both are permutations
        assert all(pi[r] == pt[r] or pi[r] == pi[r - 1] for r in range(n))                  # the lookup argument's row rule
        got = co.permute_expression_pair(fid, _enc(po, f, inputs), _enc(po, f, table), n)
        assert _dec(po, f, got[0]) == pi and _dec(po, f, got[1]) == pt
    inputs, table = _lookup_case(po, f, rng, 50, 20, 20)
    inputs[7] = (max(table) + 1) % f.p if (max(table) + 1) % f.p not in table else 12345678901234567
    assert po.permute_expression_pair(f, inputs, table, 50) is None
    assert co.permute_expression_pair(fid, _enc(po, f, inputs), _enc(po, f, table), 50) is None
    # usable_rows shorter than the columns: the tail is ignored
    inputs, table = _lookup_case(po, f, rng, 80, 30, 30)
    want = po.permute_expression_pair(f, inputs[:60] + [5] * 20, table[:60] + [7] * 20, 60)
    got = co.permute_expression_pair(fid, _enc(po, f, inputs[:60] + [5] * 20), _enc(po, f, table[:60] + [7] * 20), 60)
    assert (want is None) == (got is None)
    if want is not None:
        assert _dec(po, f, got[0]) == want[0] and _dec(po, f, got[1]) == want[1]


def _golden_graph(v):
    g = v["graph"]
    return {"constants": [int(c, 16) for c in g["constants"]], "rotations": g["rotations"], "num_intermediates": g["num_intermediates"],
            "calcs": [(op, tuple(a), tuple(b), tuple(tuple(q) for q in parts), t) for op, a, b, parts, t in g["calcs"]]}


def test_evalh_golden_vs_both_oracles(po, co, golden_loader):
    ints = lambda l: [int(x, 16) for x in l]
    for v in golden_loader("evalh"):
        f = po.FIELDS[v["field"]]
        fid = po.FIELD_IDS[v["field"]]
        enc = lambda vals: _enc(po, f, vals)
        e1 = lambda x: enc([x])[0]
        if v["op"] == "graph":
            g = _golden_graph(v)
            env = {k: [ints(c) for c in v["env"][k]] for k in ("fixed", "advice", "instance")}
            env.update(challenges=ints(v["env"]["challenges"]), **{k: int(v["env"][k], 16) for k in ("beta", "gamma", "theta", "y")})
            want, prev = ints(v["result"]), ints(v["previous"])
            assert po.graph_evaluate(f, g, env, 1 << v["ext_k"], v["rot_scale"], prev) == want
            got = co.graph_evaluate(fid, enc(g["constants"]), g["rotations"], g["calcs"], g["num_intermediates"], [enc(c) for c in env["fixed"]],
                                    [enc(c) for c in env["advice"]], [enc(c) for c in env["instance"]], enc(env["challenges"]), e1(env["beta"]), e1(env["gamma"]),
                                    e1(env["theta"]), e1(env["y"]), v["ext_k"], v["rot_scale"], enc(prev), 3)
            assert _dec(po, f, got) == want
        elif v["op"] == "permutation":
            a = {k: ints(v[k]) for k in ("l0", "l_last", "l_active", "values")}
            z, cols, sigma = ([ints(c) for c in v[k]] for k in ("z", "columns", "sigma"))
            sc = {k: int(v[k], 16) for k in ("beta", "gamma", "y", "delta", "zeta", "extended_omega")}
            want = ints(v["result"])
            assert po.permutation_h(f, a["values"], z, cols, sigma, v["chunk_len"], v["last_rotation"], a["l0"], a["l_last"], a["l_active"], sc["beta"], sc["gamma"],
                                    sc["y"], sc["delta"], sc["zeta"], sc["extended_omega"], v["rot_scale"]) == want
            got = co.permutation_h(fid, enc(a["values"]), [enc(c) for c in z], [enc(c) for c in cols], [enc(c) for c in sigma], v["chunk_len"], v["last_rotation"],
                                   enc(a["l0"]), enc(a["l_last"]), enc(a["l_active"]), e1(sc["beta"]), e1(sc["gamma"]), e1(sc["y"]), e1(sc["delta"]),
                                   e1(sc["beta"] * sc["zeta"] % f.p), e1(sc["extended_omega"]), v["ext_k"], v["rot_scale"], 2)
            assert _dec(po, f, got) == want
        elif v["op"] == "lookup":
            a = {k: ints(v[k]) for k in ("product", "permuted_input", "permuted_table", "table_value", "l0", "l_last", "l_active", "values")}
            sc = {k: int(v[k], 16) for k in ("beta", "gamma", "y")}
            want = ints(v["result"])
            assert po.lookup_h(f, a["values"], a["product"], a["permuted_input"], a["permuted_table"], a["table_value"], a["l0"], a["l_last"], a["l_active"],
                               sc["beta"], sc["gamma"], sc["y"], v["rot_scale"]) == want
            got = co.lookup_h(fid, enc(a["values"]), enc(a["product"]), enc(a["permuted_input"]), enc(a["permuted_table"]), enc(a["table_value"]), enc(a["l0"]),
                              enc(a["l_last"]), enc(a["l_active"]), e1(sc["beta"]), e1(sc["gamma"]), e1(sc["y"]), v["ext_k"], v["rot_scale"], 2)
            assert _dec(po, f, got) == want
        else:
            inp, tab = ints(v["input"]), ints(v["table"])
            want = (ints(v["permuted_input"]), ints(v["permuted_table"]))
            assert po.permute_expression_pair(f, inp, tab, v["usable"]) == want
            got = co.permute_expression_pair(fid, enc(inp), enc(tab), v["usable"])
            assert (_dec(po, f, got[0]), _dec(po, f, got[1])) == want


def test_checker_states_the_constraint_systems_itself(pkg):
    """oracle/shapes.py writes the two constraint systems of the reference's circuits down on its own (MainGate alone: pose_enc; MainGate + RangeChip:
    mod_pow / delay_enc); the product's plonk.maingate_cs must describe exactly that -- columns, the gate, the five lookups, equality columns, query order."""
    import shapes
    from dehalo2_amd import plonk

    for rl in (False, True):
        d = shapes.maingate_description(rl)
        assert d == plonk.maingate_cs(rl).description()
        assert (d[0], d[1], d[2], len(d[3]), len(d[4]), len(d[5])) == (5, 15 if rl else 9, 1, 1, 5 if rl else 0, 6)
