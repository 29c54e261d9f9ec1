"""CPU restatement of keygen + create_proof (KZG / GWC) for the parity tests.  TEST INFRASTRUCTURE ONLY: importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product never imports it.

Follows [UPSTREAM halo2_proofs @ v2023_04_20: plonk/keygen.rs, plonk/prover.rs, plonk/lookup/prover.rs,
plonk/permutation/{keygen,prover}.rs, plonk/vanishing/prover.rs, plonk/evaluation.rs, poly/domain.rs,
poly/kzg/multiopen/gwc/prover.rs, transcript.rs] -- the code behind the reference's calls at
benches/delay_enc.rs:86,103 (keygen) and :123-131 (create_proof) -- written from the published algorithm (the crate is
not in the container): **parity unpinned** against upstream itself; what pins it is oracle/verifier.py accepting the
proofs (the reference's own end-to-end check, `assert!(accept)`, benches/delay_enc.rs:147-165).

The heavy steps go through oracle.c (best_multiexp, best_fft, evaluate_h's row loops, permute_expression_pair, batch
inversion) with upstream's serial / chunk-per-thread structure; everything else is Python integers.
The circuit arrives as plain data: `desc` = (num_advice, num_fixed, num_instance, gates, lookups, permutation_columns,
advice_queries, fixed_queries, instance_queries, minimum_degree) with expressions as nested tuples
("const", c) ("fixed" | "advice" | "instance", index, rotation) ("neg", e) ("sum", a, b) ("product", a, b) ("scaled", e, c).
"""
from __future__ import annotations

import hashlib
import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np

import coracle as co
import pyoracle as po


# ---- small helpers -----------------------------------------------------------------------------------------
def arr_from_ints(vals: Sequence[int]) -> np.ndarray:
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()


def ints_from_arr(arr) -> List[int]:
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


class Fld:
    """A scalar field with its Montgomery codec on 4 x u64 limbs."""

    def __init__(self, f: po.Field):
        self.f, self.p, self.id = f, f.p, po.FIELD_IDS[f.name]
        self.R, self.Rinv = f.R, f.R_inv

    def m(self, x: int) -> np.ndarray:
        return arr_from_ints([x % self.p * self.R % self.p])[0]

    def many(self, xs: Sequence[int]) -> np.ndarray:
        return arr_from_ints([x % self.p * self.R % self.p for x in xs])

    def un(self, limbs) -> int:
        return ints_from_arr(np.asarray(limbs).reshape(1, 4))[0] * self.Rinv % self.p

    def un_many(self, arr) -> List[int]:
        return [v * self.Rinv % self.p for v in ints_from_arr(arr)]


def expr_degree(e) -> int:
    k = e[0]
    if k == "const": return 0
    if k in ("fixed", "advice", "instance"): return 1
    if k in ("neg", "scaled"): return expr_degree(e[1])
    if k == "sum": return max(expr_degree(e[1]), expr_degree(e[2]))
    return expr_degree(e[1]) + expr_degree(e[2])


class Shape:
    """ConstraintSystem::{degree, blinding_factors} and EvaluationDomain::new from the plain description."""

    def __init__(self, desc, k: int, field: po.Field):
        (self.num_advice, self.num_fixed, self.num_instance, self.gates, self.lookups, self.perm_columns, self.advice_queries, self.fixed_queries,
         self.instance_queries, self.minimum_degree) = desc
        cnt = [0] * self.num_advice
        for c, _ in self.advice_queries:
            cnt[c] += 1
        self.blinding_factors = max(3, max(cnt or [1])) + 2
        d = 3                                                  # permutation::Argument::required_degree() is 3 whether or not a column is enabled
        for ins, tabs in self.lookups:
            d = max(d, max(4, 2 + max([1] + [expr_degree(e) for e in ins]) + max([1] + [expr_degree(e) for e in tabs])))
        for g in self.gates:
            d = max(d, expr_degree(g))
        self.degree = max(d, self.minimum_degree)
        self.chunk_len = self.degree - 2
        self.k, self.n = k, 1 << k
        self.dom = po.Domain(field, k, self.degree)
        self.ext_k, self.ext_n = self.dom.extended_k, 1 << self.dom.extended_k
        self.usable = self.n - (self.blinding_factors + 1)
        self.num_sets = (len(self.perm_columns) + self.chunk_len - 1) // self.chunk_len if self.perm_columns else 0


# ---- expressions as GraphEvaluator programs, compiled naively (one calculation per node, no simplification) ----
class Prog:
    def __init__(self, p: int):
        self.p, self.constants, self.rotations, self.calcs, self.n_int = p, [0, 1, 2], [], [], 0

    def const(self, c: int):
        c %= self.p
        if c not in self.constants:
            self.constants.append(c)
        return (po.SRC_CONSTANT, self.constants.index(c), 0)

    def rot(self, r: int) -> int:
        if r not in self.rotations:
            self.rotations.append(r)
        return self.rotations.index(r)

    def calc(self, op, a, b=(po.SRC_CONSTANT, 0, 0), parts=()):
        self.calcs.append((op, a, b, tuple(parts), self.n_int))
        self.n_int += 1
        return (po.SRC_INTERMEDIATE, self.n_int - 1, 0)

    def expr(self, e):
        k = e[0]
        if k == "const": return self.const(e[1])
        if k == "fixed": return (po.SRC_FIXED, e[1], self.rot(e[2]))
        if k == "advice": return (po.SRC_ADVICE, e[1], self.rot(e[2]))
        if k == "instance": return (po.SRC_INSTANCE, e[1], self.rot(e[2]))
        if k == "neg": return self.calc(po.CALC_NEGATE, self.expr(e[1]))
        if k == "sum": return self.calc(po.CALC_ADD, self.expr(e[1]), self.expr(e[2]))
        if k == "product": return self.calc(po.CALC_MUL, self.expr(e[1]), self.expr(e[2]))
        if k == "scaled": return self.calc(po.CALC_MUL, self.expr(e[1]), self.const(e[2]))
        raise ValueError(k)

    def run(self, F: Fld, fixed, advice, instance, challenges, beta, gamma, theta, y, log_rows, rot_scale, previous, threads):
        mm = lambda v: None if v is None else F.m(v)
        ch = F.many(challenges) if challenges else None
        return co.graph_evaluate(F.id, F.many(self.constants), self.rotations, self.calcs, self.n_int, fixed, advice, instance, ch, mm(beta), mm(gamma), mm(theta),
                                 mm(y), log_rows, rot_scale, previous, threads)


# ---- transcript (Blake2bWrite<_, _, Challenge255>) ------------------------------------------------------------------
class Transcript:
    def __init__(self, curve: po.Curve):
        self.curve, self.h, self.proof = curve, hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript"), bytearray()

    def challenge(self) -> int:
        self.h.update(b"\x00")
        return int.from_bytes(self.h.copy().digest(), "little") % self.curve.scalar.p

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + int(s).to_bytes(32, "little"))

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += int(s).to_bytes(32, "little")

    def write_point(self, P):
        assert P is not None, "cannot write points at infinity to the transcript"
        self.h.update(b"\x01" + P[0].to_bytes(32, "little") + P[1].to_bytes(32, "little"))
        b = bytearray(P[0].to_bytes(32, "little"))
        b[31] |= (P[1] & 1) << 7
        self.proof += b


# ---- SRS ------------------------------------------------------------------------------------------------------
def setup_srs(curve: po.Curve, k: int, s: int, threads: int = 1) -> Dict[str, np.ndarray]:
    """ParamsKZG::setup with the toxic waste `s` given: g[i] = [s^i]G, g_lagrange[i] = [L_i(s)]G, L_i(s) = omega^i (s^n - 1) / (n (s - omega^i))
    [UPSTREAM poly/kzg/commitment.rs setup]."""
    F, cid = Fld(curve.scalar), po.CURVE_IDS[curve.name]
    p, n = F.p, 1 << k
    omega = curve.scalar.omega(k)
    powers = co.powers(F.id, F.m(s), F.m(1), n)
    g = co.fixed_base_mul(cid, powers, threads)
    w = co.powers(F.id, F.m(omega), F.m(1), n)
    den = co.field_op(F.id, "sub", np.tile(F.m(s), (n, 1)), w)                      # s - omega^i
    den = co.batch_invert(F.id, den)
    c = (pow(s, n, p) - 1) * pow(n, -1, p) % p
    lag = co.field_op(F.id, "mul", co.field_op(F.id, "mul", den, w), np.tile(F.m(c), (n, 1)))
    gl = co.fixed_base_mul(cid, lag, threads)
    return {"k": k, "g": g, "g_lagrange": gl, "s": s}


# ---- keygen ---------------------------------------------------------------------------------------------------
def commit(curve: po.Curve, bases: np.ndarray, scalars: np.ndarray, threads: int):
    """params.commit / commit_lagrange -> affine (x, y) canonical | None."""
    cid = po.CURVE_IDS[curve.name]
    n = scalars.shape[0]
    aff = co.to_affine(cid, co.best_multiexp(cid, scalars, bases[:n], threads))
    v = ints_from_arr(aff.reshape(2, 4))
    if v[0] == 0 and v[1] == 0:
        return None
    ri = curve.base.R_inv
    return (v[0] * ri % curve.base.p, v[1] * ri % curve.base.p)


def keygen(curve: po.Curve, srs, desc, k: int, fixed_canonical: np.ndarray, mapping: np.ndarray, threads: int = 1) -> dict:
    """keygen_vk + keygen_pk.  fixed_canonical: (num_fixed, n, 4) canonical limbs; mapping: flat cell -> flat cell
    (column * n + row) of the permutation's cycles."""
    F = Fld(curve.scalar)
    sh = Shape(desc, k, curve.scalar)
    d, n, p = sh.dom, sh.n, F.p
    mm = F.m
    l2c = lambda a: co.lagrange_to_coeff(F.id, a, k, mm(d.omega_inv), mm(d.ifft_divisor), threads)
    c2e = lambda a: co.coeff_to_extended(F.id, a, k, sh.ext_k, mm(d.ext_omega), mm(d.g_coset), threads)
    fixed_values = [co.field_op(F.id, "to_mont", np.ascontiguousarray(fixed_canonical[i])) for i in range(sh.num_fixed)]
    fixed_polys = [l2c(v) for v in fixed_values]
    fixed_cosets = [c2e(c) for c in fixed_polys]
    fixed_commitments = [commit(curve, srs["g_lagrange"], v, threads) for v in fixed_values]
    # permutation: sigma_j(omega^i) = delta^(col') omega^(row') for the cell (col', row') that (j, i) maps to
    npc = len(sh.perm_columns)
    delta = pow(curve.scalar.gen, 1 << curve.scalar.S, p)
    w = co.powers(F.id, mm(d.omega), mm(1), n)
    ident = np.concatenate([co.field_op(F.id, "mul", w, np.tile(mm(pow(delta, j, p)), (n, 1))) for j in range(npc)]) if npc else np.zeros((0, 4), dtype=np.uint64)
    perm_values = [ident[np.asarray(mapping[j * n:(j + 1) * n])] for j in range(npc)]
    perm_polys = [l2c(v) for v in perm_values]
    perm_cosets = [c2e(c) for c in perm_polys]
    perm_commitments = [commit(curve, srs["g_lagrange"], v, threads) for v in perm_values]
    one = mm(1)
    u = sh.usable
    l0 = np.zeros((n, 4), dtype=np.uint64); l0[0] = one
    l_last = np.zeros((n, 4), dtype=np.uint64); l_last[u] = one
    l_blind = np.zeros((n, 4), dtype=np.uint64); l_blind[u + 1:] = one
    l0e, l_last_e, l_blind_e = c2e(l2c(l0)), c2e(l2c(l_last)), c2e(l2c(l_blind))
    ones = np.tile(one, (sh.ext_n, 1))
    l_active = co.field_op(F.id, "sub", ones, co.field_op(F.id, "add", l_last_e, l_blind_e))
    return dict(shape=sh, fixed_values=fixed_values, fixed_polys=fixed_polys, fixed_cosets=fixed_cosets, fixed_commitments=fixed_commitments,
                perm_values=perm_values, perm_polys=perm_polys, perm_cosets=perm_cosets, perm_commitments=perm_commitments, l0=l0e, l_last=l_last_e,
                l_active=l_active, delta=delta, desc=desc)


def vk_bytes(curve: po.Curve, key: dict, selectors: Sequence[np.ndarray] = ()) -> bytes:
    """VerifyingKey::write, SerdeFormat::RawBytes."""
    B = curve.base
    raw = lambda P: b"".join((c * B.R % B.p).to_bytes(32, "little") for c in (P if P is not None else (0, 0)))
    out = struct.pack(">I", key["shape"].k) + struct.pack(">I", len(key["fixed_commitments"]))
    out += b"".join(raw(P) for P in key["fixed_commitments"]) + b"".join(raw(P) for P in key["perm_commitments"])
    for sel in selectors:
        out += np.packbits(np.asarray(sel, dtype=bool), bitorder="little").tobytes()
    return out


def transcript_repr(curve: po.Curve, key: dict, selectors=()) -> int:
    body = vk_bytes(curve, key, selectors) + repr(key["desc"]).encode()
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(struct.pack("<Q", len(body)) + body)
    return int.from_bytes(h.digest(), "little") % curve.scalar.p


class ScalarStream:
    """The checker's own reproducible source of "random" scalars (tests and bench only): numpy's PCG64 seeded with `seed`, four 64-bit outputs per
    scalar, least-significant word first, the top word masked to 61 bits -- a raw 253-bit value, taken as a Montgomery representation (below all
    four moduli).  create_proof consumes it in upstream's program order (one `Scheme::Scalar::random(&mut rng)` at a time, [UPSTREAM]
    plonk/prover.rs); the product's seeded generators (Python and C++) have to produce this same stream for the proofs to be comparable byte for byte."""

    def __init__(self, seed: int):
        self.gen = np.random.Generator(np.random.PCG64(seed))

    def scalars(self, count: int) -> np.ndarray:
        a = self.gen.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 61) - 1)
        return a


# ---- create_proof ---------------------------------------------------------------------------------------------
def create_proof(curve: po.Curve, srs, key: dict, advice_mont: np.ndarray, instances: Sequence[Sequence[int]], rng, vk_repr: int, threads: int = 1):
    """-> (proof bytes, trace) -- trace holds every commitment, challenge and evaluation in order, and h's coefficients.
    advice_mont: (num_advice, n, 4) Montgomery; rng.scalars(count) -> (count, 4) Montgomery (consumed in upstream's order)."""
    F = Fld(curve.scalar)
    sh: Shape = key["shape"]
    d, n, p, k, u, bf = sh.dom, sh.n, F.p, sh.k, sh.usable, sh.blinding_factors
    mm = F.m
    T = Transcript(curve)
    trace = {"commitments": [], "challenges": {}, "evals": []}
    l2c = lambda a: co.lagrange_to_coeff(F.id, a, k, mm(d.omega_inv), mm(d.ifft_divisor), threads)
    c2e = lambda a: co.coeff_to_extended(F.id, a, k, sh.ext_k, mm(d.ext_omega), mm(d.g_coset), threads)
    mul = lambda a, b: co.field_op(F.id, "mul", a, b)
    add = lambda a, b: co.field_op(F.id, "add", a, b)
    bc = lambda x: np.tile(mm(x), (n, 1))

    def write_commit(bases, scalars):
        P = commit(curve, bases, scalars, threads)
        T.write_point(P)
        trace["commitments"].append(P)

    T.common_scalar(vk_repr)
    # instances
    inst_values = []
    for vals in instances:
        for v in vals:
            T.common_scalar(v)
        col = np.zeros((n, 4), dtype=np.uint64)
        if len(vals):
            col[:len(vals)] = F.many(vals)
        inst_values.append(col)
    inst_polys = [l2c(v) for v in inst_values]
    # advice
    advice = [np.array(advice_mont[i], dtype=np.uint64).reshape(n, 4) for i in range(sh.num_advice)]
    for a in advice:
        a[u:] = rng.scalars(n - u)
    rng.scalars(sh.num_advice)
    for a in advice:
        write_commit(srs["g_lagrange"], a)
    theta = T.challenge()
    # lookups: compress, permute
    fixed_v = key["fixed_values"]
    lookups = []
    for ins, tabs in sh.lookups:
        def compress(exprs):
            acc = np.zeros((n, 4), dtype=np.uint64)
            for e in exprs:
                pr = Prog(p)
                pr.calc(po.CALC_STORE, pr.expr(e))
                val = pr.run(F, fixed_v, advice, inst_values, None, None, None, None, None, k, 1, None, threads)
                acc = add(mul(acc, bc(theta)), val)
            return acc
        ci, ct = compress(ins), compress(tabs)
        res = co.permute_expression_pair(F.id, ci, ct, u)
        assert res is not None, "lookup input not in table (ConstraintSystemFailure)"
        pi, pt = (np.concatenate([x, np.zeros((n - u, 4), dtype=np.uint64)]) for x in res)
        pi[u:] = rng.scalars(n - u)
        pt[u:] = rng.scalars(n - u)
        rng.scalars(2)
        write_commit(srs["g_lagrange"], pi)
        write_commit(srs["g_lagrange"], pt)
        lookups.append(dict(ci=ci, ct=ct, pi=pi, pt=pt))
    beta, gamma = T.challenge(), T.challenge()
    # permutation argument
    colvals = {"advice": advice, "fixed": fixed_v, "instance": inst_values}
    w = co.powers(F.id, mm(d.omega), mm(1), n)
    perm_z, last_z, dcur = [], 1, 1
    for s in range(sh.num_sets):
        cols_s = sh.perm_columns[s * sh.chunk_len:(s + 1) * sh.chunk_len]
        den = np.tile(mm(1), (n, 1))
        for j, (ck, cidx) in enumerate(cols_s, start=s * sh.chunk_len):
            den = mul(den, add(add(mul(bc(beta), key["perm_values"][j]), bc(gamma)), colvals[ck][cidx]))
        den = co.batch_invert(F.id, den)
        modified = den
        for ck, cidx in cols_s:
            modified = mul(modified, add(add(mul(w, bc(dcur * beta % p)), bc(gamma)), colvals[ck][cidx]))
            dcur = dcur * key["delta"] % p
        z = co.field_op(F.id, "mul", co.grand_product(F.id, modified, np.tile(mm(1), (n, 1))), bc(last_z))
        z[n - bf:] = rng.scalars(bf)
        rng.scalars(1)
        last_z = F.un(z[u])
        perm_z.append(z)
        write_commit(srs["g_lagrange"], z)
    # lookup products
    for lk in lookups:
        den = mul(add(lk["pi"], bc(beta)), add(lk["pt"], bc(gamma)))
        num = mul(add(lk["ci"], bc(beta)), add(lk["ct"], bc(gamma)))
        z = co.grand_product(F.id, num, den)
        z[n - bf:] = rng.scalars(bf)
        rng.scalars(1)
        lk["z"] = z
        write_commit(srs["g_lagrange"], z)
    # vanishing: random polynomial
    random_poly = rng.scalars(n)
    rng.scalars(1)
    write_commit(srs["g"], random_poly)
    y = T.challenge()
    # coefficient forms, cosets
    advice_polys = [l2c(a) for a in advice]
    perm_z_polys = [l2c(z) for z in perm_z]
    for lk in lookups:
        lk["pi_poly"], lk["pt_poly"], lk["z_poly"] = l2c(lk["pi"]), l2c(lk["pt"]), l2c(lk["z"])
    advice_c, inst_c = [c2e(a) for a in advice_polys], [c2e(a) for a in inst_polys]
    fixed_c = key["fixed_cosets"]
    rot_scale = sh.ext_n // n
    # evaluate_h
    pr = Prog(p)
    parts = [pr.expr(g) for g in sh.gates]
    pr.calc(po.CALC_HORNER, (po.SRC_PREVIOUS, 0, 0), (po.SRC_Y, 0, 0), parts)
    h = pr.run(F, fixed_c, advice_c, inst_c, None, None, None, None, y, sh.ext_k, rot_scale, None, threads)
    if sh.num_sets:
        cmap = {"advice": advice_c, "fixed": fixed_c, "instance": inst_c}
        pcols = [cmap[ck][ci] for ck, ci in sh.perm_columns]
        h = co.permutation_h(F.id, h, [c2e(zp) for zp in perm_z_polys], pcols, key["perm_cosets"], sh.chunk_len, -(bf + 1), key["l0"], key["l_last"], key["l_active"],
                             mm(beta), mm(gamma), mm(y), mm(key["delta"]), mm(beta * d.g_coset % p), mm(d.ext_omega), sh.ext_k, rot_scale, threads)
    for (ins, tabs), lk in zip(sh.lookups, lookups):
        pr = Prog(p)
        ci = pr.calc(po.CALC_HORNER, (po.SRC_CONSTANT, 0, 0), (po.SRC_THETA, 0, 0), [pr.expr(e) for e in ins])
        ct = pr.calc(po.CALC_HORNER, (po.SRC_CONSTANT, 0, 0), (po.SRC_THETA, 0, 0), [pr.expr(e) for e in tabs])
        pr.calc(po.CALC_MUL, pr.calc(po.CALC_ADD, ci, (po.SRC_BETA, 0, 0)), pr.calc(po.CALC_ADD, ct, (po.SRC_GAMMA, 0, 0)))
        tv = pr.run(F, fixed_c, advice_c, inst_c, None, beta, gamma, theta, None, sh.ext_k, rot_scale, None, threads)
        h = co.lookup_h(F.id, h, c2e(lk["z_poly"]), c2e(lk["pi_poly"]), c2e(lk["pt_poly"]), tv, key["l0"], key["l_last"], key["l_active"], mm(beta), mm(gamma), mm(y),
                        sh.ext_k, rot_scale, threads)
    # divide by t(X), back to coefficients, split, commit
    orig, step = pow(d.g_coset, n, p), pow(d.ext_omega, n, p)
    t_inv = F.many([pow((orig * pow(step, i, p) - 1) % p, -1, p) for i in range(rot_scale)])
    h = co.scale_periodic(F.id, h, t_inv)
    hc = co.extended_to_coeff(F.id, h, sh.ext_k, mm(d.ext_omega_inv), mm(d.ext_ifft_divisor), mm(d.g_coset), threads)
    pieces_n = sh.degree - 1
    pieces = [np.ascontiguousarray(hc[i * n:(i + 1) * n]) for i in range(pieces_n)]
    rng.scalars(pieces_n)
    for pc in pieces:
        write_commit(srs["g"], pc)
    x = T.challenge()
    xn = pow(x, n, p)
    rotate = lambda r: x * pow(d.omega if r >= 0 else d.omega_inv, abs(r), p) % p
    ev = lambda poly, pt: F.un(co.eval_polynomial(F.id, poly, mm(pt), threads))

    def write_eval(poly, pt):
        e = ev(poly, pt)
        T.write_scalar(e)
        trace["evals"].append(e)
        return e

    Q = []          # (point, poly, eval)
    adv_evals = [write_eval(advice_polys[c], rotate(r)) for c, r in sh.advice_queries]
    fix_evals = [write_eval(key["fixed_polys"][c], rotate(r)) for c, r in sh.fixed_queries]
    hfold = co.lincomb(F.id, pieces, F.many([pow(xn, i, p) for i in range(pieces_n)]))
    random_eval = write_eval(random_poly, x)
    sigma_evals = [write_eval(sp, x) for sp in key["perm_polys"]]
    x_next, x_inv, x_last = rotate(1), rotate(-1), rotate(-(bf + 1))
    pz_evals = []
    for s, zp in enumerate(perm_z_polys):
        e0, e1 = write_eval(zp, x), write_eval(zp, x_next)
        el = write_eval(zp, x_last) if s != len(perm_z_polys) - 1 else None
        pz_evals.append((e0, e1, el))
    lk_evals = []
    for lk in lookups:
        lk_evals.append((write_eval(lk["z_poly"], x), write_eval(lk["z_poly"], x_next), write_eval(lk["pi_poly"], x), write_eval(lk["pi_poly"], x_inv),
                         write_eval(lk["pt_poly"], x)))
    # queries
    for (c, r), e in zip(sh.advice_queries, adv_evals):
        Q.append((rotate(r), advice_polys[c], e))
    for zp, (e0, e1, _) in zip(perm_z_polys, pz_evals):
        Q += [(x, zp, e0), (x_next, zp, e1)]
    for zp, (_, _, el) in reversed(list(zip(perm_z_polys, pz_evals))[:-1]):      # [UPSTREAM permutation::prover::Evaluated::open: sets.iter().rev().skip(1)]
        Q.append((x_last, zp, el))
    for lk, (z0, z1, a0, am1, t0) in zip(lookups, lk_evals):
        Q += [(x, lk["z_poly"], z0), (x, lk["pi_poly"], a0), (x, lk["pt_poly"], t0), (x_inv, lk["pi_poly"], am1), (x_next, lk["z_poly"], z1)]
    for (c, r), e in zip(sh.fixed_queries, fix_evals):
        Q.append((rotate(r), key["fixed_polys"][c], e))
    for sp, e in zip(key["perm_polys"], sigma_evals):
        Q.append((x, sp, e))
    Q.append((x, hfold, ev(hfold, x)))
    Q.append((x, random_poly, random_eval))
    # GWC
    v = T.challenge()
    points: List[int] = []
    groups: Dict[int, list] = {}
    for pt, poly, e in Q:
        if pt not in groups:
            groups[pt] = []
            points.append(pt)
        groups[pt].append((poly, e))
    for pt in points:
        coefs = [pow(v, i, p) for i in range(len(groups[pt]))]
        eval_batch = sum(c * e for c, (_, e) in zip(coefs, groups[pt])) % p
        poly_batch = co.lincomb(F.id, [q for q, _ in groups[pt]], F.many(coefs), mm(eval_batch))
        witness = co.kate_division(F.id, poly_batch, mm(pt))
        write_commit(srs["g"], witness)
    trace["challenges"] = dict(theta=theta, beta=beta, gamma=gamma, y=y, x=x, v=v)
    trace["perm_last_evals"] = [el for _, _, el in pz_evals[:-1]]                        # as written to the transcript: set 0, 1, ...
    trace["x_last_group"] = [e for _, e in groups.get(x_last, [])] if sh.num_sets > 1 else []      # as batched with 1, v, v^2, ...
    trace["h_coeffs"] = hc
    return bytes(T.proof), trace
