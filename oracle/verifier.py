"""verify_proof over KZG / GWC on Python integers.  TEST INFRASTRUCTURE ONLY.

The reference's only check of the hot path's results is `assert!(accept)` after
`verify_proof::<KZGCommitmentScheme<Bn256>, VerifierGWC<_>, Challenge255<_>, Blake2bRead<_, _, _>, SingleStrategy<_>>`
(benches/delay_enc.rs:147-165).  This restates that verifier [UPSTREAM halo2_proofs @ v2023_04_20: plonk/verifier.rs,
plonk/{lookup,permutation,vanishing}/verifier.rs, poly/kzg/multiopen/gwc/verifier.rs, poly/kzg/strategy.rs] from the published
protocol: read the proof through the transcript, recompute h(x) from the evaluations, and check the batched KZG openings
with one pairing product.  A proof is accepted only if every commitment, evaluation and quotient in it is consistent --
which is what ties the device's MSM / NTT results to the mathematics rather than to this repository's own oracles.

Inputs are plain data: the circuit description tuple (see plonk_oracle.py), the verifying key's commitments as canonical
affine points, g[0], g2 and s_g2.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Sequence

import pairing as pr
import pyoracle as po
from plonk_oracle import Shape


class ReadTranscript:
    def __init__(self, curve: po.Curve, proof: bytes):
        self.curve, self.h, self.data, self.pos = curve, hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript"), bytes(proof), 0

    def challenge(self) -> int:
        self.h.update(b"\x00")
        return int.from_bytes(self.h.copy().digest(), "little") % self.curve.scalar.p

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + int(s).to_bytes(32, "little"))

    def _take(self) -> bytes:
        if self.pos + 32 > len(self.data):
            raise ValueError("proof too short")
        self.pos += 32
        return self.data[self.pos - 32:self.pos]

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(), "little")
        if s >= self.curve.scalar.p:
            raise ValueError("non-canonical scalar")
        self.common_scalar(s)
        return s

    def read_point(self):
        b = self._take()
        sign = b[31] >> 7
        x = int.from_bytes(b[:31] + bytes([b[31] & 0x7F]), "little")
        p = self.curve.base.p
        if x >= p:
            raise ValueError("non-canonical x")
        if x == 0 and sign == 0:
            raise ValueError("identity in proof")
        rhs = (x * x % p * x + self.curve.b) % p
        assert p % 4 == 3
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p != rhs:
            raise ValueError("not on curve")
        if (y & 1) != sign:
            y = p - y
        self.h.update(b"\x01" + x.to_bytes(32, "little") + y.to_bytes(32, "little"))
        return (x, y)


def eval_expr(e, p, fixed, advice, instance) -> int:
    """Expression::evaluate with the queried evaluations: fixed / advice / instance are dicts (column, rotation) -> value."""
    k = e[0]
    if k == "const": return e[1] % p
    if k == "fixed": return fixed[(e[1], e[2])]
    if k == "advice": return advice[(e[1], e[2])]
    if k == "instance": return instance[(e[1], e[2])]
    if k == "neg": return -eval_expr(e[1], p, fixed, advice, instance) % p
    if k == "sum": return (eval_expr(e[1], p, fixed, advice, instance) + eval_expr(e[2], p, fixed, advice, instance)) % p
    if k == "product": return eval_expr(e[1], p, fixed, advice, instance) * eval_expr(e[2], p, fixed, advice, instance) % p
    if k == "scaled": return eval_expr(e[1], p, fixed, advice, instance) * e[2] % p
    raise ValueError(k)


def verify_proof(curve: po.Curve, desc, k: int, fixed_commitments, perm_commitments, vk_repr: int, g0, g2, s_g2, instances: Sequence[Sequence[int]], proof: bytes) -> bool:
    f = curve.scalar
    p = f.p
    sh = Shape(desc, k, f)
    d, n, bf = sh.dom, sh.n, sh.blinding_factors
    T = ReadTranscript(curve, proof)
    try:
        T.common_scalar(vk_repr)
        for vals in instances:
            for v in vals:
                T.common_scalar(v)
        advice_commitments = [T.read_point() for _ in range(sh.num_advice)]
        theta = T.challenge()
        lookups_permuted = [(T.read_point(), T.read_point()) for _ in sh.lookups]
        beta, gamma = T.challenge(), T.challenge()
        perm_z_commitments = [T.read_point() for _ in range(sh.num_sets)]
        lookup_z_commitments = [T.read_point() for _ in sh.lookups]
        random_commitment = T.read_point()
        y = T.challenge()
        h_commitments = [T.read_point() for _ in range(sh.degree - 1)]
        x = T.challenge()
        advice_evals = [T.read_scalar() for _ in sh.advice_queries]
        fixed_evals = [T.read_scalar() for _ in sh.fixed_queries]
        random_eval = T.read_scalar()
        sigma_evals = [T.read_scalar() for _ in sh.perm_columns]
        perm_evals = []
        for s in range(sh.num_sets):
            e0, e1 = T.read_scalar(), T.read_scalar()
            perm_evals.append((e0, e1, T.read_scalar() if s != sh.num_sets - 1 else None))
        lookup_evals = [tuple(T.read_scalar() for _ in range(5)) for _ in sh.lookups]     # product, product_next, permuted_input, permuted_input_inv, permuted_table
    except ValueError:
        return False
    xn = pow(x, n, p)
    rotate = lambda r: x * pow(d.omega if r >= 0 else d.omega_inv, abs(r), p) % p
    # l_i(x) for i in [-(bf+1), 0] (domain.l_i_range): l_i(x) = omega^i (x^n - 1) / (n (x - omega^i))
    def l_i(i):
        wi = pow(d.omega, i % n, p)
        return wi * (xn - 1) % p * pow(n * (x - wi) % p, -1, p) % p
    l_evals = [l_i(-i) for i in range(bf + 2)]            # l_0, l_{-1}, ..., l_{-(bf+1)}
    l_0, l_last = l_evals[0], l_evals[bf + 1]
    l_blind = sum(l_evals[1:bf + 1]) % p
    # instance evaluations: interpolated by the verifier (KZG: QUERY_INSTANCE = false)
    inst = {}
    for (c, r) in sh.instance_queries:
        vals = instances[c]
        inst[(c, r)] = sum(v * l_i(j - r) for j, v in enumerate(vals)) % p
    fx = {q: e for q, e in zip(sh.fixed_queries, fixed_evals)}
    av = {q: e for q, e in zip(sh.advice_queries, advice_evals)}
    # expressions, folded with y
    exprs: List[int] = [eval_expr(g, p, fx, av, inst) for g in sh.gates]
    # permutation::verifier::Evaluated::expressions
    if sh.num_sets:
        colval = lambda ck, ci: {"advice": av, "fixed": fx, "instance": inst}[ck][(ci, 0)]
        exprs.append(l_0 * (1 - perm_evals[0][0]) % p)
        zl = perm_evals[-1][0]
        exprs.append(l_last * (zl * zl - zl) % p)
        for s in range(1, sh.num_sets):
            exprs.append(l_0 * (perm_evals[s][0] - perm_evals[s - 1][2]) % p)
        delta = pow(f.gen, 1 << f.S, p)
        for s in range(sh.num_sets):
            cols = sh.perm_columns[s * sh.chunk_len:(s + 1) * sh.chunk_len]
            left = perm_evals[s][1]
            for j, (ck, ci) in enumerate(cols, start=s * sh.chunk_len):
                left = left * (colval(ck, ci) + beta * sigma_evals[j] + gamma) % p
            right = perm_evals[s][0]
            cur = beta * x % p * pow(delta, s * sh.chunk_len, p) % p
            for ck, ci in cols:
                right = right * (colval(ck, ci) + cur + gamma) % p
                cur = cur * delta % p
            exprs.append((left - right) * (1 - (l_last + l_blind)) % p)
    # lookup::verifier::Evaluated::expressions
    active = (1 - (l_last + l_blind)) % p
    for (ins, tabs), (z0, z1, a0, am1, s0) in zip(sh.lookups, lookup_evals):
        def compress(es):
            acc = 0
            for e in es:
                acc = (acc * theta + eval_expr(e, p, fx, av, inst)) % p
            return acc
        left = z1 * (a0 + beta) % p * (s0 + gamma) % p
        right = z0 * (compress(ins) + beta) % p * (compress(tabs) + gamma) % p
        exprs += [l_0 * (1 - z0) % p, l_last * (z0 * z0 - z0) % p, (left - right) * active % p, l_0 * (a0 - s0) % p, (a0 - s0) * (a0 - am1) % p * active % p]
    expected_h = 0
    for e in exprs:
        expected_h = (expected_h * y + e) % p
    expected_h = expected_h * pow(xn - 1, -1, p) % p
    # h commitment folded with x^n
    C = curve
    h_commitment = None
    for hc in reversed(h_commitments):
        h_commitment = po.ec_add(C, po.ec_mul(C, xn, h_commitment) if h_commitment is not None else None, hc)
    # queries (commitment, point, eval), in the prover's order
    x_next, x_inv, x_last = rotate(1), rotate(-1), rotate(-(bf + 1))
    Q = []
    for (c, r), e in zip(sh.advice_queries, advice_evals):
        Q.append((advice_commitments[c], rotate(r), e))
    for zc, (e0, e1, _) in zip(perm_z_commitments, perm_evals):
        Q += [(zc, x, e0), (zc, x_next, e1)]
    for zc, (_, _, el) in reversed(list(zip(perm_z_commitments, perm_evals))[:-1]):      # [UPSTREAM permutation::verifier::Evaluated::queries: sets.iter().rev().skip(1)]
        Q.append((zc, x_last, el))
    for (ai, ti), zc, (z0, z1, a0, am1, s0) in zip(lookups_permuted, lookup_z_commitments, lookup_evals):
        Q += [(zc, x, z0), (ai, x, a0), (ti, x, s0), (ai, x_inv, am1), (zc, x_next, z1)]
    for (c, r), e in zip(sh.fixed_queries, fixed_evals):
        Q.append((fixed_commitments[c], rotate(r), e))
    for sc, e in zip(perm_commitments, sigma_evals):
        Q.append((sc, x, e))
    Q.append((h_commitment, x, expected_h))
    Q.append((random_commitment, x, random_eval))
    # VerifierGWC::verify_proof
    v = T.challenge()
    points, groups = [], {}
    for cm, pt, e in Q:
        if pt not in groups:
            groups[pt] = []
            points.append(pt)
        groups[pt].append((cm, e))
    try:
        ws = [T.read_point() for _ in points]
    except ValueError:
        return False
    if T.pos != len(T.data):
        return False                                     # trailing bytes
    u = T.challenge()
    commitment_multi, eval_multi, witness, witness_with_aux = None, 0, None, None
    pu = 1
    for pt, wi in zip(points, ws):
        cb, eb, pv = None, 0, 1
        for cm, e in groups[pt]:
            cb = po.ec_add(C, cb, po.ec_mul(C, pv, cm))
            eb = (eb + pv * e) % p
            pv = pv * v % p
        commitment_multi = po.ec_add(C, commitment_multi, po.ec_mul(C, pu, cb))
        eval_multi = (eval_multi + pu * eb) % p
        witness_with_aux = po.ec_add(C, witness_with_aux, po.ec_mul(C, pu * pt % p, wi))
        witness = po.ec_add(C, witness, po.ec_mul(C, pu, wi))
        pu = pu * u % p
    right = po.ec_add(C, po.ec_add(C, witness_with_aux, commitment_multi), po.ec_neg(C, po.ec_mul(C, eval_multi, g0)))
    # e(witness, [s]G2) == e(right, G2)
    return pr.pairing_product_is_one([(witness, s_g2), (po.ec_neg(C, right), g2)])
