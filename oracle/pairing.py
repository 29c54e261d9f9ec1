"""BN254 (alt_bn128 / halo2curves `bn256`) optimal-ate pairing on Python integers.  TEST INFRASTRUCTURE ONLY.

Needed for the one end-to-end check the reference itself makes on the hot path's results: `verify_proof(...)` must accept
what `create_proof` produced (benches/delay_enc.rs:147-165, via VerifierGWC / SingleStrategy [UPSTREAM
halo2_proofs/src/poly/kzg/{multiopen/gwc/verifier.rs, strategy.rs}]).  Written from the textbook construction:
Fq12 = Fq[w] / (w^12 - 18 w^6 + 82), G2 on the sextic twist y^2 = x^3 + 3 / (9 + i) mapped into E(Fq12), Miller loop over
6u + 2 with the two Frobenius line corrections, final exponentiation (q^12 - 1) / r by plain square-and-multiply.
Slow (about a second per pairing) and simple; checked by bilinearity in tests/test_oracle.py.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47      # base field modulus
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001      # group order
ATE_LOOP_COUNT = 29793968203157093288                                          # 6u + 2, u = 4965661367192848881
LOG_ATE_LOOP_COUNT = 63

# G2 generator (x = x0 + x1 i, y = y0 + y1 i)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))
G1 = (1, 2)


# ---- Fq2 = Fq[i] / (i^2 + 1), elements (a, b) = a + b i ------------------------------------------------------
def f2_add(x, y): return ((x[0] + y[0]) % Q, (x[1] + y[1]) % Q)
def f2_sub(x, y): return ((x[0] - y[0]) % Q, (x[1] - y[1]) % Q)
def f2_mul(x, y): return ((x[0] * y[0] - x[1] * y[1]) % Q, (x[0] * y[1] + x[1] * y[0]) % Q)
def f2_neg(x): return (-x[0] % Q, -x[1] % Q)
def f2_inv(x):
    d = pow(x[0] * x[0] + x[1] * x[1], -1, Q)
    return (x[0] * d % Q, -x[1] * d % Q)

B2 = f2_mul((3, 0), f2_inv((9, 1)))          # twist curve constant 3 / (9 + i)


def g2_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == (0, 0)


def g2_add(P, Qp):
    if P is None: return Qp
    if Qp is None: return P
    (x1, y1), (x2, y2) = P, Qp
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_mul((2, 0), y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(k: int, P):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, P)
        P = g2_add(P, P)
        k >>= 1
    return acc


# ---- Fq12 = Fq[w] / (w^12 - 18 w^6 + 82): coefficient lists of length 12 ----------------------------------------
def f12_mul(a: List[int], b: List[int]) -> List[int]:
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for i in range(22, 11, -1):            # w^12 = 18 w^6 - 82
        c = t[i]
        if c:
            t[i - 6] += 18 * c
            t[i - 12] -= 82 * c
    return [x % Q for x in t[:12]]


F12_ONE = [1] + [0] * 11


def f12_pow(a: List[int], e: int) -> List[int]:
    r, base = F12_ONE, a
    while e:
        if e & 1:
            r = f12_mul(r, base)
        base = f12_mul(base, base)
        e >>= 1
    return r


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a: List[int]) -> List[int]:
    """extended Euclid in Fq[w] against the modulus polynomial"""
    mod = [82, 0, 0, 0, 0, 0, -18 % Q, 0, 0, 0, 0, 0, 1]
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], mod
    while _poly_deg(low):
        # r = high / low (polynomial division, rounded to the quotient)
        dl, dh = _poly_deg(low), _poly_deg(high)
        r = [0] * 13
        temp = list(high)
        inv_lead = pow(low[dl], -1, Q)
        for i in range(dh - dl, -1, -1):
            r[i] = temp[dl + i] * inv_lead % Q
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - low[c] * r[i]) % Q
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                new[i + j] = (new[i + j] - low[i] * r[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], -1, Q)
    return [x * inv0 % Q for x in lm[:12]]


def f12_sub(a, b): return [(x - y) % Q for x, y in zip(a, b)]
def f12_add(a, b): return [(x + y) % Q for x, y in zip(a, b)]
def f12_scalar(c: int): return [c % Q] + [0] * 11


W2 = [0, 0, 1] + [0] * 9
W3 = [0, 0, 0, 1] + [0] * 8


def twist(P):
    """E'(Fq2) -> E(Fq12): (x, y) -> (x w^2, y w^3) with i = w^6 - 9."""
    if P is None:
        return None
    (x0, x1), (y0, y1) = P
    nx = [(x0 - 9 * x1) % Q] + [0] * 5 + [x1] + [0] * 5
    ny = [(y0 - 9 * y1) % Q] + [0] * 5 + [y1] + [0] * 5
    return (f12_mul(nx, W2), f12_mul(ny, W3))


def cast_g1(P):
    return (f12_scalar(P[0]), f12_scalar(P[1]))


def _e12_double(P):
    x, y = P
    lam = f12_mul(f12_mul(f12_scalar(3), f12_mul(x, x)), f12_inv(f12_mul(f12_scalar(2), y)))
    nx = f12_sub(f12_sub(f12_mul(lam, lam), x), x)
    return (nx, f12_sub(f12_mul(lam, f12_sub(x, nx)), y))


def _e12_add(P, S):
    if P is None: return S
    if S is None: return P
    (x1, y1), (x2, y2) = P, S
    if x1 == x2:
        return _e12_double(P) if y1 == y2 else None
    lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
    return (nx, f12_sub(f12_mul(lam, f12_sub(x1, nx)), y1))


def _linefunc(P1, P2, T):
    (x1, y1), (x2, y2), (xt, yt) = P1, P2, T
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_mul(f12_scalar(3), f12_mul(x1, x1)), f12_inv(f12_mul(f12_scalar(2), y1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def miller_loop(Q2, P1) -> List[int]:
    """f_{6u+2,Q}(P) with the two Frobenius corrections, WITHOUT the final exponentiation.  Q2 in G2 (Fq2 coords), P1 in G1."""
    if Q2 is None or P1 is None:
        return F12_ONE
    Qt, Pt = twist(Q2), cast_g1(P1)
    Rp, f = Qt, F12_ONE
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rp, Rp, Pt))
        Rp = _e12_double(Rp)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _linefunc(Rp, Qt, Pt))
            Rp = _e12_add(Rp, Qt)
    Q1 = (f12_pow(Qt[0], Q), f12_pow(Qt[1], Q))
    nQ2 = (f12_pow(Q1[0], Q), [(-c) % Q for c in f12_pow(Q1[1], Q)])
    f = f12_mul(f, _linefunc(Rp, Q1, Pt))
    Rp = _e12_add(Rp, Q1)
    f = f12_mul(f, _linefunc(Rp, nQ2, Pt))
    return f


def final_exponentiation(f: List[int]) -> List[int]:
    return f12_pow(f, (Q ** 12 - 1) // R)


def pairing(Q2, P1) -> List[int]:
    return final_exponentiation(miller_loop(Q2, P1))


def pairing_product_is_one(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 for [(P1 in G1, Q2 in G2)]: one final exponentiation for the whole product."""
    f = F12_ONE
    for P1, Q2 in pairs:
        f = f12_mul(f, miller_loop(Q2, P1))
    return final_exponentiation(f) == F12_ONE


# ---- RawBytes of a G2 point (ParamsKZG's g2 / s_g2): x.c0, x.c1, y.c0, y.c1 as Montgomery limbs ----
def g2_to_raw(P) -> bytes:
    Rm = (1 << 256) % Q
    if P is None:
        return bytes(128)
    return b"".join((c * Rm % Q).to_bytes(32, "little") for c in (P[0][0], P[0][1], P[1][0], P[1][1]))


def g2_from_raw(b: bytes):
    Ri = pow(1 << 256, -1, Q)
    c = [int.from_bytes(b[i:i + 32], "little") * Ri % Q for i in range(0, 128, 32)]
    if not any(c):
        return None
    return ((c[0], c[1]), (c[2], c[3]))
