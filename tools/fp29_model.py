#!/usr/bin/env python3
"""Bit-accurate Python model of csrc/fp29.cuh: 9 x 29-bit limb, carry-free, lazily reduced
Montgomery arithmetic (R' = 2^261) and the XYZZ formulas built on it.

Every u32 limb and u64 accumulator is range-asserted, so running the group formulas on random
and adversarial inputs proves the absence of overflow for the bounds the kernels rely on, and
the results are checked against oracle/pyoracle.py.  Also emits the per-field constants
(csrc/fp29_constants.h).  Dev-time tool; not part of the product path.

    python tools/fp29_model.py            # self-test
    python tools/fp29_model.py --emit     # write csrc/fp29_constants.h
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import pyoracle as po

B = 29
L = 9
MASK = (1 << B) - 1
RBITS = B * L  # 261
U32 = 1 << 32
U64 = 1 << 64


def limbs(x):
    assert 0 <= x < (1 << RBITS)
    return [(x >> (B * i)) & MASK for i in range(L)]


def value(l):
    return sum(v << (B * i) for i, v in enumerate(l))


class F29:
    def __init__(self, f: po.Field):
        self.f = f
        self.p = f.p
        self.P = limbs(f.p)
        self.INV = (-pow(f.p, -1, 1 << B)) % (1 << B)
        self.R = (1 << RBITS) % f.p            # Montgomery one
        self.R2 = pow(1 << RBITS, 2, f.p)
        # converts: standard form (x * 2^256) <-> internal (x * 2^261)
        self.TO29 = (1 << (RBITS + (RBITS - 256))) % f.p      # mont(m, TO29) = m * 2^5
        self.FROM29 = (1 << 256) % f.p                          # mont(x, FROM29) = x * 2^-5
        # subtraction constants: multiples of p whose limb representation dominates the subtrahend
        # (value bounds, in units of p, are the fixpoint of the XYZZ formulas: see DESIGN.md)
        self.KM = self.sub_const(3.0, 29)    # subtrahend: a product (normalized limbs, value < 3 p)
        self.KA = self.sub_const(9.5, 29)    # subtrahend: a stored coordinate (normalized limbs, value < 9.5 p)
        self.KB = self.sub_const(4.5, 31)    # subtrahend: PPP + 2Q (limbs < 3 * 2^29, value < 4.5 p)
        self.KN = self.sub_const(1.0, 29)    # negation of a canonical value (< p)

    def sub_const(self, vbound, s):
        """limbs K'_i of a multiple K of p such that K'_i >= (max subtrahend limb_i) for i < 8 and
        K'_8 >= (max subtrahend top limb); offsets 2^s borrowed from the next limb."""
        top_need = (int(vbound * self.p) >> (B * 8)) + 1
        borrow = 1 << (s - B)
        k = 1
        while True:
            K = k * self.p
            kl = limbs(K) if K < (1 << RBITS) else None
            if kl is not None and kl[8] - borrow >= top_need:
                break
            k += 1
        out = [kl[0] + (1 << s)] + [kl[i] + (1 << s) - borrow for i in range(1, 8)] + [kl[8] - borrow]
        assert value(out) == K and all(0 <= v < U32 for v in out)
        return out, K

    # ---- primitives (mirroring the HIP code) ----
    def mont(self, a, b):
        """a * b * 2^-261 mod p, not fully reduced.  Result limbs normalized (< 2^29), value < a*b/2^261 + p."""
        assert len(a) == L and len(b) == L
        assert all(0 <= x < U32 for x in a) and all(0 <= x < U32 for x in b)
        acc = [0] * (2 * L)
        for i in range(L):
            for j in range(L):
                acc[i + j] += a[i] * b[j]
                assert acc[i + j] < U64, "product column overflow"
            m = ((acc[i] & 0xFFFFFFFF) * self.INV) & MASK
            for j in range(L):
                acc[i + j] += m * self.P[j]
                assert acc[i + j] < U64, "reduction column overflow"
            assert acc[i] & MASK == 0
            acc[i + 1] += acc[i] >> B
            assert acc[i + 1] < U64
        out = []
        carry = 0
        for j in range(L):
            v = acc[L + j] + carry
            assert v < U64
            if j < L - 1:
                out.append(v & MASK)
                carry = v >> B
            else:
                assert v < (1 << B), "result exceeds 2^261"
                out.append(v)
        return out

    def mont2(self, a, b, c, d):
        """(a * b + c * d) * 2^-261 mod p with ONE reduction (f29_mul2): both products go into the same columns."""
        for x in (a, b, c, d):
            assert len(x) == L and all(0 <= v < U32 for v in x)
        acc = [0] * (2 * L)
        for i in range(L):
            for j in range(L):
                acc[i + j] += a[i] * b[j] + c[i] * d[j]
        # column totals including every reduction term and carry must fit 64 bits (product scanning keeps ONE accumulator per column)
        carry = 0
        m = [0] * L
        tot = list(acc)
        out = []
        for k in range(2 * L - 1):
            col = tot[k] + carry
            for i in range(L):
                if i < k and k - i < L and i < L:
                    col += m[i] * self.P[k - i] if i < L and k - i >= 1 else 0
            if k < L:
                m[k] = ((col & 0xFFFFFFFF) * self.INV) & MASK
                col += m[k] * self.P[0]
                assert col & MASK == 0
            assert col < U64, "fused column overflow"
            if k >= L:
                out.append(col & MASK)
            carry = col >> B
        assert carry < (1 << B), "result exceeds 2^261"
        out.append(carry)
        want = (value(a) * value(b) + value(c) * value(d)) * pow(1 << RBITS, -1, self.p) % self.p
        assert value(out) % self.p == want
        assert value(out) < (value(a) * value(b) + value(c) * value(d)) // (1 << RBITS) + self.p + 1
        return out

    def sqr(self, a):
        """a^2 * 2^-261 with cross products taken once against the doubled operand (f29_sqr)."""
        assert all(0 <= x < U32 for x in a)
        d = [2 * x for x in a]
        assert all(x < U32 for x in d)
        acc = [0] * (2 * L)
        for i in range(L):
            acc[2 * i] += a[i] * a[i]
            assert acc[2 * i] < U64
            for j in range(i + 1, L):
                acc[i + j] += d[i] * a[j]
                assert acc[i + j] < U64, "square column overflow"
        for i in range(L):
            m = ((acc[i] & 0xFFFFFFFF) * self.INV) & MASK
            for j in range(L):
                acc[i + j] += m * self.P[j]
                assert acc[i + j] < U64, "reduction column overflow"
            assert acc[i] & MASK == 0
            acc[i + 1] += acc[i] >> B
            assert acc[i + 1] < U64
        out, carry = [], 0
        for j in range(L):
            v = acc[L + j] + carry
            if j < L - 1:
                out.append(v & MASK)
                carry = v >> B
            else:
                assert v < (1 << B)
                out.append(v)
        assert out == self.mont(a, a)
        return out

    def norm(self, a):
        assert all(x + 8 < U32 for x in a), "normalize: u32 limb + carry overflow"
        out, carry = [], 0
        for j in range(L):
            v = a[j] + carry
            assert v < U64
            if j < L - 1:
                out.append(v & MASK)
                carry = v >> B
            else:
                assert v < (1 << B), "normalize: value exceeds 2^261"
                out.append(v)
        return out

    def add(self, a, b):
        out = [x + y for x, y in zip(a, b)]
        assert all(v < U32 for v in out)
        return out

    def dbl(self, a):
        return self.add(a, a)

    def sub(self, a, b, kc):
        K, _ = kc
        out = []
        for x, y, k in zip(a, b, K):
            assert k >= y, "subtraction constant does not dominate the subtrahend limb"
            v = x + k - y
            assert v < U32
            out.append(v)
        return out

    def canon(self, a):
        """fully reduced value in [0, p), normalized limbs"""
        v = value(self.norm(a)) % self.p
        return limbs(v)

    def is_zero_mod_p(self, a_norm, max_mult):
        v = value(a_norm)
        assert v < (max_mult + 1) * self.p, "zero-test candidate set too small"
        return any(v == k * self.p for k in range(max_mult + 1))

    # ---- codecs ----
    def enc(self, x):       # integer -> internal Montgomery limbs (canonical)
        return limbs(x * (1 << RBITS) % self.p)

    def dec(self, l):
        return value(l) * pow(1 << RBITS, -1, self.p) % self.p

    def from_std(self, m256):   # standard-form packed value (x * 2^256 mod p) -> internal
        return self.mont(limbs(m256), limbs(self.TO29))

    def to_std(self, l):        # internal -> standard-form canonical integer
        return value(self.canon(self.mont(self.norm(l), limbs(self.FROM29))))


class XYZZ29:
    """Group formulas exactly as csrc/ec29.cuh evaluates them (value bounds tracked as maxima)."""

    def __init__(self, curve: po.Curve):
        self.c = curve
        self.F = F29(curve.base)
        self.maxv = {}

    def track(self, name, l):
        v = value(l)
        self.maxv[name] = max(self.maxv.get(name, 0), v)

    def identity(self):
        z = [0] * L
        return (z, z, z, z)

    def from_affine(self, P):
        F = self.F
        if P is None:
            return self.identity()
        one = limbs(F.R)
        return (F.enc(P[0]), F.enc(P[1]), one, one)

    def to_affine(self, A):
        F = self.F
        zz = F.dec(A[2])
        if zz == 0:
            return None
        p = F.p
        return (F.dec(A[0]) * pow(zz, -1, p) % p, F.dec(A[1]) * pow(F.dec(A[3]), -1, p) % p)

    def madd(self, A, q, neg=False):
        """A + q (q affine canonical internal limbs, or None)."""
        F = self.F
        if q is None:
            return A
        X1, Y1, ZZ1, ZZZ1 = A
        x2, y2 = q
        if neg:
            y2 = F.sub([0] * L, y2, F.KN)          # p' - y, loose
        U2 = F.mont(x2, ZZ1)
        S2 = F.mont(y2, ZZZ1)
        P = F.norm(F.sub(U2, X1, F.KA))
        Rr = F.norm(F.sub(S2, Y1, F.KA))
        PP = F.sqr(P)
        PPP = F.mont(P, PP)
        Q = F.mont(X1, PP)
        RR = F.sqr(Rr)
        X3 = F.norm(F.sub(RR, F.add(PPP, F.dbl(Q)), F.KB))
        T = F.sub(Q, X3, F.KA)
        NY = F.norm(F.sub([0] * L, Y1, F.KA))           # -Y1 + 10p
        Y3 = F.mont2(Rr, T, NY, PPP)                     # R (Q - X3) - Y1 PPP, one reduction
        ZZ3 = F.mont(ZZ1, PP)
        ZZZ3 = F.mont(ZZZ1, PPP)
        for n, v in (("P", P), ("X3", X3), ("Y3", Y3), ("ZZ3", ZZ3), ("ZZZ3", ZZZ3), ("T", T), ("PPP", PPP)):
            self.track(n, v)
        if F.is_zero_mod_p(ZZ3, 1):
            # slow path: decide exactly
            a_id = value(F.canon(ZZ1)) == 0
            if a_id:
                one = limbs(F.R)
                return (x2, F.canon(y2), one, one)
            if value(F.canon(Rr)) == 0:
                return self.dbl_affine((x2, F.canon(y2)))
            return self.identity()
        return (X3, Y3, ZZ3, ZZZ3)

    def dbl_affine(self, q):
        one = limbs(self.F.R)
        return self.dbl((q[0], q[1], one, one))

    def dbl(self, A):
        F = self.F
        X1, Y1, ZZ1, ZZZ1 = A
        U = F.dbl(Y1)                       # limbs < 2^30
        V = F.sqr(U)
        W = F.mont(U, V)
        S = F.mont(X1, V)
        XX = F.sqr(X1)
        M = F.norm(F.add(F.dbl(XX), XX))
        MM = F.sqr(M)
        X3 = F.norm(F.sub(MM, F.dbl(S), F.KB))
        T = F.sub(S, X3, F.KA)
        Y3 = F.norm(F.sub(F.mont(M, T), F.mont(W, Y1), F.KM))
        ZZ3 = F.mont(V, ZZ1)
        ZZZ3 = F.mont(W, ZZZ1)
        for n, v in (("dX3", X3), ("dY3", Y3), ("dZZ3", ZZ3)):
            self.track(n, v)
        return (X3, Y3, ZZ3, ZZZ3)

    def add(self, A, Bp):
        F = self.F
        X1, Y1, ZZ1, ZZZ1 = A
        X2, Y2, ZZ2, ZZZ2 = Bp
        U1 = F.mont(X1, ZZ2)
        U2 = F.mont(X2, ZZ1)
        S1 = F.mont(Y1, ZZZ2)
        S2 = F.mont(Y2, ZZZ1)
        P = F.norm(F.sub(U2, U1, F.KM))
        Rr = F.norm(F.sub(S2, S1, F.KM))
        PP = F.sqr(P)
        PPP = F.mont(P, PP)
        Q = F.mont(U1, PP)
        RR = F.sqr(Rr)
        X3 = F.norm(F.sub(RR, F.add(PPP, F.dbl(Q)), F.KB))
        T = F.sub(Q, X3, F.KA)
        Y3 = F.norm(F.sub(F.mont(Rr, T), F.mont(S1, PPP), F.KM))
        ZZ3 = F.mont(F.mont(ZZ1, ZZ2), PP)
        ZZZ3 = F.mont(F.mont(ZZZ1, ZZZ2), PPP)
        for n, v in (("aX3", X3), ("aY3", Y3), ("aZZ3", ZZ3)):
            self.track(n, v)
        if F.is_zero_mod_p(ZZ3, 1):
            if value(F.canon(ZZ1)) == 0:
                return Bp
            if value(F.canon(ZZ2)) == 0:
                return A
            if value(F.canon(Rr)) == 0:
                return self.dbl(A)
            return self.identity()
        return (X3, Y3, ZZ3, ZZZ3)


def self_test():
    rnd = random.Random(1)
    for fname in ("pasta_fp", "pasta_fq", "bn254_fq", "bn254_fr"):
        F = F29(po.FIELDS[fname])
        p = F.p
        # mont on canonical and on maximally loose inputs
        for _ in range(300):
            a, b = rnd.randrange(p), rnd.randrange(p)
            assert F.dec(F.mont(F.enc(a), F.enc(b))) == a * b % p
        worst_a = [(1 << 31) + (1 << 30) - 1] * 8 + [(1 << 26)]     # loose-31 limbs, ~2^258
        worst_b = [MASK] * 8 + [1 << 26]
        F.mont(worst_a, worst_b)
        loose30 = [(1 << 30) - 1] * 8 + [1 << 27]
        F.mont(loose30, loose30)
        # conversions
        for _ in range(50):
            x = rnd.randrange(p)
            std = x * (1 << 256) % p
            assert F.dec(F.from_std(std)) == x
            assert F.to_std(F.enc(x)) == std
        print(fname, "mont ok; KM mult", F.KM[1] // p, "KA mult", F.KA[1] // p, "KB mult", F.KB[1] // p, "KN mult", F.KN[1] // p, "INV", hex(F.INV), "P limbs", [hex(v) for v in F.P])
    for cname in ("pallas", "vesta", "bn254"):
        c = po.CURVES[cname]
        G = XYZZ29(c)
        F = G.F
        pts = po.synth_bases(c, 40)
        enc = lambda P: None if P is None else (F.enc(P[0]), F.enc(P[1]))
        # long random accumulation chains incl. negations, duplicates, identities, cancellations
        for trial in range(6):
            acc = G.identity()
            ref = None
            seq = [rnd.choice(pts) for _ in range(120)]
            seq[5] = seq[4]            # P + P  (doubling branch)
            seq[10] = None             # identity point
            for i, P in enumerate(seq):
                neg = rnd.random() < 0.3
                acc = G.madd(acc, enc(P), neg)
                ref = po.ec_add(c, ref, po.ec_neg(c, P) if neg else P)
                if i % 17 == 0:
                    assert G.to_affine(acc) == ref
            assert G.to_affine(acc) == ref
            # cancel: acc + (-acc_affine) = identity, then continue
            a_aff = G.to_affine(acc)
            acc2 = G.madd(acc, enc(a_aff), True)
            assert G.to_affine(acc2) is None
            acc2 = G.madd(acc2, enc(pts[3]), False)
            assert G.to_affine(acc2) == pts[3]
            # full adds and doublings
            other = G.identity()
            oref = None
            for P in seq[:30]:
                other = G.madd(other, enc(P))
                oref = po.ec_add(c, oref, P)
            s = G.add(acc, other)
            assert G.to_affine(s) == po.ec_add(c, ref, oref)
            assert G.to_affine(G.add(acc, acc)) == po.ec_add(c, ref, ref)
            assert G.to_affine(G.add(acc, G.identity())) == ref
            assert G.to_affine(G.add(G.identity(), acc)) == ref
            d = acc
            dref = ref
            for _ in range(20):
                d = G.dbl(d)
                dref = po.ec_add(c, dref, dref)
            assert G.to_affine(d) == dref
            assert G.to_affine(G.dbl(G.identity())) is None
        print(cname, "xyzz ok; max value / p:", {k: round(v / c.base.p, 2) for k, v in sorted(G.maxv.items())})


def emit():
    out = ["// GENERATED by tools/fp29_model.py --emit -- do not edit.",
           "// 9 x 29-bit limb constants (R' = 2^261) for csrc/fp29.cuh.", "#pragma once", "#include <stdint.h>", ""]
    names = {"bn254_fr": "Bn254Fr", "bn254_fq": "Bn254Fq", "pasta_fp": "PastaFp", "pasta_fq": "PastaFq"}
    w = lambda l: ", ".join("0x%08xu" % v for v in l)
    for fname, sn in names.items():
        F = F29(po.FIELDS[fname])
        out += ["struct %s29 {" % sn,
                "  typedef %s Std;" % sn,
                "  static constexpr uint32_t INV = 0x%08xu;          // -p^-1 mod 2^29" % F.INV,
                "  static constexpr uint32_t P[9] = {%s};" % w(F.P),
                "  static constexpr uint32_t ONE[9] = {%s};   // 2^261 mod p" % w(limbs(F.R)),
                "  static constexpr uint32_t TO29[9] = {%s};  // mont(std, TO29) = std * 2^5" % w(limbs(F.TO29)),
                "  static constexpr uint32_t FROM29[9] = {%s}; // mont(x, FROM29) = x * 2^-5" % w(limbs(F.FROM29)),
                "  static constexpr uint32_t KM[9] = {%s};  // %d p, dominates normalized limbs of a value < 3 p" % (w(F.KM[0]), F.KM[1] // F.p),
                "  static constexpr uint32_t KA[9] = {%s};  // %d p, dominates normalized limbs of a value < 9.5 p" % (w(F.KA[0]), F.KA[1] // F.p),
                "  static constexpr uint32_t KB[9] = {%s};  // %d p, dominates limbs < 3 * 2^29" % (w(F.KB[0]), F.KB[1] // F.p),
                "  static constexpr uint32_t KN[9] = {%s};  // %d p, negation of a canonical value" % (w(F.KN[0]), F.KN[1] // F.p),
                "  static constexpr uint32_t P2[9] = {%s};  // 2p" % w(limbs(2 * F.p)),
                "  static constexpr uint32_t P4[9] = {%s};  // 4p" % w(limbs(4 * F.p)),
                "  static constexpr uint32_t P8[9] = {%s};  // 8p" % w(limbs(8 * F.p)),
                "  static constexpr uint32_t R3[9] = {%s};  // 2^783 mod p: mont(v, R3) = v * 2^522 (integer inverse -> internal form)" % w(limbs(pow(2, 783, F.p))),
                "  static constexpr uint32_t PINV30 = 0x%08xu;       // p^-1 mod 2^30 (safegcd inversion, 30-bit signed limbs)" % pow(F.p, -1, 1 << 30),
                "};", ""]
    out += ["template <class F> struct f29_of;"]
    for fname, sn in names.items():
        out += ["template <> struct f29_of<%s> { typedef %s29 type; };" % (sn, sn)]
    out += [""]
    path = os.path.join(HERE, "..", "delay-encryption-in-halo2_amd", "csrc", "fp29_constants.h")
    open(path, "w").write("\n".join(out))
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    if "--emit" in sys.argv:
        emit()
    else:
        self_test()
