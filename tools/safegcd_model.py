"""Bit-accurate Python model of f29_inv_safegcd (csrc/poly.cuh): int32 limbs, int64 accumulators, range asserts.
Dev tool: python tools/safegcd_model.py"""
# bit-accurate model of the 30-bit-limb safegcd inversion (Bernstein-Yang divsteps, constant time)
import sys, random
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import pyoracle as po
M30 = (1 << 30) - 1
def s32(x):  # wrap to int32
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x
def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x
def to30(x):
    return [(x >> (30 * i)) & M30 for i in range(9)]
def val30(l):
    return sum(v << (30 * i) for i, v in enumerate(l))
def divsteps_30(zeta, f0, g0):
    u, v, q, r = 1, 0, 0, 1
    f, g = f0 & 0xFFFFFFFF, g0 & 0xFFFFFFFF
    for _ in range(30):
        c1 = (zeta >> 31) & 0xFFFFFFFF if zeta >= 0 else 0xFFFFFFFF   # arithmetic shift of int32
        c1 = 0xFFFFFFFF if zeta < 0 else 0
        c2 = (-(g & 1)) & 0xFFFFFFFF
        x = ((f ^ c1) - c1) & 0xFFFFFFFF; y = ((u ^ c1) - c1) & 0xFFFFFFFF; z = ((v ^ c1) - c1) & 0xFFFFFFFF
        g = (g + (x & c2)) & 0xFFFFFFFF; q = (q + (y & c2)) & 0xFFFFFFFF; r = (r + (z & c2)) & 0xFFFFFFFF
        c1 &= c2
        zeta = s32((zeta ^ s32(c1)) - 1)
        f = (f + (g & c1)) & 0xFFFFFFFF; u = (u + (q & c1)) & 0xFFFFFFFF; v = (v + (r & c1)) & 0xFFFFFFFF
        g >>= 1; u = (u << 1) & 0xFFFFFFFF; v = (v << 1) & 0xFFFFFFFF
    return zeta, (s32(u), s32(v), s32(q), s32(r))
def update_de(d, e, t, P30, pinv30):
    u, v, q, r = t
    sd = -1 if d[8] < 0 else 0; se = -1 if e[8] < 0 else 0
    md = (u & sd) + (v & se); me = (q & sd) + (r & se)
    md = s32(md); me = s32(me)
    cd = s64(u * d[0] + v * e[0]); ce = s64(q * d[0] + r * e[0])
    md = s32(md - ((pinv30 * (cd & 0xFFFFFFFF) + md) & M30))
    me = s32(me - ((pinv30 * (ce & 0xFFFFFFFF) + me) & M30))
    cd = s64(cd + P30[0] * md); ce = s64(ce + P30[0] * me)
    assert cd & M30 == 0 and ce & M30 == 0
    cd >>= 30; ce >>= 30
    nd, ne = [0] * 9, [0] * 9
    for i in range(1, 9):
        cd = s64(cd + u * d[i] + v * e[i] + P30[i] * md)
        ce = s64(ce + q * d[i] + r * e[i] + P30[i] * me)
        nd[i - 1] = cd & M30; cd >>= 30
        ne[i - 1] = ce & M30; ce >>= 30
    nd[8] = s32(cd); ne[8] = s32(ce)
    assert -(1 << 31) <= cd < (1 << 31) and -(1 << 31) <= ce < (1 << 31)
    return nd, ne
def update_fg(f, g, t):
    u, v, q, r = t
    cf = s64(u * f[0] + v * g[0]); cg = s64(q * f[0] + r * g[0])
    assert cf & M30 == 0 and cg & M30 == 0
    cf >>= 30; cg >>= 30
    nf, ng = [0] * 9, [0] * 9
    for i in range(1, 9):
        cf = s64(cf + u * f[i] + v * g[i]); cg = s64(cg + q * f[i] + r * g[i])
        nf[i - 1] = cf & M30; cf >>= 30
        ng[i - 1] = cg & M30; cg >>= 30
    nf[8] = s32(cf); ng[8] = s32(cg)
    return nf, ng
def sval(l):  # signed value: limbs 0..7 unsigned 30-bit, limb 8 signed
    return sum(v << (30 * i) for i, v in enumerate(l))
def normalize(r, sign_neg, P30):
    # r in (-2p, p); add p if negative; negate if sign_neg; add p if negative again
    r = list(r)
    def addp(r, cond):
        if cond: r = [a + b for a, b in zip(r, P30)]
        return r
    def prop(r):
        r = list(r)
        for i in range(8):
            r[i + 1] += r[i] >> 30; r[i] &= M30
        return r
    r = prop(addp(r, r[8] < 0))
    if sign_neg:
        r = [-x for x in r]
        r = prop(r)
    r = prop(addp(r, r[8] < 0))
    return r
def modinv(x, p):
    P30 = to30(p); pinv30 = pow(p, -1, 1 << 30)
    f, g = to30(p), to30(x)
    d, e = [0] * 9, [1] + [0] * 8
    zeta = -1
    for _ in range(20):
        zeta, t = divsteps_30(zeta, f[0], g[0])
        d, e = update_de(d, e, t, P30, pinv30)
        f, g = update_fg(f, g, t)
    assert sval(g) == 0, "g != 0 after 600 divsteps"
    assert sval(f) in (1, -1)
    out = normalize(d, f[8] < 0, P30)
    return sval(out)
for fname in ("bn254_fr", "bn254_fq", "pasta_fp", "pasta_fq"):
    p = po.FIELDS[fname].p
    rnd = random.Random(5)
    for x in [1, 2, p - 1, p - 2, (p - 1) // 2, 1 << 200] + [rnd.randrange(1, p) for _ in range(300)]:
        got = modinv(x, p)
        assert 0 <= got < p and got * x % p == 1, (fname, x, got)
    print(fname, "safegcd ok")
