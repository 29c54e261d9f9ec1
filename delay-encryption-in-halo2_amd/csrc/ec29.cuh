// ec29.cuh -- XYZZ group arithmetic (a = 0 short-Weierstrass: BN254 G1, Pallas, Vesta) on the
// lazily reduced 9 x 29-bit field of fp29.cuh.  Same group elements as halo2curves'
// G1/Ep/Eq::{add, double, add_mixed} [UPSTREAM, SURVEY.md Appendix B]; formulas
// madd-2008-s / add-2008-s / dbl-2008-s-1 (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2).
//
// Lazy-reduction contract (proved by interval analysis in DESIGN.md section 3 and asserted
// limb by limb in tools/fp29_model.py):
//   * stored coordinates have normalized limbs; X, Y < 9.5 p; ZZ, ZZZ < 1.1 p;
//   * products are < 3 p; the identity is ANY point with ZZ = 0 (mod p) -- all-zero limbs
//     when built here, but a result that became the identity through P + (-P) carries
//     ZZ = 0 or p and garbage elsewhere;
//   * exceptional cases (an identity operand, P == Q, P == -Q) all force ZZ3 = 0 (mod p), so
//     one test on ZZ3 (< 2p: equal to 0 or to p) guards the fast path, and only when its
//     one-limb filter fires is anything decided exactly.
#pragma once
#include "ec.cuh"
#include "fp29.cuh"

struct xyzz29 {
    f29 x, y, zz, zzz;
};
struct aff29 {
    f29 x, y;
};

// 144-byte record: 4 x 9 limbs, 16-byte aligned (9 x dwordx4)
struct alignas(16) xyzz29_rec {
    u32 w[36];
};

FP_DEV xyzz29 x29_identity() {
    xyzz29 r;
    r.x = f29_zero(); r.y = f29_zero(); r.zz = f29_zero(); r.zzz = f29_zero();
    return r;
}

FP_DEV xyzz29 x29_load(const xyzz29_rec* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint4 t = q[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
    xyzz29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.v[i] = w[i]; r.y.v[i] = w[9 + i]; r.zz.v[i] = w[18 + i]; r.zzz.v[i] = w[27 + i]; }
    return r;
}
FP_DEV void x29_store(xyzz29_rec* p, const xyzz29& a) {
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) { w[i] = a.x.v[i]; w[9 + i] = a.y.v[i]; w[18 + i] = a.zz.v[i]; w[27 + i] = a.zzz.v[i]; }
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// table entry (internal canonical, packed 2 x 32 B; identity = all zero) -> limbs
FP_DEV aff29 a29_from_packed(const affine_t& p) {
    aff29 r;
    r.x = f29_unpack(p.x);
    r.y = f29_unpack(p.y);
    return r;
}

template <class F>
FP_DEV xyzz29 x29_from_affine(const aff29& q, bool q_is_identity) {
    if (q_is_identity) return x29_identity();
    xyzz29 r;
    r.x = q.x; r.y = q.y; r.zz = f29_one<F>(); r.zzz = f29_one<F>();
    return r;
}

// 2 * A  (identity in -> identity out: ZZ3 = V * ZZ1 = 0 mod p)
template <class F>
FP_DEV xyzz29 x29_double(const xyzz29& a) {
    f29 u = f29_dbl(a.y);                       // limbs < 2^30
    f29 v = f29_sqr<F>(u);
    f29 w = f29_mul<F>(u, v);
    f29 s = f29_mul<F>(a.x, v);
    f29 xx = f29_sqr<F>(a.x);
    f29 m = f29_norm(f29_add(f29_dbl(xx), xx));
    f29 mm = f29_sqr<F>(m);
    xyzz29 r;
    r.x = f29_norm(f29_sub(mm, f29_dbl(s), F::KB));
    f29 t = f29_sub(s, r.x, F::KA);
    r.y = f29_norm(f29_sub(f29_mul<F>(m, t), f29_mul<F>(w, a.y), F::KM));
    r.zz = f29_mul<F>(v, a.zz);
    r.zzz = f29_mul<F>(w, a.zzz);
    return r;
}

// A + q, q affine (internal, canonical or negated-loose y), q != identity
template <class F>
FP_DEV xyzz29 x29_add_mixed(const xyzz29& a, const aff29& q) {
    // independent products are issued in pairs (f29_mul_pair / f29_sqr_pair): interleaved instruction streams
    f29 u2, s2;
    f29_mul_pair<F>(q.x, a.zz, q.y, a.zzz, u2, s2);
    f29 p = f29_norm(f29_sub(u2, a.x, F::KA));
    f29 rr = f29_norm(f29_sub(s2, a.y, F::KA));
    f29 pp, r2;
    f29_sqr_pair<F>(p, rr, pp, r2);
    f29 ppp, qq;
    f29_mul_pair<F>(p, pp, a.x, pp, ppp, qq);
    xyzz29 r;
    r.x = f29_norm(f29_sub(r2, f29_add(ppp, f29_dbl(qq)), F::KB));
    f29 t = f29_sub(qq, r.x, F::KA);
    if constexpr (f29_is_lat<F>::value) {
        r.y = f29_norm(f29_sub(f29_mul<F>(rr, t), f29_mul<F>(a.y, ppp), F::KM));
    } else {
        // R (Q - X3) - Y1 PPP as ONE reduction of two products: R (Q - X3) + (10p - Y1) PPP
        r.y = f29_mul2<F>(rr, t, f29_norm(f29_sub(f29_zero(), a.y, F::KA)), ppp);
    }
    f29_mul_pair<F>(a.zz, pp, a.zzz, ppp, r.zz, r.zzz);
    if (__builtin_expect(f29_maybe_zero_lt2p<F>(r.zz), 0)) {
        if (f29_is_zero_lt2p<F>(r.zz)) {
            // exact resolution (rare): identity accumulator, doubling, or cancellation
            if (f29_is_zero_slow<F>(a.zz)) {
                r.x = q.x; r.y = f29_norm(q.y); r.zz = f29_one<F>(); r.zzz = f29_one<F>();
            } else if (f29_is_zero_slow<F>(rr)) {
                xyzz29 qa;
                qa.x = q.x; qa.y = f29_norm(q.y); qa.zz = f29_one<F>(); qa.zzz = f29_one<F>();
                r = x29_double<F>(qa);
            } else {
                r = x29_identity();
            }
        }
    }
    return r;
}

// A + B
template <class F>
FP_DEV xyzz29 x29_add(const xyzz29& a, const xyzz29& b) {
    if (f29_all_zero(a.zz)) return b;   // literal identities: cheap exit (the general ZZ = 0 mod p case is below)
    if (f29_all_zero(b.zz)) return a;
    f29 u1 = f29_mul<F>(a.x, b.zz);
    f29 u2 = f29_mul<F>(b.x, a.zz);
    f29 s1 = f29_mul<F>(a.y, b.zzz);
    f29 s2 = f29_mul<F>(b.y, a.zzz);
    f29 p = f29_norm(f29_sub(u2, u1, F::KM));
    f29 rr = f29_norm(f29_sub(s2, s1, F::KM));
    f29 pp = f29_sqr<F>(p);
    f29 ppp = f29_mul<F>(p, pp);
    f29 qq = f29_mul<F>(u1, pp);
    f29 r2 = f29_sqr<F>(rr);
    xyzz29 r;
    r.x = f29_norm(f29_sub(r2, f29_add(ppp, f29_dbl(qq)), F::KB));
    f29 t = f29_sub(qq, r.x, F::KA);
    r.y = f29_norm(f29_sub(f29_mul<F>(rr, t), f29_mul<F>(s1, ppp), F::KM));
    r.zz = f29_mul<F>(f29_mul<F>(a.zz, b.zz), pp);
    r.zzz = f29_mul<F>(f29_mul<F>(a.zzz, b.zzz), ppp);
    if (__builtin_expect(f29_maybe_zero_lt2p<F>(r.zz), 0)) {
        if (f29_is_zero_lt2p<F>(r.zz)) {
            if (f29_is_zero_slow<F>(a.zz)) r = b;
            else if (f29_is_zero_slow<F>(b.zz)) r = a;
            else if (f29_is_zero_slow<F>(rr)) r = x29_double<F>(a);
            else r = x29_identity();
        }
    }
    return r;
}

// internal XYZZ -> upstream Jacobian {x, y, z} in standard memory form (Z = ZZ: X' = X*ZZ, Y' = Y*ZZZ)
template <class F>
FP_DEV jacobian_t x29_to_jacobian_std(const xyzz29& p) {
    jacobian_t r;
    if (f29_is_zero_slow<F>(p.zz)) {
        r.x = f_zero(); r.y = f_zero(); r.z = f_zero();
        return r;
    }
    r.x = f29_to_std<F>(f29_mul<F>(p.x, p.zz));
    r.y = f29_to_std<F>(f29_mul<F>(p.y, p.zzz));
    r.z = f29_to_std<F>(p.zz);
    return r;
}

// ---- wave-level reduction (64 lanes -> lane 0) through register shuffles; every lane runs the
// additions (a wave executes them in lock-step anyway), only lane 0's result is meaningful.
template <int W>
FP_DEV xyzz29 x29_shfl_down(const xyzz29& v, int d) {
    xyzz29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.v[i] = __shfl_down(v.x.v[i], d, W);
        r.y.v[i] = __shfl_down(v.y.v[i], d, W);
        r.zz.v[i] = __shfl_down(v.zz.v[i], d, W);
        r.zzz.v[i] = __shfl_down(v.zzz.v[i], d, W);
    }
    return r;
}
// sum over groups of W consecutive lanes (W a power of two <= 64); lane 0 of each group gets it
template <class F, int W>
FP_DEV xyzz29 x29_group_reduce(xyzz29 v) {
    const int gl = threadIdx.x & (W - 1);
    for (int d = W >> 1; d > 0; d >>= 1) {
        xyzz29 o = x29_shfl_down<W>(v, d);
        if (gl + d >= W) o = x29_identity();   // lanes past the fold add nothing (and never see P + P)
        v = x29_add<F>(v, o);
    }
    return v;
}

// ---- quad-cooperative group operations -----------------------------------------------------
// The bucket-reduction tail is a chain of ~45 dependent group additions executed by a handful of
// waves: latency-bound, and a lone wave issues one VALU instruction per ~4 cycles whatever the
// instruction-level parallelism.  Here FOUR adjacent lanes (a DPP quad) hold the same operands
// and share one addition: the 14 multiplications of add-2008-s fall into 4 dependency levels of
// <= 4 independent products, so each lane computes one product per level and the quad exchanges
// results with quad_perm broadcasts (v_mov_b32_dpp, no LDS).  4 multiplication rounds instead
// of 14, at 4x the lanes -- free while the chip is otherwise idle.  All four lanes of a quad
// must be active and return the same (replicated) result.
template <int K>
FP_DEV f29 f29_quad_bcast(const f29& a) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 t = (u32)__builtin_amdgcn_update_dpp(0, (int)a.v[i], K * 0x55, 0xf, 0xf, false);
        // keep the broadcast a plain v_mov_b32_dpp: hipcc (ROCm 7.2) otherwise folds some of them into
        // v_sub(rev)_u32_dpp consumers and x29_double_quad then returns a wrong y (tools/test_quad2.hip)
        asm volatile("" : "+v"(t));
        r.v[i] = t;
    }
    return r;
}
FP_DEV f29 f29_sel4(const f29& a, const f29& b, const f29& c, const f29& d, u32 role) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 lo = (role & 1) ? b.v[i] : a.v[i];
        u32 hi = (role & 1) ? d.v[i] : c.v[i];
        r.v[i] = (role & 2) ? hi : lo;
    }
    return r;
}

template <class F>
FP_DEV xyzz29 x29_add_quad(const xyzz29& a, const xyzz29& b) {
    // literal identities (empty buckets, x29_identity()) are the common case in sparse columns
    if (f29_all_zero(a.zz)) return b;
    if (f29_all_zero(b.zz)) return a;
    const u32 role = threadIdx.x & 3;
    // level 1: u1 = X1 ZZ2 | u2 = X2 ZZ1 | s1 = Y1 ZZZ2 | s2 = Y2 ZZZ1
    f29 m = f29_mul<F>(f29_sel4(a.x, b.x, a.y, b.y, role), f29_sel4(b.zz, a.zz, b.zzz, a.zzz, role));
    f29 u1 = f29_quad_bcast<0>(m), u2 = f29_quad_bcast<1>(m), s1 = f29_quad_bcast<2>(m), s2 = f29_quad_bcast<3>(m);
    f29 p = f29_norm(f29_sub(u2, u1, F::KM));
    f29 rr = f29_norm(f29_sub(s2, s1, F::KM));
    // level 2: pp = P^2 | r2 = R^2 | zz12 = ZZ1 ZZ2 | zzz12 = ZZZ1 ZZZ2
    m = f29_mul<F>(f29_sel4(p, rr, a.zz, a.zzz, role), f29_sel4(p, rr, b.zz, b.zzz, role));
    f29 pp = f29_quad_bcast<0>(m), r2 = f29_quad_bcast<1>(m);
    f29 keep = m;                                   // lanes 2 / 3 keep zz12 / zzz12
    // level 3: ppp = P PP | qq = U1 PP | zz3 = zz12 PP | (lane 3: spare)
    m = f29_mul<F>(f29_sel4(p, u1, keep, p, role), pp);
    f29 ppp = f29_quad_bcast<0>(m), qq = f29_quad_bcast<1>(m);
    xyzz29 r;
    r.zz = f29_quad_bcast<2>(m);
    r.x = f29_norm(f29_sub(r2, f29_add(ppp, f29_dbl(qq)), F::KB));
    f29 t = f29_sub(qq, r.x, F::KA);
    // level 4: R T | S1 PPP | (lane 2: spare) | zzz3 = zzz12 PPP
    m = f29_mul<F>(f29_sel4(rr, s1, rr, keep, role), f29_sel4(t, ppp, t, ppp, role));
    f29 rt = f29_quad_bcast<0>(m), sp = f29_quad_bcast<1>(m);
    r.zzz = f29_quad_bcast<3>(m);
    r.y = f29_norm(f29_sub(rt, sp, F::KM));
    if (__builtin_expect(f29_maybe_zero_lt2p<F>(r.zz), 0)) {
        if (f29_is_zero_lt2p<F>(r.zz)) r = x29_add<F>(a, b);   // exceptional cases: every lane resolves them alone (identically)
    }
    return r;
}

template <class F>
FP_DEV xyzz29 x29_double_quad(const xyzz29& a) {
    if (f29_all_zero(a.zz)) return a;
    const u32 role = threadIdx.x & 3;
    f29 u = f29_dbl(a.y);
    // level 1: v = U^2 | xx = X^2
    f29 op = f29_sel4(u, a.x, u, a.x, role);
    f29 m = f29_mul<F>(op, op);
    f29 v = f29_quad_bcast<0>(m), xx = f29_quad_bcast<1>(m);
    f29 mm_in = f29_norm(f29_add(f29_dbl(xx), xx));
    // level 2: w = U V | s = X V | mm = M^2 | zz3 = V ZZ
    m = f29_mul<F>(f29_sel4(u, a.x, mm_in, v, role), f29_sel4(v, v, mm_in, a.zz, role));
    f29 w = f29_quad_bcast<0>(m), s = f29_quad_bcast<1>(m), mm = f29_quad_bcast<2>(m);
    xyzz29 r;
    r.zz = f29_quad_bcast<3>(m);
    r.x = f29_norm(f29_sub(mm, f29_dbl(s), F::KB));
    f29 t = f29_sub(s, r.x, F::KA);
    // level 3: M T | W Y | zzz3 = W ZZZ | (lane 3: spare)
    m = f29_mul<F>(f29_sel4(mm_in, w, w, w, role), f29_sel4(t, a.y, a.zzz, a.y, role));
    f29 mt = f29_quad_bcast<0>(m), wy = f29_quad_bcast<1>(m);
    r.zzz = f29_quad_bcast<2>(m);
    r.y = f29_norm(f29_sub(mt, wy, F::KM));
    return r;
}

// sum over groups of W consecutive lanes (W a power of two, 8 <= W <= 64) whose QUADS hold replicated values: log2(W / 4) levels of
// quad-cooperative additions (4 multiplication rounds each) instead of log2(W) full additions on single lanes; quad 0 of each group gets it.
template <class F, int W>
FP_DEV xyzz29 x29_group_reduce_quad(xyzz29 v) {
    const int gl = threadIdx.x & (W - 1);
    for (int d = W >> 1; d >= 4; d >>= 1) {
        xyzz29 o = x29_shfl_down<W>(v, d);
        if (gl + d >= W) o = x29_identity();   // quads past the fold add nothing (and never see P + P)
        v = x29_add_quad<F>(v, o);
    }
    return v;
}
