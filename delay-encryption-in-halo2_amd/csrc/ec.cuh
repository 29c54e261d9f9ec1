// ec.cuh -- short-Weierstrass a = 0 group arithmetic (BN254 G1, Pallas, Vesta) for gfx950.
//
// Restates the VALUES of halo2curves' G1/Ep/Eq::{add, double, add_mixed, to_affine}
// (upstream halo2curves): the group element is what must match,
// not the coordinate system.  Accumulators use extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity ZZ = 0) because the bucket loop is all
// mixed additions: 8M + 2S against 7M + 4S for Jacobian, and no field inversion.
// Host-visible formats stay upstream's: affine {x, y} with identity (0, 0);
// projective = Jacobian {x, y, z}, identity z = 0.
#pragma once
#include "fp.cuh"
#include "fp29.cuh"   // f29_inv_safegcd

struct alignas(16) affine_t {
    fe x, y;
};
struct alignas(16) xyzz_t {
    fe x, y, zz, zzz;
};
struct alignas(16) jacobian_t {
    fe x, y, z;
};

FP_DEV affine_t aff_load(const affine_t* p) {
    affine_t r;
    r.x = f_load(&p->x);
    r.y = f_load(&p->y);
    return r;
}
FP_DEV void aff_store(affine_t* p, const affine_t& a) {
    f_store(&p->x, a.x);
    f_store(&p->y, a.y);
}
FP_DEV bool aff_is_identity(const affine_t& a) { return f_is_zero(a.x) && f_is_zero(a.y); }

FP_DEV xyzz_t xyzz_identity() {
    xyzz_t r;
    r.x = f_zero(); r.y = f_zero(); r.zz = f_zero(); r.zzz = f_zero();
    return r;
}
FP_DEV bool xyzz_is_identity(const xyzz_t& p) { return f_is_zero(p.zz); }
FP_DEV xyzz_t xyzz_load(const xyzz_t* p) {
    xyzz_t r;
    r.x = f_load(&p->x); r.y = f_load(&p->y); r.zz = f_load(&p->zz); r.zzz = f_load(&p->zzz);
    return r;
}
FP_DEV void xyzz_store(xyzz_t* p, const xyzz_t& a) {
    f_store(&p->x, a.x); f_store(&p->y, a.y); f_store(&p->zz, a.zz); f_store(&p->zzz, a.zzz);
}
template <class F>
FP_DEV xyzz_t xyzz_from_affine(const affine_t& a) {
    xyzz_t r;
    if (aff_is_identity(a)) return xyzz_identity();
    r.x = a.x; r.y = a.y; r.zz = f_one<F>(); r.zzz = f_one<F>();
    return r;
}

// 2*(x, y) for an affine non-identity point  (mdbl-2008-s-1, a = 0)
template <class F>
FP_DEV xyzz_t xyzz_double_affine(const affine_t& p) {
    xyzz_t r;
    fe u = f_dbl<F>(p.y);
    fe v = f_sqr<F>(u);
    fe w = f_mul<F>(u, v);
    fe s = f_mul<F>(p.x, v);
    fe xx = f_sqr<F>(p.x);
    fe m = f_add<F>(f_dbl<F>(xx), xx);
    r.x = f_sub<F>(f_sqr<F>(m), f_dbl<F>(s));
    r.y = f_sub<F>(f_mul<F>(m, f_sub<F>(s, r.x)), f_mul<F>(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P  (dbl-2008-s-1, a = 0)
template <class F>
FP_DEV xyzz_t xyzz_double(const xyzz_t& p) {
    if (xyzz_is_identity(p)) return p;
    xyzz_t r;
    fe u = f_dbl<F>(p.y);
    fe v = f_sqr<F>(u);
    fe w = f_mul<F>(u, v);
    fe s = f_mul<F>(p.x, v);
    fe xx = f_sqr<F>(p.x);
    fe m = f_add<F>(f_dbl<F>(xx), xx);
    r.x = f_sub<F>(f_sqr<F>(m), f_dbl<F>(s));
    r.y = f_sub<F>(f_mul<F>(m, f_sub<F>(s, r.x)), f_mul<F>(w, p.y));
    r.zz = f_mul<F>(v, p.zz);
    r.zzz = f_mul<F>(w, p.zzz);
    return r;
}

// acc + (x2, y2)   (madd-2008-s: 8M + 2S); handles identity on either side, P == Q, P == -Q
template <class F>
FP_DEV xyzz_t xyzz_add_mixed(const xyzz_t& acc, const affine_t& q) {
    if (aff_is_identity(q)) return acc;
    if (xyzz_is_identity(acc)) {
        xyzz_t r;
        r.x = q.x; r.y = q.y; r.zz = f_one<F>(); r.zzz = f_one<F>();
        return r;
    }
    fe u2 = f_mul<F>(q.x, acc.zz);
    fe s2 = f_mul<F>(q.y, acc.zzz);
    fe p = f_sub<F>(u2, acc.x);
    fe r_ = f_sub<F>(s2, acc.y);
    if (f_is_zero(p)) {
        if (f_is_zero(r_)) return xyzz_double_affine<F>(q);
        return xyzz_identity();
    }
    fe pp = f_sqr<F>(p);
    fe ppp = f_mul<F>(p, pp);
    fe qq = f_mul<F>(acc.x, pp);
    xyzz_t r;
    r.x = f_sub<F>(f_sub<F>(f_sqr<F>(r_), ppp), f_dbl<F>(qq));
    r.y = f_sub<F>(f_mul<F>(r_, f_sub<F>(qq, r.x)), f_mul<F>(acc.y, ppp));
    r.zz = f_mul<F>(acc.zz, pp);
    r.zzz = f_mul<F>(acc.zzz, ppp);
    return r;
}

// P + Q  (add-2008-s: 12M + 2S)
template <class F>
FP_DEV xyzz_t xyzz_add(const xyzz_t& a, const xyzz_t& b) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    fe u1 = f_mul<F>(a.x, b.zz);
    fe u2 = f_mul<F>(b.x, a.zz);
    fe s1 = f_mul<F>(a.y, b.zzz);
    fe s2 = f_mul<F>(b.y, a.zzz);
    fe p = f_sub<F>(u2, u1);
    fe r_ = f_sub<F>(s2, s1);
    if (f_is_zero(p)) {
        if (f_is_zero(r_)) return xyzz_double<F>(a);
        return xyzz_identity();
    }
    fe pp = f_sqr<F>(p);
    fe ppp = f_mul<F>(p, pp);
    fe qq = f_mul<F>(u1, pp);
    xyzz_t r;
    r.x = f_sub<F>(f_sub<F>(f_sqr<F>(r_), ppp), f_dbl<F>(qq));
    r.y = f_sub<F>(f_mul<F>(r_, f_sub<F>(qq, r.x)), f_mul<F>(s1, ppp));
    r.zz = f_mul<F>(f_mul<F>(a.zz, b.zz), pp);
    r.zzz = f_mul<F>(f_mul<F>(a.zzz, b.zzz), ppp);
    return r;
}

// XYZZ -> Jacobian without inversion: Z = ZZ  =>  X' = X*ZZ, Y' = Y*ZZZ   (ZZ^3 = ZZZ^2)
template <class F>
FP_DEV jacobian_t xyzz_to_jacobian(const xyzz_t& p) {
    jacobian_t r;
    if (xyzz_is_identity(p)) {
        r.x = f_zero(); r.y = f_zero(); r.z = f_zero();
        return r;
    }
    r.x = f_mul<F>(p.x, p.zz);
    r.y = f_mul<F>(p.y, p.zzz);
    r.z = p.zz;
    return r;
}

// XYZZ -> affine (one inversion per point: safegcd, ~20x fewer instructions than the 32-bit Fermat chain)
template <class F>
__device__ affine_t xyzz_to_affine(const xyzz_t& p) {
    affine_t r;
    if (xyzz_is_identity(p)) {
        r.x = f_zero(); r.y = f_zero();
        return r;
    }
    typedef typename f29_of<F>::type F9;
    fe zi = f29_to_std<F9>(f29_inv_safegcd<F9>(f29_from_std<F9>(p.zzz)));   // 1/ZZZ
    fe t = f_mul<F>(zi, p.zz);          // ZZ/ZZZ = 1/Z
    fe zz_inv = f_sqr<F>(t);            // 1/Z^2 = 1/ZZ
    r.x = f_mul<F>(p.x, zz_inv);
    r.y = f_mul<F>(p.y, zi);
    return r;
}
