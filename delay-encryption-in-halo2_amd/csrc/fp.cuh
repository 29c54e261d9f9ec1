// fp.cuh -- 255-bit prime-field arithmetic for gfx950, 8 x u32 Montgomery limbs (R = 2^256).
//
// Restates the VALUES of halo2curves' Fr/Fq/Fp::{add,sub,mul,square,neg,to_repr}
// (upstream halo2curves / pasta_curves) -- not their instruction
// sequence.  In memory an element is the upstream layout, 4 x u64 little-endian limbs in
// Montgomery form = 8 x u32 words here.  Every function returns a fully reduced value
// in [0, p).
//
// CDNA4 notes: there is no 64x64->128 multiply; the widest integer product is
// v_mad_u64_u32 (32x32+64).  The modulus is a template constant, so for the Pasta
// primes (p = 2^254 + t, low word 1, three zero words) the compiler folds the
// Montgomery reduction rows down to 3 real products each (INV = 0xffffffff, P[0] = 1,
// P[4..6] = 0, P[7] = 2^30).  No MFMA: nothing here is a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field_constants.h"

typedef uint32_t u32;
typedef uint64_t u64;

struct alignas(16) fe {
    u32 v[8];
};

#define FP_DEV __device__ __forceinline__

template <class F>
FP_DEV fe f_const(const u32 (&c)[8]) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = c[i];
    return r;
}
template <class F> FP_DEV fe f_one() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = F::ONE_M[i];
    return r;
}
FP_DEV fe f_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
FP_DEV bool f_is_zero(const fe& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
FP_DEV bool f_eq(const fe& a, const fe& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// 16-byte vector loads/stores (2 per element): global_load_dwordx4 / ds_read_b128
FP_DEV fe f_load(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    fe r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
FP_DEV void f_store(fe* p, const fe& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// r = a - p if a >= p else a     (a < 2p, given with its carry-out word `hi`)
template <class F>
FP_DEV fe f_reduce_once(const fe& a, u32 hi) {
    fe d;
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 x = (u64)a.v[i] - F::P[i] - br;
        d.v[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    bool ge = (hi != 0) | (br == 0);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? d.v[i] : a.v[i];
    return r;
}

template <class F>
FP_DEV fe f_add(const fe& a, const fe& b) {
    fe s;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] + b.v[i];
        s.v[i] = (u32)c;
        c >>= 32;
    }
    return f_reduce_once<F>(s, (u32)c);
}

template <class F>
FP_DEV fe f_sub(const fe& a, const fe& b) {
    fe d;
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 x = (u64)a.v[i] - b.v[i] - br;
        d.v[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    u32 mask = (u32)0 - (u32)br;  // borrow -> add p back
    u64 c = 0;
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)d.v[i] + (F::P[i] & mask);
        r.v[i] = (u32)c;
        c >>= 32;
    }
    return r;
}

template <class F>
FP_DEV fe f_neg(const fe& a) {
    fe d;
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 x = (u64)F::P[i] - a.v[i] - br;
        d.v[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    bool z = f_is_zero(a);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = z ? 0u : d.v[i];
    return r;
}

template <class F> FP_DEV fe f_dbl(const fe& a) { return f_add<F>(a, a); }

// Montgomery product a*b*R^-1 mod p.  Operand-scanning CIOS over 32-bit words; every
// step is x = a_j*b_i + t_j + carry, which fits 64 bits exactly, i.e. one
// v_mad_u64_u32 plus the carry add.
template <class F>
FP_DEV fe f_mul(const fe& a, const fe& b) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
        const u32 bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 x = (u64)a.v[j] * bi + t[j] + c;
            t[j] = (u32)x;
            c = x >> 32;
        }
        u64 s = (u64)t[8] + c;
        t[8] = (u32)s;
        u32 t9 = (u32)(s >> 32);
        const u32 m = t[0] * F::INV;
        u64 x = (u64)m * F::P[0] + t[0];
        c = x >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            x = (u64)m * F::P[j] + t[j] + c;
            t[j - 1] = (u32)x;
            c = x >> 32;
        }
        s = (u64)t[8] + c;
        t[7] = (u32)s;
        t[8] = t9 + (u32)(s >> 32);
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return f_reduce_once<F>(r, t[8]);
}

template <class F> FP_DEV fe f_sqr(const fe& a) { return f_mul<F>(a, a); }

// Montgomery -> canonical (upstream to_repr()): a * 1 * R^-1, reduction rows only
template <class F>
FP_DEV fe f_from_mont(const fe& a) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.v[i];
    t[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 m = t[0] * F::INV;
        u64 x = (u64)m * F::P[0] + t[0];
        u64 c = x >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            x = (u64)m * F::P[j] + t[j] + c;
            t[j - 1] = (u32)x;
            c = x >> 32;
        }
        u64 s = (u64)t[8] + c;
        t[7] = (u32)s;
        t[8] = (u32)(s >> 32);
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return f_reduce_once<F>(r, t[8]);
}

template <class F> FP_DEV fe f_to_mont(const fe& a) { return f_mul<F>(a, f_const<F>(F::R2)); }

// a^e for a 256-bit exponent given as 8 runtime words (variable time; setup paths only)
template <class F>
__device__ fe f_pow(const fe& a, const u32* e) {
    fe acc = f_one<F>();
    for (int i = 255; i >= 0; i--) {
        acc = f_sqr<F>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = f_mul<F>(acc, a);
    }
    return acc;
}

// a^-1 by Fermat (a != 0); setup / epilogue paths only
template <class F>
__device__ fe f_inv(const fe& a) {
    u32 e[8];
    u64 br = 2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 x = (u64)F::P[i] - br;
        e[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    return f_pow<F>(a, e);
}
