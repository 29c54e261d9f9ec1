// fp29.cuh -- carry-free, lazily reduced Montgomery arithmetic for gfx950:
// 9 limbs x 29 bits in u32 registers, R' = 2^261, 64-bit in-place column accumulators.
//
// Why not 8 x 32-bit limbs (fp.cuh): on CDNA4 v_mad_u64_u32 issues in ~4 cycles per wave64,
// but it has no carry-in, and every carry step of a 32-bit-limb CIOS costs as much as the
// multiply itself (v_lshl_add_u64 ~4.5 cycles, v_add_co/v_addc ~4.75 each, plus v_mov to build
// zero-extended register pairs: the 32-bit f_mul compiles to 128 mads + 135 64-bit adds +
// 297 moves; profiles/r01_ubench_instruction_rates.txt).  With 29-bit limbs a 64-bit
// accumulator absorbs every product of a column (<= 9 products of <= 2^60 plus 9 reduction
// terms of <= 2^58 < 2^64) with the multiply-add itself: D = a_i * b_j + D, in place, no carry
// instruction, no moves.  81 + 81 mads (81 + 45 for the Pasta primes, whose limbs 5..7 are
// zero) against 64 + 64, but nothing else in the inner loop.
//
// Values are kept lazily reduced: a product is < a*b/2^261 + p, sums are limb-wise adds,
// differences add a multiple of p whose limbs dominate the subtrahend's (constants KM/KA/KB/KN,
// derived and range-checked by tools/fp29_model.py, a bit-accurate Python model of this file
// that asserts every u32 limb and u64 accumulator bound on random and adversarial inputs).
//
// Restates the VALUES of halo2curves' field mul/add/sub [UPSTREAM, SURVEY.md Appendix B]; the
// memory format at the C ABI stays upstream's (4 x u64, R = 2^256): from_std / to_std convert.
#pragma once
#include "fp.cuh"
#include "fp29_constants.h"

#define F29_BITS 29
#define F29_MASK 0x1fffffffu

struct f29 {
    u32 v[9];
};

template <class F>
FP_DEV f29 f29_const(const u32 (&c)[9]) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c[i];
    return r;
}
FP_DEV f29 f29_zero() {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
template <class F> FP_DEV f29 f29_one() { return f29_const<F>(F::ONE); }

FP_DEV bool f29_all_zero(const f29& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.v[i];
    return o == 0;
}

// 8 x 32-bit words (a value < 2^256) -> 9 x 29-bit limbs
FP_DEV f29 f29_unpack(const fe& w) {
    f29 r;
    r.v[0] = w.v[0] & F29_MASK;
    r.v[1] = ((w.v[0] >> 29) | (w.v[1] << 3)) & F29_MASK;
    r.v[2] = ((w.v[1] >> 26) | (w.v[2] << 6)) & F29_MASK;
    r.v[3] = ((w.v[2] >> 23) | (w.v[3] << 9)) & F29_MASK;
    r.v[4] = ((w.v[3] >> 20) | (w.v[4] << 12)) & F29_MASK;
    r.v[5] = ((w.v[4] >> 17) | (w.v[5] << 15)) & F29_MASK;
    r.v[6] = ((w.v[5] >> 14) | (w.v[6] << 18)) & F29_MASK;
    r.v[7] = ((w.v[6] >> 11) | (w.v[7] << 21)) & F29_MASK;
    r.v[8] = w.v[7] >> 8;
    return r;
}
// normalized limbs of a value < 2^256 -> 8 words
FP_DEV fe f29_pack(const f29& a) {
    fe w;
    w.v[0] = a.v[0] | (a.v[1] << 29);
    w.v[1] = (a.v[1] >> 3) | (a.v[2] << 26);
    w.v[2] = (a.v[2] >> 6) | (a.v[3] << 23);
    w.v[3] = (a.v[3] >> 9) | (a.v[4] << 20);
    w.v[4] = (a.v[4] >> 12) | (a.v[5] << 17);
    w.v[5] = (a.v[5] >> 15) | (a.v[6] << 14);
    w.v[6] = (a.v[6] >> 18) | (a.v[7] << 11);
    w.v[7] = (a.v[7] >> 21) | (a.v[8] << 8);
    return w;
}

// carry propagation: limbs < 2^32 (value < 2^261) -> limbs < 2^29
FP_DEV f29 f29_norm(const f29& a) {
    f29 r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 t = a.v[i] + c;        // a.v[i] <= 2^32 - 2^29: see the limb bounds in fp29_model.py
        r.v[i] = t & F29_MASK;
        c = t >> F29_BITS;
    }
    r.v[8] = a.v[8] + c;
    return r;
}

FP_DEV f29 f29_add(const f29& a, const f29& b) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
FP_DEV f29 f29_dbl(const f29& a) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << 1;
    return r;
}
// a - b + K  (K = k*p in a limb form that dominates b's limbs; no borrow can occur)
FP_DEV f29 f29_sub(const f29& a, const f29& b, const u32 (&K)[9]) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (K[i] - b.v[i]);
    return r;
}

FP_DEV u64 mad_wide(u32 a, u32 b, u64 c) { return (u64)a * b + c; }

// Operand scanning (18 independent column accumulators).  More instructions
// than product scanning (a carry add per column) but no dependent multiply-add chain, hence none
// of the s_nop wait states hipcc must put between dependent v_mad_u64_u32 (about 130 per
// multiplication, 4 cycles each when a wave runs alone).  Same values.
template <class F>
FP_DEV f29 f29_mul_os(const f29& a, const f29& b) {
    u64 acc[18];
#pragma unroll
    for (int i = 0; i < 18; i++) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) acc[i + j] = mad_wide(a.v[i], b.v[j], acc[i + j]);
        u32 m;
        if (F::INV == F29_MASK) m = (0u - (u32)acc[i]) & F29_MASK;   // p = 1 mod 2^29 (Pasta)
        else m = ((u32)acc[i] * F::INV) & F29_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (F::P[j] != 0) acc[i + j] = mad_wide(m, F::P[j], acc[i + j]);
        acc[i + 1] += acc[i] >> F29_BITS;
    }
    f29 r;
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u64 t = acc[9 + j] + c;
        r.v[j] = (u32)t & F29_MASK;
        c = t >> F29_BITS;
    }
    r.v[8] = (u32)(acc[17] + c);
    return r;
}

// a^2, latency schedule
template <class F>
FP_DEV f29 f29_sqr_os(const f29& a) {
    u64 acc[18];
#pragma unroll
    for (int i = 0; i < 18; i++) acc[i] = 0;
    u32 d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        // column contributions whose smaller index is i: a_i^2 and 2 a_i a_j (j > i)
        acc[2 * i] = mad_wide(a.v[i], a.v[i], acc[2 * i]);
#pragma unroll
        for (int j = i + 1; j < 9; j++) acc[i + j] = mad_wide(d[i], a.v[j], acc[i + j]);
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 m;
        if (F::INV == F29_MASK) m = (0u - (u32)acc[i]) & F29_MASK;
        else m = ((u32)acc[i] * F::INV) & F29_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (F::P[j] != 0) acc[i + j] = mad_wide(m, F::P[j], acc[i + j]);
        acc[i + 1] += acc[i] >> F29_BITS;
    }
    f29 r;
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u64 t = acc[9 + j] + c;
        r.v[j] = (u32)t & F29_MASK;
        c = t >> F29_BITS;
    }
    r.v[8] = (u32)(acc[17] + c);
    return r;
}

// Schedule selection: f29_lat<F> marks a field whose multiplications use the operand-scanning
// schedule.  Chosen per kernel by measurement: the MSM's merge / bucket-reduction kernels (group
// additions on ~170 live registers at 2-3 waves per SIMD) run 8 % faster with it; the bucket
// accumulation, the NTT and even the lone-wave Fermat inversion are faster with product scanning.
template <class F9> struct f29_lat : F9 { static constexpr bool LATENCY = true; };
template <class F, class = void> struct f29_is_lat { static constexpr bool value = false; };
template <class F> struct f29_is_lat<F, decltype((void)F::LATENCY)> { static constexpr bool value = true; };

// One multiply-add step of a column.  The empty asm pins the association order: without it LLVM
// reassociates the 64-bit sums and the carry of the previous column comes back as a separate
// v_lshl_add_u64 (4.4 cycles) instead of being the addend of the column's first v_mad_u64_u32.
FP_DEV void f29_col_mad(u64& acc, u32 a, u32 b) {
    acc = mad_wide(a, b, acc);
    asm("" : "+v"(acc));
}
// the modulus limbs as opaque scalar registers: keeps hipcc from strength-reducing m * P[j] for
// the sparse Pasta limbs (1, 2^22) into 64-bit shift/add sequences that issue slower than the mad
template <class F>
FP_DEV void f29_load_p(u32 (&P)[9]) {
#pragma unroll
    for (int j = 0; j < 9; j++) {
        P[j] = F::P[j];
        if (F::P[j] != 0) asm("" : "+s"(P[j]));
    }
}
// Montgomery reduction interleaved column by column (product scanning).  Column k of
// a*b + sum_i m_i p 2^(29 i) is accumulated in ONE 64-bit register that starts as the carry of
// column k-1, so the only non-multiply work per column is m_k (k < 9) or the 29-bit mask
// (k >= 9) and one 64-bit shift.  Same values as operand scanning; measured on MI355X
// (tools/ubench_mul.hip): Pasta 172 -> 200 Gmul/s, BN254 172 -> 175.
// Column bound: <= 9 products of <= 2^60 + 9 reduction terms of <= 2^58 + carry < 2^64.
template <class F, class PROD>
FP_DEV f29 f29_montgomery_columns(const PROD& products) {
    u32 m[9], P[9];
    f29_load_p<F>(P);
    f29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        products(k, acc);
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (i < k && k - i < 9 && F::P[k - i] != 0) f29_col_mad(acc, m[i], P[k - i]);
        if (k < 9) {
            if (F::INV == F29_MASK) m[k] = (0u - (u32)acc) & F29_MASK;   // p = 1 mod 2^29 (Pasta)
            else m[k] = ((u32)acc * F::INV) & F29_MASK;
            f29_col_mad(acc, m[k], P[0]);
        } else {
            r.v[k - 9] = (u32)acc & F29_MASK;
        }
        acc >>= F29_BITS;
    }
    r.v[8] = (u32)acc;
    return r;
}

// as f29_col_mad, and the machine scheduler may not move anything across it: keeps two interleaved
// chains interleaved (left alone, the scheduler re-serialises them)
FP_DEV void f29_col_mad_pin(u64& acc, u32 a, u32 b) {
    acc = mad_wide(a, b, acc);
    asm("" : "+v"(acc));
    __builtin_amdgcn_sched_barrier(0);
}
// Two independent reductions in lockstep: the instruction stream alternates between the two column
// accumulators, so a multiply-add never directly follows the one it depends on (hipcc otherwise has
// to separate dependent v_mad_u64_u32 by s_nop wait states: ~130 per multiplication).
template <class F, class PROD1, class PROD2>
FP_DEV void f29_montgomery_columns2(const PROD1& products1, const PROD2& products2, f29& r1, f29& r2) {
    u32 m1[9], m2[9], P[9];
    f29_load_p<F>(P);
    u64 x = 0, y = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        products1(k, x);
        products2(k, y);
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (i < k && k - i < 9 && F::P[k - i] != 0) { f29_col_mad_pin(x, m1[i], P[k - i]); f29_col_mad_pin(y, m2[i], P[k - i]); }
        if (k < 9) {
            if (F::INV == F29_MASK) { m1[k] = (0u - (u32)x) & F29_MASK; m2[k] = (0u - (u32)y) & F29_MASK; }
            else { m1[k] = ((u32)x * F::INV) & F29_MASK; m2[k] = ((u32)y * F::INV) & F29_MASK; }
            f29_col_mad_pin(x, m1[k], P[0]);
            f29_col_mad_pin(y, m2[k], P[0]);
        } else {
            r1.v[k - 9] = (u32)x & F29_MASK;
            r2.v[k - 9] = (u32)y & F29_MASK;
        }
        x >>= F29_BITS;
        y >>= F29_BITS;
    }
    r1.v[8] = (u32)x;
    r2.v[8] = (u32)y;
}
// r1 = a * b, r2 = c * d (each * 2^-261), interleaved
template <class F>
FP_DEV void f29_mul_pair(const f29& a, const f29& b, const f29& c, const f29& d, f29& r1, f29& r2) {
    if constexpr (f29_is_lat<F>::value) { r1 = f29_mul_os<F>(a, b); r2 = f29_mul_os<F>(c, d); }
    else {
        // the products of one column alternate between the two accumulators too
        u32 m1[9], m2[9], P[9];
        f29_load_p<F>(P);
        u64 x = 0, y = 0;
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (k - i >= 0 && k - i < 9) { f29_col_mad_pin(x, a.v[i], b.v[k - i]); f29_col_mad_pin(y, c.v[i], d.v[k - i]); }
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (i < k && k - i < 9 && F::P[k - i] != 0) { f29_col_mad_pin(x, m1[i], P[k - i]); f29_col_mad_pin(y, m2[i], P[k - i]); }
            if (k < 9) {
                if (F::INV == F29_MASK) { m1[k] = (0u - (u32)x) & F29_MASK; m2[k] = (0u - (u32)y) & F29_MASK; }
                else { m1[k] = ((u32)x * F::INV) & F29_MASK; m2[k] = ((u32)y * F::INV) & F29_MASK; }
                f29_col_mad_pin(x, m1[k], P[0]);
                f29_col_mad_pin(y, m2[k], P[0]);
            } else {
                r1.v[k - 9] = (u32)x & F29_MASK;
                r2.v[k - 9] = (u32)y & F29_MASK;
            }
            x >>= F29_BITS;
            y >>= F29_BITS;
        }
        r1.v[8] = (u32)x;
        r2.v[8] = (u32)y;
    }
}
// r1 = a^2, r2 = c^2, interleaved
template <class F>
FP_DEV void f29_sqr_pair(const f29& a, const f29& c, f29& r1, f29& r2) {
    if constexpr (f29_is_lat<F>::value) { r1 = f29_sqr_os<F>(a); r2 = f29_sqr_os<F>(c); }
    else {
        u32 da[9], dc[9];
#pragma unroll
        for (int i = 0; i < 9; i++) { da[i] = a.v[i] << 1; dc[i] = c.v[i] << 1; }
        f29_montgomery_columns2<F>(
            [&](int k, u64& acc) {
#pragma unroll
                for (int i = 0; i < 9; i++)
                    if (k - i > i && k - i < 9) f29_col_mad_pin(acc, da[i], a.v[k - i]);
                if ((k & 1) == 0 && k / 2 < 9) f29_col_mad_pin(acc, a.v[k / 2], a.v[k / 2]);
            },
            [&](int k, u64& acc) {
#pragma unroll
                for (int i = 0; i < 9; i++)
                    if (k - i > i && k - i < 9) f29_col_mad_pin(acc, dc[i], c.v[k - i]);
                if ((k & 1) == 0 && k / 2 < 9) f29_col_mad_pin(acc, c.v[k / 2], c.v[k / 2]);
            },
            r1, r2);
    }
}

// a * b * 2^-261 mod p, lazily reduced: result limbs normalized, value < a*b/2^261 + p.
// Limb-size contract: bits(max a limb) + bits(max b limb) <= 60.
template <class F>
FP_DEV f29 f29_mul(const f29& a, const f29& b) {
    if constexpr (f29_is_lat<F>::value) return f29_mul_os<F>(a, b);
    else return f29_montgomery_columns<F>([&](int k, u64& acc) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) f29_col_mad(acc, a.v[i], b.v[k - i]);
    });
}

// (a * b + c * d) * 2^-261 with ONE reduction: both products go into the same columns
// (result < (a b + c d) / 2^261 + p).  Column budget: limbs of a, c, d < 2^29 and of b < 2^31:
// 9 * 2^60 + 9 * 2^58 products + 9 * 2^58 reduction terms + carry < 2^64 (tools/fp29_model.py mont2).
template <class F>
FP_DEV f29 f29_mul2(const f29& a, const f29& b, const f29& c, const f29& d) {
    return f29_montgomery_columns<F>([&](int k, u64& acc) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) {
                f29_col_mad(acc, a.v[i], b.v[k - i]);
                f29_col_mad(acc, c.v[i], d.v[k - i]);
            }
    });
}

// a^2 * 2^-261: cross products once, against the doubled operand (limbs < 2^30 for a normalized a)
template <class F>
FP_DEV f29 f29_sqr(const f29& a) {
    if constexpr (f29_is_lat<F>::value) return f29_sqr_os<F>(a);
    u32 d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    return f29_montgomery_columns<F>([&](int k, u64& acc) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i > i && k - i < 9) f29_col_mad(acc, d[i], a.v[k - i]);
        if ((k & 1) == 0 && k / 2 < 9) f29_col_mad(acc, a.v[k / 2], a.v[k / 2]);
    });
}

// v >= c ? v - c : v   on normalized limbs (c a normalized constant)
FP_DEV f29 f29_cond_sub(const f29& a, const u32 (&C)[9]) {
    f29 d;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t t = (int32_t)a.v[i] - (int32_t)C[i] + borrow;
        d.v[i] = (u32)t & F29_MASK;
        borrow = t >> 31;
    }
    int32_t top = (int32_t)a.v[8] - (int32_t)C[8] + borrow;
    d.v[8] = (u32)top;
    bool ge = top >= 0;
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = ge ? d.v[i] : a.v[i];
    return r;
}

// fully reduced representative in [0, p) of a normalized value < 16 p  (slow paths, stores)
template <class F>
FP_DEV f29 f29_canon(const f29& a) {
    f29 r = f29_cond_sub(a, F::P8);
    r = f29_cond_sub(r, F::P4);
    r = f29_cond_sub(r, F::P2);
    r = f29_cond_sub(r, F::P);
    return r;
}

// is a normalized value < 2p congruent to 0?  (v == 0 or v == p)
template <class F>
FP_DEV bool f29_is_zero_lt2p(const f29& a) {
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { z |= a.v[i]; e |= a.v[i] ^ F::P[i]; }
    return (z == 0) | (e == 0);
}
// cheap necessary condition for the above (limb 0 only): filters all but ~2^-28 of inputs
template <class F>
FP_DEV bool f29_maybe_zero_lt2p(const f29& a) { return (a.v[0] == 0) | (a.v[0] == F::P[0]); }

// exact zero test of any lazily reduced value (< 16 p, limbs < 2^31): one multiplication by R' mod p
template <class F>
FP_DEV bool f29_is_zero_slow(const f29& a) {
    f29 w = f29_mul<F>(a, f29_one<F>());     // = a mod p, value < (1 + 16 p / 2^261) p < 2p
    return f29_is_zero_lt2p<F>(w);
}

// standard memory form (x * 2^256 mod p, 8 words) <-> internal (x * 2^261, lazily reduced)
template <class F>
FP_DEV f29 f29_from_std(const fe& m) { return f29_mul<F>(f29_unpack(m), f29_const<F>(F::TO29)); }
// internal (any lazily reduced value < 16p with normalized limbs) -> canonical standard form
template <class F>
FP_DEV fe f29_to_std(const f29& a) {
    f29 t = f29_mul<F>(a, f29_const<F>(F::FROM29));   // < 2p
    return f29_pack(f29_cond_sub(t, F::P));
}
// internal lazily reduced -> internal canonical, packed in 8 words (SRS table format)
template <class F>
FP_DEV fe f29_to_packed_canon(const f29& a_norm) { return f29_pack(f29_canon<F>(a_norm)); }

// a^-1 by the Bernstein-Yang "safegcd" algorithm in its constant-time batched form (30 divsteps on
// the low words give a 2x2 transition matrix, which is then applied to the full-width f, g and to
// the Bezout pair d, e modulo p; 20 rounds = 600 divsteps, enough for 256-bit moduli).  Signed
// 30-bit limbs in int32, int64 accumulators.  About 12 k instructions on one dependent chain
// against ~67 k for the Fermat exponentiation above: the latency floor of a batch inversion drops
// from ~0.26 to ~0.06 ms.  The limb arithmetic was first written as a bit-accurate Python model that
// asserts every intermediate range; the parity tests compare the results with Fermat inverses.
struct s30x9 { int32_t v[9]; };
#define S30_MASK 0x3fffffff

template <class F9>
FP_DEV f29 f29_inv_safegcd(const f29& a) {
    typedef typename F9::Std F;
    // the integer V = a mod p (the caller's internal form is just an integer here) and the modulus, as 9 x 30-bit limbs
    const fe wv = f29_pack(f29_canon<F9>(a));
    s30x9 f, g, d, e;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, word = bit >> 5, sh = bit & 31;
        u64 pv = (u64)F::P[word] | (word + 1 < 8 ? (u64)F::P[word + 1] << 32 : 0);
        u64 gv = (u64)wv.v[word] | (word + 1 < 8 ? (u64)wv.v[word + 1] << 32 : 0);
        f.v[i] = (int32_t)((pv >> sh) & S30_MASK);
        g.v[i] = (int32_t)((gv >> sh) & S30_MASK);
        d.v[i] = 0; e.v[i] = 0;
    }
    s30x9 P30 = f;
    e.v[0] = 1;
    int32_t zeta = -1;
    for (int round = 0; round < 20; round++) {
        // ---- 30 divsteps on the low limbs -> transition matrix t = [[u, v], [q, r]] ----
        u32 u = 1, v = 0, q = 0, r = 1;
        u32 fl = (u32)f.v[0] | ((u32)f.v[1] << 30), gl = (u32)g.v[0] | ((u32)g.v[1] << 30);
        for (int i = 0; i < 30; i++) {
            u32 c1 = (u32)(zeta >> 31);
            u32 c2 = 0u - (gl & 1u);
            u32 x = (fl ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
            gl += x & c2; q += y & c2; r += z & c2;
            c1 &= c2;
            zeta = (int32_t)((u32)zeta ^ c1) - 1;
            fl += gl & c1; u += q & c1; v += r & c1;
            gl >>= 1; u <<= 1; v <<= 1;
        }
        const int64_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        // ---- d, e <- t * [d, e] / 2^30 mod p ----
        {
            const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
            int32_t md = ((int32_t)u & sd) + ((int32_t)v & se), me = ((int32_t)q & sd) + ((int32_t)r & se);
            int64_t cd = tu * d.v[0] + tv * e.v[0], ce = tq * d.v[0] + tr * e.v[0];
            md -= (int32_t)((F9::PINV30 * (u32)cd + (u32)md) & S30_MASK);
            me -= (int32_t)((F9::PINV30 * (u32)ce + (u32)me) & S30_MASK);
            cd += (int64_t)P30.v[0] * md; ce += (int64_t)P30.v[0] * me;
            cd >>= 30; ce >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                cd += tu * d.v[i] + tv * e.v[i] + (int64_t)P30.v[i] * md;
                ce += tq * d.v[i] + tr * e.v[i] + (int64_t)P30.v[i] * me;
                d.v[i - 1] = (int32_t)cd & S30_MASK; cd >>= 30;
                e.v[i - 1] = (int32_t)ce & S30_MASK; ce >>= 30;
            }
            d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
        }
        // ---- f, g <- t * [f, g] / 2^30 (exact) ----
        {
            int64_t cf = tu * f.v[0] + tv * g.v[0], cg = tq * f.v[0] + tr * g.v[0];
            cf >>= 30; cg >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                cf += tu * f.v[i] + tv * g.v[i];
                cg += tq * f.v[i] + tr * g.v[i];
                f.v[i - 1] = (int32_t)cf & S30_MASK; cf >>= 30;
                g.v[i - 1] = (int32_t)cg & S30_MASK; cg >>= 30;
            }
            f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
        }
    }
    // g = 0 and f = +-1 now; the inverse is sign(f) * d, brought into [0, p)
    {
        int32_t add = d.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] += P30.v[i] & add;
        const int32_t neg = f.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] = (d.v[i] ^ neg) - neg;
#pragma unroll
        for (int i = 0; i < 8; i++) { d.v[i + 1] += d.v[i] >> 30; d.v[i] &= S30_MASK; }
        add = d.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] += P30.v[i] & add;
#pragma unroll
        for (int i = 0; i < 8; i++) { d.v[i + 1] += d.v[i] >> 30; d.v[i] &= S30_MASK; }
    }
    // 30-bit limbs -> 8 words -> 29-bit limbs; V^-1 * 2^783 * 2^-261 = (x 2^261)^-1 2^522 = x^-1 2^261
    fe out;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, limb = bit / 30, sh = bit % 30;
        u64 lo = (u64)(u32)d.v[limb] | ((u64)(u32)d.v[limb + 1] << 30) | (limb + 2 < 9 ? (u64)(u32)d.v[limb + 2] << 60 : 0);
        out.v[j] = (u32)(lo >> sh);
    }
    return f29_mul<F9>(f29_unpack(out), f29_const<F9>(F9::R3));
}

