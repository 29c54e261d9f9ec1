// ntt.cuh -- radix-2 NTT over a 255-bit prime field for gfx950 (replaces upstream
// halo2_proofs::arithmetic::best_fft and the EvaluationDomain wrappers around it,
// halo2_proofs/src/arithmetic.rs + src/poly/domain.rs @ v2023_04_20).
//
// Same function as best_fft -- a'[i] = sum_j a[j] * omega^(i*j), natural order in and
// out -- but not its algorithm.  best_fft bit-reverses, then runs log n radix-2 sweeps
// over the whole array; here the transform is factored N = R_1 * R_2 * ... * R_L
// (Cooley-Tukey, L <= 4) so each pass moves every element through HBM once:
//
//   pass p < L : for every sub-problem of size M and every column n', an R_p-point DIT
//                transform along the stride-M/R_p axis inside LDS, times the inter-pass
//                twiddle omega_M^(n' * k), stored back at the same positions;
//   pass L     : R_L-point transforms of contiguous rows; the store performs the digit
//                reversal (k = k_1 + R_1 k_2 + ...), so no separate bit-reverse pass.
//
// A workgroup owns a tile of R x C elements (R*C <= 2048) kept in LDS as lazily reduced 9 x 29-bit
// limbs (72 KiB of the CU's 160 KiB, two workgroups per CU).  Global accesses are C*32-byte
// contiguous segments.  Twiddles omega^j * 2^261 (j < N/2) live in an HBM/L2-resident table built
// once per (field, omega, log_n); the R/2 roots of the sub-transform are staged in LDS.  Every
// multiplication runs on the carry-free multiplier of fp29.cuh.  Pre-/post-scaling of
// lagrange_to_coeff / coeff_to_extended / extended_to_coeff (x n^-1, x zeta^(i mod 3), zero
// padding) is fused into the first load and the last store.  64 B/element algorithmic, but on
// gfx950 the bound is the VALU: log2(N)/2 + L multiplications per element at ~200 Gmul/s is 6x
// the HBM time.  Measured by elimination (23 x 2^19, bn256::Fr, 1.37 ms): butterfly stages 0.63,
// global load + store 0.26, everything else (index math, LDS fill, the per-pass output
// multiplication) 0.39; multiplication floor 0.86 ms.  Two-pass plans (radix 2^10) were measured
// slower than three (1.60 vs 1.34 ms): their 64/128-byte segments cost more than the pass saved.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "fp29.cuh"
#include "internal.hpp"

#ifndef NTT_TILE_LOG
#define NTT_TILE_LOG 11
#endif
#define NTT_TILE (1 << NTT_TILE_LOG)
#define NTT_MAX_PASSES 4

struct NttPassParams {
    const fe* src;
    fe* dst;
    const fe* tw;        // omega^j * 2^261 (canonical, packed), j < N (tw_full) or j < N/2
    u32 log_n;           // N
    u32 log_m;           // sub-problem size at this pass
    u32 r;               // log2 of this pass's radix
    u32 log_c;           // log2 of columns per tile
    u32 is_final;
    u32 skip2;           // first pass of a transform whose input is zero beyond N / 4: the first two stages are copies (see the load)
    u32 tile_log;        // log2 of the elements a workgroup owns (NTT_TILE_LOG, or one less for launches too small to fill the chip: run_ntt_t)
    u32 tw_full;         // the table holds all N powers (log_n <= ntt_full_table_log): the inter-pass twiddle is one load, no negation
    u32 r1;              // log2 of the first pass's radix (final pass store)
    u32 nrev;            // number of middle digits to reverse in the final pass
    u32 rev_r[NTT_MAX_PASSES];  // their log2 radices, most significant first
    u64 src_len;         // elements valid in src per polynomial (zero beyond)
    u64 src_stride;      // elements between consecutive polynomials in src
    u64 dst_stride;
    u32 pre_mode;        // 1: x [1, z, z^2][i % 3] on the very first load (z = pre_z)
    u32 post_mode;       // 1: x post0 ; 2: x post0 * [1, z^2, z][i % 3] on the last store (z = post_z)
    int form_shift;      // +1 / -1: fold 2^5 / 2^-5 into the output multiplication (internal <-> standard form, see dehalo.h)
    fe pre_z;
    fe post0, post_z;
#ifdef DEHALO_EXPERIMENTS
    unsigned long long* stamps;   // measurement only (DEHALO_NTT_STAMPS=1): 4 wall-clock stamps per workgroup -- start, tile in LDS, stages done, stores issued
#endif
};
#ifdef DEHALO_EXPERIMENTS
#define NTT_STAMP(i) do { if (stamp && tid == 0) stamp[i] = wall_clock64(); } while (0)
#else
#define NTT_STAMP(i) do { } while (0)
#endif

FP_DEV u32 bitrev32(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// omega^e for e < N: straight from a full table, or from the half table (omega^(N/2) = -1)
template <class F>
FP_DEV fe tw_lookup(const fe* tw, u64 e, u32 log_n, bool full) {
    if (full) return f_load(&tw[e]);
    u64 half = 1ull << (log_n - 1);
    bool neg = e >= half;
    fe w = f_load(&tw[neg ? e - half : e]);
    return neg ? f_neg<F>(w) : w;
}

// ---- pass kernel ----------------------------------------------------------------------------
// The tile lives in LDS as lazily reduced
// 9 x 29-bit limbs (five planes: four u64 limb pairs + one u32, 36 B per element, conflict-free
// ds_read/write_b64) and the butterflies are decimation-in-time:
//     t = w * v (carry-free multiplier, < 2p);  u' = u + t;  v' = u - t + 4p
// so values grow ADDITIVELY (< p + 4p per stage, < 30p after 7 stages, far inside 2^261) and no
// stage needs a modular reduction -- only a carry normalisation every second stage.  Two stages
// run in registers per LDS round trip (radix 4).  The element is the plain integer of upstream's
// standard form (x * 2^256); twiddles are w * 2^261, so products stay in that form, and the
// multiplication every element needs on the way out (inter-pass twiddle, post-scale, or 1)
// brings it back below 2p for one conditional subtraction.
#ifndef NTT_THREADS
#define NTT_THREADS 512
#endif
// measurement-only variants (tools/ab_ntt.sh rebuilds the four ntt units with one of these; results are WRONG with them, only the time means anything):
//   NTT_X_NOSYNC   no workgroup barrier between the butterfly rounds (an upper bound on what wave-private sub-transforms could win)
//   NTT_X_NOLOAD   the tile is filled from the index instead of from global memory (what hiding the whole load latency could win)
//   NTT_X_NOTW     the inter-pass twiddle of the output multiplication is not loaded
//   NTT_X_LINEAR_FILL  the last pass fills its tile row-wise (no LDS bank conflicts in the transposing fill)
#define NTT_LOADS 4        // elements of a tile per thread (2048 / 512, 1024 / 256)
#ifdef NTT_X_NOSYNC
#define NTT_STAGE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define NTT_STAGE_SYNC() __syncthreads()
#endif

struct Lds29 {
    u64* p01; u64* p23; u64* p45; u64* p67; u32* p8;
};
FP_DEV f29 lds29_load(const Lds29& L, u32 i) {
    f29 r;
    u64 a = L.p01[i], b = L.p23[i], c = L.p45[i], d = L.p67[i];
    r.v[0] = (u32)a; r.v[1] = (u32)(a >> 32); r.v[2] = (u32)b; r.v[3] = (u32)(b >> 32);
    r.v[4] = (u32)c; r.v[5] = (u32)(c >> 32); r.v[6] = (u32)d; r.v[7] = (u32)(d >> 32);
    r.v[8] = L.p8[i];
    return r;
}
FP_DEV void lds29_store(const Lds29& L, u32 i, const f29& v) {
    L.p01[i] = (u64)v.v[0] | ((u64)v.v[1] << 32);
    L.p23[i] = (u64)v.v[2] | ((u64)v.v[3] << 32);
    L.p45[i] = (u64)v.v[4] | ((u64)v.v[5] << 32);
    L.p67[i] = (u64)v.v[6] | ((u64)v.v[7] << 32);
    L.p8[i] = v.v[8];
}

// Where element (row, column c) of the R x C tile sits in a plane: the column is XORed with the row's TOP log2 C bits.  A row is still C consecutive slots (every
// round, the row-wise fills and the store touch whole rows: conflict-free as before), but a COLUMN -- the last pass fills its tile down the columns, 64 lanes with
// consecutive j writing rows bitrev(j) of one column -- no longer lies on two banks (64 LDS cycles per store instruction instead of 4: measured by elimination,
// profiles/r06_ntt_transposing_fill.txt, 1.2-2.7 % of a transform).
#ifdef NTT_NO_SWIZZLE      // (tools/attic/ab_ntt.sh "-DNTT_NO_SWIZZLE": the plain layout of rounds 1-5, for the A/B)
FP_DEV u32 ntt_at(u32 row, u32 c, u32 log_c, u32) { return (row << log_c) + c; }
#else
FP_DEV u32 ntt_at(u32 row, u32 c, u32 log_c, u32 sw_shift) { return (row << log_c) + (c ^ ((row >> sw_shift) & ((1u << log_c) - 1))); }
#endif

// DIT butterfly on lazily reduced limbs; v may carry limbs < 2^31, w is normalized
template <class F9>
FP_DEV void bfly29(f29& u, f29& v, const f29& w, bool mul) {
    f29 t = mul ? f29_mul<F9>(v, w) : f29_norm(v);
    f29 nu = f29_add(u, t);
    v = f29_sub(u, t, F9::KM);
    u = nu;
}

template <class F>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(NttPassParams P) {
    typedef typename f29_of<F>::type F9;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Lds29 L;
    L.p01 = reinterpret_cast<u64*>(smem_raw);
    const u32 plane = 1u << P.tile_log, nthr = blockDim.x;
    L.p23 = L.p01 + plane; L.p45 = L.p23 + plane; L.p67 = L.p45 + plane;
    L.p8 = reinterpret_cast<u32*>(L.p67 + plane);
    f29* ltw = reinterpret_cast<f29*>(L.p8 + plane);  // R/2 sub-transform roots (w * 2^261), unpacked limbs

    const u32 tid = threadIdx.x;
    const u32 r = P.r, log_c = P.log_c;
    const u32 R = 1u << r, Cc = 1u << log_c;
    const u32 tile = R << log_c;
    const u32 sw = r > log_c ? r - log_c : 0;      // ntt_at's shift: the row's top log_c bits
#ifdef DEHALO_EXPERIMENTS
    unsigned long long* stamp = P.stamps ? P.stamps + 4 * ((u64)blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
#endif
    NTT_STAMP(0);
    // (Round 5, measured and removed: re-reading the linear workgroup id so that the batch's copies of one tile -- which gather the SAME inter-pass twiddles -- run
    // back to back on one XCD and share its L2.  23 x 2^19: 1.18-1.20 -> 1.33-1.34 ms.  The copies' data lie exactly N x 32 bytes apart, so the 64 workgroups an XCD has
    // in flight then hammer the same memory channels; profiles/r05_ntt_ab.txt.)
    const u32 bx = blockIdx.x, by = blockIdx.y;
    const fe* src = P.src + (u64)by * P.src_stride;
    fe* dst = P.dst + (u64)by * P.dst_stride;

    for (u32 j = tid; j < (R >> 1); j += nthr) ltw[j] = f29_unpack(f_load(&P.tw[(u64)j << (P.log_n - r)]));

    // ---- tile coordinates ----
    u64 q = 0, np0 = 0;
    u32 k1blk = 0; u64 rest = 0;
    const u32 log_cols = P.log_m - r;
    if (!P.is_final) {
        u32 tiles_per_sub_log = log_cols - log_c;
        q = (u64)bx >> tiles_per_sub_log;
        np0 = ((u64)bx & ((1ull << tiles_per_sub_log) - 1)) << log_c;
    } else {
        u32 kb_log = P.r1 - log_c;
        k1blk = bx & ((1u << kb_log) - 1);
        rest = (u64)bx >> kb_log;
    }
    const u32 log_q_per_k1 = P.log_n - r - P.r1;

    f29 pre1 = f29_zero(), pre2 = f29_zero();
    if (P.pre_mode) {
        pre1 = f29_from_std<F9>(P.pre_z);
        pre2 = f29_mul<F9>(pre1, pre1);
    }

    // ---- load: global (standard form, canonical) -> limbs, into row bitrev(j) (DIT input order) ----
    if (P.skip2) {
        // zero-padded input, at most N / 4 coefficients (coeff_to_extended): only rows j < R / 4 hold anything, they land on rows
        // 4 i of the bit-reversed order, and the first two stages -- (x, 0) -> (x, x), then (x, 0), (x, 0) -> (x, x), (x, x) -- only
        // copy: row 4 i is written to rows 4 i .. 4 i + 3 and the stages start at s = 2 (no loads of zero rows, one LDS round trip,
        // one barrier and a quarter multiplication per element less)
        for (u32 idx = tid; idx < (tile >> 2); idx += nthr) {
            const u32 j = idx >> log_c, c = idx & (Cc - 1);
            const u64 g = (q << P.log_m) + ((u64)j << log_cols) + np0 + c;
            f29 v = f29_zero();
            if (g < P.src_len) {
                v = f29_unpack(f_load(&src[g]));
                if (P.pre_mode) {
                    u32 m3 = (u32)(g % 3);
                    if (m3 == 1) v = f29_mul<F9>(v, pre1);
                    else if (m3 == 2) v = f29_mul<F9>(v, pre2);
                }
            }
            const u32 b0 = bitrev32(j, r);
            lds29_store(L, ntt_at(b0, c, log_c, sw), v); lds29_store(L, ntt_at(b0 + 1, c, log_c, sw), v); lds29_store(L, ntt_at(b0 + 2, c, log_c, sw), v); lds29_store(L, ntt_at(b0 + 3, c, log_c, sw), v);
        }
    } else {
        // a thread's (at most four: tile <= NTT_LOADS x threads by construction, run_ntt_t) elements: all four global loads are issued before the first is unpacked -- one memory round trip per tile instead of four
        // (profiles/r05_ntt_elimination.txt: a tile filled without global loads is the upper bound, 7 % of the batch)
        // (four named values, not an array: an array indexed in a loop the compiler declines to unroll -- the body holds two multiplications -- lands in scratch)
        auto locate = [&](u32 u, u32& slot, u64& g) __attribute__((always_inline)) {
            const u32 idx = tid + u * nthr;
            u32 j, c;
            if (!P.is_final) {
                j = idx >> log_c; c = idx & (Cc - 1);
                g = (q << P.log_m) + ((u64)j << log_cols) + np0 + c;
            } else {
                c = idx >> r; j = idx & (R - 1);
                u64 qq = (((u64)k1blk << log_c) + c) << log_q_per_k1;
                qq += rest;
                g = (qq << r) + j;
            }
#ifdef NTT_X_LINEAR_FILL      // (measurement only, WRONG results: the last pass's fill written row-wise instead of down a column -- what its LDS bank conflicts cost)
            if (P.is_final) { j = idx >> log_c; c = idx & (Cc - 1); }
#endif
            const u32 at = ntt_at(bitrev32(j, r), c, log_c, sw);
            slot = idx < tile && g < P.src_len ? at : (idx < tile ? 0x80000000u | at : 0xffffffffu);
        };
        auto fetch = [&](u32 slot, u64 g) __attribute__((always_inline)) -> fe {
            if (slot & 0x80000000u) return f_zero();      // beyond the tile, or beyond the valid coefficients (zero-extended)
#ifdef NTT_X_NOLOAD
            fe x; for (int i = 0; i < 8; i++) x.v[i] = (u32)g * (2654435761u + i);
            return x;
#else
            return f_load(&src[g]);
#endif
        };
        auto put = [&](const fe& raw, u32 slot, u64 g) __attribute__((always_inline)) {
            if (slot == 0xffffffffu) return;
            lds29_store(L, slot & 0x7fffffffu, f29_unpack(raw));
        };
        static_assert(NTT_LOADS == 4, "");
        static_assert(NTT_TILE <= NTT_LOADS * NTT_THREADS, "a thread fills at most NTT_LOADS slots of a tile: a larger tile / thread ratio would leave LDS slots unfilled (the half-tile launch keeps the ratio)");
        u32 s0, s1, s2, s3;
        u64 g0, g1, g2, g3;
        locate(0, s0, g0); locate(1, s1, g1); locate(2, s2, g2); locate(3, s3, g3);
#ifdef NTT_X_SERIAL_LOADS      // (measurement only: one memory round trip per element, as before round 5)
        { const fe r0 = fetch(s0, g0); put(r0, s0, g0); __builtin_amdgcn_sched_barrier(0); }
        { const fe r1 = fetch(s1, g1); put(r1, s1, g1); __builtin_amdgcn_sched_barrier(0); }
        { const fe r2 = fetch(s2, g2); put(r2, s2, g2); __builtin_amdgcn_sched_barrier(0); }
        { const fe r3 = fetch(s3, g3); put(r3, s3, g3); }
#else
        const fe r0 = fetch(s0, g0), r1 = fetch(s1, g1), r2 = fetch(s2, g2), r3 = fetch(s3, g3);
        put(r0, s0, g0); put(r1, s1, g1); put(r2, s2, g2); put(r3, s3, g3);
#endif
        if (P.pre_mode) {      // x zeta^(i mod 3) of a coset transform whose input is longer than N / 4 (rare: the padded case is the branch above): one loop, one copy of the multiplication
            for (u32 u = 0; u < NTT_LOADS; u++) {
                const u32 sl = u == 0 ? s0 : u == 1 ? s1 : u == 2 ? s2 : s3;
                const u32 m3 = (u32)((u == 0 ? g0 : u == 1 ? g1 : u == 2 ? g2 : g3) % 3);
                if ((sl & 0x80000000u) || m3 == 0) continue;
                lds29_store(L, sl, f29_mul<F9>(lds29_load(L, sl), m3 == 1 ? pre1 : pre2));      // (the thread's own slot: no barrier needed)
            }
        }
    }
    __syncthreads();
    NTT_STAMP(1);

    // ---- DIT stages, two per round ----
    u32 s = P.skip2 ? 2 : 0;
    for (; s + 1 < r; s += 2) {
        const u32 h = 1u << s;
        const u32 ngroups = tile >> 2;   // radix-4 groups in the tile
        for (u32 gidx = tid; gidx < ngroups; gidx += nthr) {
            u32 c = gidx & (Cc - 1), gq = gidx >> log_c;          // gq in [0, R/4)
            u32 pos = gq & (h - 1);
            u32 i0 = ((gq >> s) << (s + 2)) | pos;
            const u32 a0 = ntt_at(i0, c, log_c, sw), a1 = ntt_at(i0 + h, c, log_c, sw), a2 = ntt_at(i0 + 2 * h, c, log_c, sw), a3 = ntt_at(i0 + 3 * h, c, log_c, sw);
            f29 e0 = lds29_load(L, a0), e1 = lds29_load(L, a1), e2 = lds29_load(L, a2), e3 = lds29_load(L, a3);
            // stage s: (e0, e1) and (e2, e3), twiddle w_R^(pos * R / 2h)
            f29 w = ltw[pos << (r - 1 - s)];
            bfly29<F9>(e0, e1, w, s != 0);
            bfly29<F9>(e2, e3, w, s != 0);
            // stage s + 1: (e0, e2) with pos, (e1, e3) with pos + h, twiddle w_R^(pos' * R / 4h);
            // in the first round pos = 0, so the (e0, e2) twiddle is 1
            f29 w0 = ltw[pos << (r - 2 - s)];
            f29 w1 = ltw[(pos + h) << (r - 2 - s)];
            bfly29<F9>(e0, e2, w0, s != 0);
            bfly29<F9>(e1, e3, w1, true);
            lds29_store(L, a0, f29_norm(e0)); lds29_store(L, a1, f29_norm(e1));
            lds29_store(L, a2, f29_norm(e2)); lds29_store(L, a3, f29_norm(e3));
        }
        NTT_STAGE_SYNC();
    }
    if (s < r) {   // odd r: one last radix-2 stage
        const u32 h = 1u << s;
        const u32 nbf = tile >> 1;
        for (u32 bidx = tid; bidx < nbf; bidx += nthr) {
            u32 c = bidx & (Cc - 1), b = bidx >> log_c;
            u32 pos = b & (h - 1);
            u32 i0 = ((b >> s) << (s + 1)) | pos;
            const u32 a0 = ntt_at(i0, c, log_c, sw), a1 = ntt_at(i0 + h, c, log_c, sw);
            f29 e0 = lds29_load(L, a0), e1 = lds29_load(L, a1);
            f29 w = ltw[pos << (r - 1 - s)];
            bfly29<F9>(e0, e1, w, s != 0);
            lds29_store(L, a0, f29_norm(e0)); lds29_store(L, a1, f29_norm(e1));
        }
        __syncthreads();
    }

    NTT_STAMP(2);
    // ---- store: row k holds output digit k; one multiplication brings every element below 2p ----
    f29 post0m = f29_one<F9>(), post1m = f29_zero(), post2m = f29_zero();
    if (P.is_final && (P.post_mode || P.form_shift)) {
        if (P.post_mode) post0m = f29_from_std<F9>(P.post0);
        // the transform is linear, so a change of Montgomery radix is one more factor of the output
        // multiplication: 2^266 (standard -> internal) or 2^256 (internal -> standard), both * 2^-261
        if (P.form_shift > 0) post0m = f29_mul<F9>(post0m, f29_const<F9>(F9::TO29));
        if (P.form_shift < 0) post0m = f29_mul<F9>(post0m, f29_const<F9>(F9::FROM29));
        if (P.post_mode == 2) {
            f29 z = f29_from_std<F9>(P.post_z);
            post2m = f29_mul<F9>(post0m, z);
            post1m = f29_mul<F9>(post2m, z);
        }
    }
    u64 revrest = 0;
    if (P.is_final) {
        u64 rem = rest;
        u32 bits_left = log_q_per_k1, mult = 0;
        for (u32 i = 0; i < P.nrev; i++) {
            bits_left -= P.rev_r[i];
            u64 d = rem >> bits_left;
            rem -= d << bits_left;
            revrest += d << mult;
            mult += P.rev_r[i];
        }
    }
    for (u32 idx = tid; idx < tile; idx += nthr) {
        u32 k = idx >> log_c, c = idx & (Cc - 1);
        f29 v = lds29_load(L, ntt_at(k, c, log_c, sw));
        u64 o;
        f29 w;
        if (!P.is_final) {
            u64 np = np0 + c;
            u64 e = (np * k) << (P.log_n - P.log_m);
#ifdef NTT_X_NOTW
            w = ltw[(u32)e & ((R >> 1) - 1)];
#else
            w = f29_unpack(tw_lookup<F>(P.tw, e, P.log_n, P.tw_full != 0));
#endif
            o = (q << P.log_m) + ((u64)k << log_cols) + np;
        } else {
            o = (((u64)k1blk << log_c) + c) + (revrest << P.r1) + ((u64)k << (P.log_n - r));
            w = post0m;
            if (P.post_mode == 2) {
                u32 m3 = (u32)(o % 3);
                w = m3 == 0 ? post0m : (m3 == 1 ? post1m : post2m);
            }
        }
        f29 t = f29_mul<F9>(v, w);   // < 30p * p / 2^261 + p < 2p
        f_store(&dst[o], f29_pack(f29_cond_sub(t, F9::P)));
    }
    NTT_STAMP(3);
}

// tw[j] = omega^j * 2^261 mod p (canonical, packed), j < half (`half` = the table's length: N for a full table).  Thread t fills a run of 64
// starting from omega^(64 t); the running power is kept in standard form.
template <class F>
__global__ void k_twiddle_gen(fe* tw, fe omega, u64 half) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 start = t * 64;
    if (start >= half) return;
    // omega^start by square-and-multiply
    fe acc = f_one<F>();
    fe base = omega;
    for (u64 e = start; e; e >>= 1) {
        if (e & 1) acc = f_mul<F>(acc, base);
        base = f_sqr<F>(base);
    }
    typedef typename f29_of<F>::type F9;
    const fe c261 = f29_pack(f29_const<F9>(F9::ONE));   // 2^261 mod p as a plain integer
    u64 end = start + 64 < half ? start + 64 : half;
    for (u64 j = start; j < end; j++) {
        f_store(&tw[j], f_mul<F>(acc, c261));               // (w 2^256)(2^261) 2^-256 = w 2^261
        acc = f_mul<F>(acc, omega);
    }
}

// element-wise field ops for the parity tests (dehalo_field_op)
template <class F>
__global__ void k_field_op(int op, const fe* a, const fe* b, fe* out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = f_load(&a[i]);
    fe y = b ? f_load(&b[i]) : x;
    fe r;
    switch (op) {
        case 0: r = f_add<F>(x, y); break;
        case 1: r = f_sub<F>(x, y); break;
        case 2: r = f_mul<F>(x, y); break;
        case 3: r = f_is_zero(x) ? f_zero() : f_inv<F>(x); break;
        case 4: r = f_to_mont<F>(x); break;
        case 5: r = f_from_mont<F>(x); break;
        default: {  // 6: the same product through the carry-free 9 x 29-bit path (fp29.cuh)
            typedef typename f29_of<F>::type F9;
            r = f29_to_std<F9>(f29_mul<F9>(f29_from_std<F9>(x), f29_from_std<F9>(y)));
        } break;
    }
    f_store(&out[i], r);
}

// ==========================================================================================
// host driver (instantiated once per field in ntt_<field>.hip)
// ==========================================================================================
// `full` (optional out): whether the table returned holds all N powers.  A caller that only reads j < N / 2 (evalh.cuh) takes either.
template <class F>
int get_twiddles(dehalo_ctx* ctx, uint32_t log_n, const uint64_t omega[4], hipStream_t s, const fe** out, bool* full = nullptr) {
    // up to 2^ntt_full_table_log the table holds all N powers (32 B x N: 16 MiB at 2^19), beyond that the first half
    const bool want_full = log_n >= 1 && log_n <= (uint32_t)ctx->ntt_full_table_log;
    uint64_t half = log_n == 0 ? 1 : want_full ? (1ull << log_n) : (1ull << (log_n - 1));
    if (full) *full = want_full;
    for (auto& t : ctx->twiddles)
        if (t.field == F::ID && t.log_n == log_n && t.form == 0 && t.len == half && !memcmp(t.omega, omega, 32)) {
            *out = t.tw;
            return 0;
        }
    fe* tw = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&tw, half * sizeof(fe)));
    uint64_t threads = (half + 63) / 64;
    uint32_t blocks = (uint32_t)((threads + 127) / 128);
    k_twiddle_gen<F><<<blocks, 128, 0, s>>>(tw, fe_from_u64(omega), half);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->twiddles.size() >= 16) {  // bounded cache: drop the oldest
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipFree(ctx->twiddles.front().tw));
        ctx->twiddles.erase(ctx->twiddles.begin());
    }
    TwiddleEntry e;
    e.field = F::ID; e.log_n = log_n; e.form = 0; memcpy(e.omega, omega, 32); e.len = half; e.tw = tw;
    ctx->twiddles.push_back(e);
    *out = tw;
    return 0;
}

#ifdef DEHALO_EXPERIMENTS
// measurement only: phases of the workgroups of one pass from their wall-clock stamps (100 MHz), relative to the first workgroup's start
static inline void ntt_report_stamps(dehalo_ctx* ctx, unsigned long long* d, uint64_t nblk, uint32_t pass, uint32_t r, uint32_t tile_log, hipStream_t s) {
    std::vector<unsigned long long> h(nblk * 4);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h.data(), d, nblk * 32, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(d); return; }
    (void)hipFree(d);
    unsigned long long t0 = ~0ull;
    for (uint64_t i = 0; i < nblk; i++) t0 = std::min(t0, h[4 * i]);
    const char* names[4] = {"start", "tile in LDS", "stages done", "stores issued"};
    fprintf(stderr, "ntt pass %u (radix 2^%u, tile 2^%u, %llu workgroups), us after the first workgroup's start: ", pass, r, tile_log, (unsigned long long)nblk);
    for (int k = 0; k < 4; k++) {
        std::vector<double> v(nblk);
        for (uint64_t i = 0; i < nblk; i++) v[i] = (double)(h[4 * i + k] - t0) / 100.0;
        std::sort(v.begin(), v.end());
        fprintf(stderr, "%s min %.1f median %.1f p90 %.1f max %.1f | ", names[k], v[0], v[nblk / 2], v[nblk * 9 / 10], v[nblk - 1]);
    }
    double dl = 0, dc = 0, ds = 0;
    for (uint64_t i = 0; i < nblk; i++) { dl += (double)(h[4 * i + 1] - h[4 * i]); dc += (double)(h[4 * i + 2] - h[4 * i + 1]); ds += (double)(h[4 * i + 3] - h[4 * i + 2]); }
    fprintf(stderr, "mean per workgroup: load %.1f us, stages %.1f us, store %.1f us\n", dl / nblk / 100.0, dc / nblk / 100.0, ds / nblk / 100.0);
}
#endif

// Transforms `batch` polynomials: src (src_len valid elements each, zero-extended to 2^log_n,
// src_stride apart) -> dst (dst_stride apart).  src == dst allowed.
template <class F>
int run_ntt_t(dehalo_ctx* ctx, const fe* src, uint64_t src_len, uint64_t src_stride, fe* dst, uint64_t dst_stride, uint32_t log_n,
              const uint64_t omega[4], size_t batch, const NttScale& sc, hipStream_t s) {
    if (log_n > (uint32_t)F::TWO_ADICITY) return dh_fail(ctx, DEHALO_ERR_UNSUPPORTED, "log_n exceeds the field's two-adicity");
    if (log_n > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "log_n > 30");
    if (batch > 65535) return dh_fail(ctx, DEHALO_ERR_INVALID, "ntt: batch > 65535 (grid.y)");
    if (batch == 0) return 0;
    const fe* tw = nullptr;
    bool tw_full = false;
    TRY(get_twiddles<F>(ctx, log_n, omega, s, &tw, &tw_full));
    ScopedTimer timer(ctx, s, DEHALO_K_NTT_PASS);

    // plan: radices
    uint32_t L, rad[NTT_MAX_PASSES];
    if (log_n <= NTT_TILE_LOG) { L = 1; rad[0] = log_n; }
    else {
        L = (log_n + 7) / 8;
        if (L > NTT_MAX_PASSES) return dh_fail(ctx, DEHALO_ERR_INVALID, "log_n too large");
        uint32_t base = log_n / L, extra = log_n % L;
        for (uint32_t i = 0; i < L; i++) rad[i] = base + (i < extra ? 1 : 0);
    }
    const uint64_t N = 1ull << log_n;
    fe* scratch = nullptr;
    if (L > 1) {
        TRY(dh_ensure(ctx, ctx->ws_ntt_scratch, batch * N * sizeof(fe)));
        scratch = (fe*)ctx->ws_ntt_scratch.p;
    }
    const size_t lds_max = (size_t)NTT_TILE * 36 + (NTT_TILE / 2) * sizeof(f29);
    // per call: the attribute belongs to the device the context is bound to
    HIP_TRY(ctx, dh_func_lds(ctx, (const void*)k_ntt_pass<F>, (int)lds_max));
    // A launch of few tiles (one 2^20 transform: 512 of 2048 elements, two per CU) leaves every CU with two workgroups that load, transform and store in
    // lock-step; half-size tiles on half-size workgroups give it four.  Launches of up to 2^22 elements take them (measured, tools/ab_ntt_tile.sh: 22 x 2^17
    // 0.300 -> 0.280 ms, one 2^20 0.142 -> 0.139, 23 x 2^19 unchanged either way); DEHALO_NTT_SMALL_TILE_LOG = log2 of that bound (0: never) for the A/B.
    static const uint32_t small_log = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_NTT_SMALL_TILE_LOG"); return e ? (uint32_t)atoi(e) : 22u; }();
    const uint32_t tile_log = L > 1 && rad[0] + 1 < NTT_TILE_LOG && batch * N <= (1ull << small_log) ? NTT_TILE_LOG - 1 : NTT_TILE_LOG;
    uint32_t log_m = log_n;
    for (uint32_t p = 0; p < L; p++) {
        NttPassParams P;
        memset(&P, 0, sizeof(P));
        bool first = p == 0, last = p == L - 1;
        P.tw = tw;
        P.log_n = log_n; P.log_m = log_m; P.r = rad[p];
        P.is_final = last ? 1 : 0;
        P.tw_full = tw_full ? 1 : 0;
        // (DEHALO_NTT_SKIP=0: the general first pass, for A/B measurements)
        static const bool skip_ok = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_NTT_SKIP"); return !(e && e[0] == '0'); }();
        P.skip2 = skip_ok && first && !last && rad[p] >= 2 && src_len * 4 <= N ? 1 : 0;
        P.r1 = L > 1 ? rad[0] : 0;
        if (first) { P.src = src; P.src_len = src_len; P.src_stride = src_stride; }
        else { P.src = scratch; P.src_len = N; P.src_stride = N; }
        if (last) { P.dst = dst; P.dst_stride = dst_stride; }
        else { P.dst = scratch; P.dst_stride = N; }
        if (first && sc.pre_mode) { P.pre_mode = sc.pre_mode; P.pre_z = sc.pre_z; }
        if (last && sc.post_mode) { P.post_mode = sc.post_mode; P.post0 = sc.post0; P.post_z = sc.post_z; }
        if (last) P.form_shift = sc.form_shift;
        uint32_t log_c;
        if (!last) {
            uint32_t log_cols = log_m - rad[p];
            log_c = std::min<uint32_t>(tile_log - rad[p], log_cols);
        } else {
            log_c = std::min<uint32_t>(tile_log - rad[p], P.r1);
            P.nrev = L > 2 ? L - 2 : 0;
            for (uint32_t i = 0; i < P.nrev; i++) P.rev_r[i] = rad[1 + i];
        }
        P.log_c = log_c;
        P.tile_log = tile_log;
        uint64_t tiles = N >> (rad[p] + log_c);
        dim3 grid((uint32_t)tiles, (uint32_t)batch);
        size_t lds = ((size_t)36 << tile_log) + ((size_t)1 << rad[p]) / 2 * sizeof(f29);
        if (tile_log < NTT_TILE_LOG) {      // half tiles run in 256-thread blocks: the shape that fits beside a resident accumulation (DEHALO_CO_LDS, internal.hpp)
            lds = dh_co_lds_pad(0, lds);
            TRY(dh_co_lds_attr(ctx, (const void*)k_ntt_pass<F>, lds));
        }
#ifdef DEHALO_EXPERIMENTS
        static const bool stamps_on = getenv("DEHALO_NTT_STAMPS") != nullptr;
        if (stamps_on) HIP_TRY(ctx, hipMalloc((void**)&P.stamps, tiles * batch * 4 * sizeof(unsigned long long)));
#endif
        k_ntt_pass<F><<<grid, NTT_THREADS >> (NTT_TILE_LOG - tile_log), lds, s>>>(P);
        HIP_TRY(ctx, hipGetLastError());
#ifdef DEHALO_EXPERIMENTS
        if (stamps_on) ntt_report_stamps(ctx, P.stamps, tiles * batch, p, rad[p], tile_log, s);
#endif
        log_m -= rad[p];
    }
    return 0;
}

template <class F>
int field_op_t(dehalo_ctx* ctx, int op, const fe* a, const fe* b, fe* out, uint64_t n, hipStream_t s) {
    k_field_op<F><<<(u32)((n + 127) / 128), 128, 0, s>>>(op, a, b, out, n);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

#define DEFINE_NTT_ENTRY(NAME, F)                                                                                                            \
    int run_ntt_##NAME(dehalo_ctx* ctx, const fe* src, uint64_t src_len, uint64_t src_stride, fe* dst, uint64_t dst_stride, uint32_t log_n, \
                       const uint64_t omega[4], size_t batch, const NttScale& sc, hipStream_t s) {                                           \
        return run_ntt_t<F>(ctx, src, src_len, src_stride, dst, dst_stride, log_n, omega, batch, sc, s); }                                   \
    int field_op_##NAME(dehalo_ctx* ctx, int op, const fe* a, const fe* b, fe* out, uint64_t n, hipStream_t s) {                             \
        return field_op_t<F>(ctx, op, a, b, out, n, s); }
