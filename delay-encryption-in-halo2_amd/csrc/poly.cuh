// poly.cuh -- field-vector primitives either side of the MSM / NTT path:
//
//   eval_polynomial   sum_i c_i x^i                    [UPSTREAM halo2_proofs/src/arithmetic.rs
//                                                       eval_polynomial: chunked Horner]
//   batch_invert      v_i <- v_i^-1, zeros untouched   [UPSTREAM ff::BatchInvert, used by
//                                                       plonk/permutation/prover.rs and
//                                                       plonk/lookup/prover.rs]
//   prefix_product    z_0 = 1, z_i = prod_{j<i} f_j    [UPSTREAM the running product loops of
//   grand_product     f_j = num_j / den_j               permutation::Argument::commit and
//                                                       lookup::Permuted::commit_product]
//
// Same values as upstream, not its algorithms: the CPU versions are sequential recurrences
// chunked over rayon threads; here every step is a block-wide scan on the carry-free multiplier
// (fp29.cuh).  All of them move 32-B elements through HBM once or twice and are HBM- or
// latency-bound (a batch inversion cannot finish before ONE Fermat exponentiation, ~380 dependent
// multiplications, has run: ~0.13 ms on this clock whatever the size).
//
// Elements at the interface are upstream's 4 x u64 Montgomery form (x * 2^256); inside a kernel
// they are x * 2^261, lazily reduced (< 2p), limbs normalized.
#pragma once
#include "fp29.cuh"
#include "internal.hpp"

#define POLY_THREADS 256
#define POLY_EVAL_EPT 8
#define POLY_EVAL_TILE (POLY_THREADS * POLY_EVAL_EPT)
#define POLY_K 4                                 // elements per thread in the product kernels
#define POLY_PTILE (POLY_THREADS * POLY_K)

template <class F9> FP_DEV f29 poly_load(const fe* p) { return f29_from_std<F9>(f_load(p)); }
template <class F9> FP_DEV void poly_store(fe* p, const f29& v) { f_store(p, f29_to_std<F9>(v)); }
// internal value <-> 32-B canonical packed (scratch buffers between kernels: no conversion multiply)
template <class F9> FP_DEV void poly_store_packed(fe* p, const f29& v) { f_store(p, f29_to_packed_canon<F9>(v)); }
FP_DEV f29 poly_load_packed(const fe* p) { return f29_unpack(f_load(p)); }

// a^(p-2), a != 0 (mod p); binary square-and-multiply over the constant exponent
template <class F9>
FP_DEV f29 f29_inv(const f29& a) {
    typedef typename F9::Std F;
    // (measured: the operand-scanning schedule is 1.5x SLOWER here -- a lone wave pays per instruction,
    // and product scanning has fewer of them even counting its wait states)
    u32 e[8];
    u64 br = 2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 x = (u64)F::P[i] - br;
        e[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    f29 r = f29_one<F9>();
#pragma unroll
    for (int w = 7; w >= 0; w--) {
        const u32 word = e[w];
        for (int b = 31; b >= 0; b--) {
            r = f29_sqr<F9>(r);
            if ((word >> b) & 1) r = f29_mul<F9>(r, a);
        }
    }
    return r;
}

// Block-wide inclusive scans (prefix, and optionally suffix) of one value per thread under field
// multiplication; Hillis-Steele in LDS.  On return pre[t] = v_0 ... v_t, suf[t] = v_t ... v_255.
template <class F9, bool SUFFIX>
FP_DEV void block_scan_mul(f29* pre, f29* suf, const f29& v) {
    const u32 t = threadIdx.x;
    pre[t] = v;
    if (SUFFIX) suf[t] = v;
    __syncthreads();
    for (u32 d = 1; d < POLY_THREADS; d <<= 1) {
        f29 a = pre[t], b;
        if (t >= d) a = f29_mul<F9>(a, pre[t - d]);
        if (SUFFIX) {
            b = suf[t];
            if (t + d < POLY_THREADS) b = f29_mul<F9>(b, suf[t + d]);
        }
        __syncthreads();
        pre[t] = a;
        if (SUFFIX) suf[t] = b;
        __syncthreads();
    }
}

// ---- eval_polynomial ---------------------------------------------------------------------------
// One level of the evaluation tree: block b reduces coefficients [2048 b, 2048 b + 2048) of each
// polynomial to  sum_j c_j x^(j - 2048 b)  (Horner over 8 per thread, then a pairwise tree with
// x^8, x^16, ...), in standard form, so the next level is the same kernel on the partials with
// the point x^2048 (left in *next_point by block 0).
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_poly_eval(const fe* coeffs, u64 len, u64 stride, fe point_val, const fe* point_ptr, fe* partials,
                                                            u64 partial_stride, fe* next_point) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 sh[POLY_THREADS];
    const u32 t = threadIdx.x;
    const fe* c = coeffs + (u64)blockIdx.y * stride;
    const f29 x = f29_from_std<F9>(point_ptr ? f_load(point_ptr) : point_val);
    const u64 base = (u64)blockIdx.x * POLY_EVAL_TILE + (u64)t * POLY_EVAL_EPT;
    // coefficients stay plain integers of the standard form: (c 2^256)(x 2^261) 2^-261 = c x 2^256
    f29 acc = f29_zero();
#pragma unroll
    for (int j = POLY_EVAL_EPT - 1; j >= 0; j--) {
        f29 cj = base + j < len ? f29_unpack(f_load(&c[base + j])) : f29_zero();
        acc = f29_add(f29_mul<F9>(acc, x), cj);              // < 2p + p, limbs < 2^30
    }
    f29 X = f29_sqr<F9>(f29_sqr<F9>(f29_sqr<F9>(x)));       // x^8
    sh[t] = f29_norm(acc);
    __syncthreads();
    for (u32 d = 1; d < POLY_THREADS; d <<= 1) {
        if ((t & (2 * d - 1)) == 0) {
            f29 hi = f29_mul<F9>(sh[t + d], X);
            sh[t] = f29_norm(f29_add(sh[t], hi));            // grows by < 2p per level: < 3p + 8 * 2p = 19p << 2^261
        }
        X = f29_sqr<F9>(X);
        __syncthreads();
    }
    if (t == 0) {
        f29 r = f29_mul<F9>(sh[0], f29_one<F9>());           // < 2p
        f_store(&partials[(u64)blockIdx.y * partial_stride + blockIdx.x], f29_pack(f29_cond_sub(r, F9::P)));
        if (blockIdx.x == 0 && blockIdx.y == 0 && next_point) poly_store<F9>(next_point, X);   // x^2048
    }
}

// Up to POLY_MP_MAX points at once (the prover opens every column at x, omega x, omega^-1 x and omega^last x), with the block
// reduction done by ADDITIONS: a tiny kernel first tabulates pw[pt][t] = (x_pt^8)^t, t < 256 (and x_pt^2048 for the next level);
// a thread then runs the Horner chains of all points over its 8 coefficients together (independent chains), multiplies each
// by its table entry and the block sums the 256 terms in LDS -- 10 multiplications per 8 coefficients and point instead of the
// 8 + 16 of k_poly_eval's multiplicative tree, at the same occupancy.  partials[pt][poly][block].
#define POLY_MP_MAX 4
struct MultiPoints { fe pt[POLY_MP_MAX]; u32 count; };
// grid (points, levels): block (ip, l) tabulates level l's powers, whose point is x_ip^(2048^l); pw[l][ip][t], level_points[l][ip]
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_poly_pow_table(MultiPoints P, fe* pw, fe* level_points) {
    typedef typename f29_of<F>::type F9;
    const u32 t = threadIdx.x, ip = blockIdx.x, lvl = blockIdx.y;
    f29 x = f29_from_std<F9>(P.pt[ip]);
    for (u32 l = 0; l < lvl; l++)
        for (int q = 0; q < 11; q++) x = f29_sqr<F9>(x);                          // x^2048 per level
    if (t == 0) poly_store<F9>(&level_points[lvl * POLY_MP_MAX + ip], f29_norm(x));
    pw += (u64)lvl * POLY_MP_MAX * POLY_THREADS;
    f29 base = f29_sqr<F9>(f29_sqr<F9>(f29_sqr<F9>(x)));                        // x^8
    f29 r = f29_one<F9>();
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {                                           // (x^8)^t
        if ((t >> bit) & 1) r = f29_mul<F9>(r, base);
        base = f29_sqr<F9>(base);
    }
    f_store(&pw[ip * POLY_THREADS + t], f29_to_packed_canon<F9>(f29_norm(r)));   // internal form, canonical, packed
}

#define POLY_MP_POLYS 96
struct PolyPtrs { const fe* p[POLY_MP_POLYS]; unsigned char m[POLY_MP_POLYS]; };      // m: bit i = this polynomial is wanted at point i
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_poly_eval_multi(PolyPtrs polys, const fe* coeffs, u64 len, u64 stride, u64 point_stride, u32 npts, const fe* point_ptr,
                                                                  const fe* pw, fe* partials, u64 partial_stride, u64 partial_point_stride) {
    struct { u32 count; } P{npts};
    typedef typename f29_of<F>::type F9;
    __shared__ f29 sh[POLY_THREADS];
    const u32 t = threadIdx.x;
    const u64 base = (u64)blockIdx.x * POLY_EVAL_TILE + (u64)t * POLY_EVAL_EPT;
    f29 x[POLY_MP_MAX], acc[POLY_MP_MAX];
#pragma unroll
    for (int ip = 0; ip < POLY_MP_MAX; ip++) {
        x[ip] = (u32)ip < P.count ? f29_from_std<F9>(f_load(&point_ptr[ip])) : f29_zero();
        acc[ip] = f29_zero();
    }
    const fe* c0 = coeffs ? coeffs + (u64)blockIdx.y * stride : polys.p[blockIdx.y];       // level 0: one device pointer per polynomial
    const u32 want = polys.m[blockIdx.y];                                                   // (uniform in the block) points nobody asked for are skipped
    for (int j = POLY_EVAL_EPT - 1; j >= 0; j--) {      // (a rolled loop: eight rounds of up to four multiplications are past the unroller's size limit, and nothing is indexed by j)
        const bool in = base + j < len;
        // level >= 1: every point has its own partial sums (point_stride apart); level 0: one coefficient array for all points
        f29 cj = in && !point_stride ? f29_unpack(f_load(&c0[base + j])) : f29_zero();
#pragma unroll
        for (int ip = 0; ip < POLY_MP_MAX; ip++) {
            if (!((want >> ip) & 1)) continue;
            if (point_stride) cj = in ? f29_unpack(f_load(&c0[(u64)ip * point_stride + base + j])) : f29_zero();
            acc[ip] = f29_add(f29_mul<F9>(acc[ip], x[ip]), cj);
        }
    }
#pragma unroll
    for (int ip = 0; ip < POLY_MP_MAX; ip++) {
        if ((u32)ip >= P.count) break;
        fe* out = &partials[(u64)ip * partial_point_stride + (u64)blockIdx.y * partial_stride + blockIdx.x];
        if (!((want >> ip) & 1)) {
            if (t == 0) f_store(out, f_zero());
            continue;
        }
        f29 term = f29_mul<F9>(f29_norm(acc[ip]), f29_unpack(f_load(&pw[ip * POLY_THREADS + t])));     // < 2p
        __syncthreads();                                     // the previous point's sums have been read
        sh[t] = term;
        __syncthreads();
        for (u32 d = POLY_THREADS / 2; d >= 1; d >>= 1) {
            if (t < d) sh[t] = f29_norm(f29_add(sh[t], sh[t + d]));               // < 512 p << 2^261
            __syncthreads();
        }
        if (t == 0) {
            f29 r = f29_mul<F9>(sh[0], f29_one<F9>());
            f_store(out, f29_pack(f29_cond_sub(r, F9::P)));
        }
    }
}

// ---- batch inversion -----------------------------------------------------------------------------
// A block owns 1024 elements, four per thread (strided, so the loads coalesce -- Montgomery's
// trick does not care about order): thread products, block-wide prefix and suffix products in
// LDS, then  inv(T_t) = inv(block total) * prefix_excl(t) * suffix_excl(t)  and the usual
// back-substitution.  The block totals are themselves batch-inverted one level up (same kernels
// on the packed totals), so a call of any size runs exactly ONE Fermat exponentiation -- the
// ~0.2 ms latency floor of the whole operation -- in the single block at the top of the recursion:
//   k_bi_reduce (per level, going up)  ->  k_batch_invert<.., TOP> (1 block)  ->  k_batch_invert (going down).
// PACKED: elements are internal-form packed words (the totals), never zero.
template <class F, bool PACKED>
FP_DEV void bi_load(const fe* v, u64 len, u64 base, f29 (&a)[POLY_K], bool (&live)[POLY_K]) {
    typedef typename f29_of<F>::type F9;
#pragma unroll
    for (int j = 0; j < POLY_K; j++) {
        u64 i = base + (u64)j * POLY_THREADS;
        live[j] = false;
        a[j] = f29_one<F9>();
        if (i < len) {
            fe raw = f_load(&v[i]);
            if (PACKED) { live[j] = true; a[j] = f29_unpack(raw); }
            else if (!f_is_zero(raw)) { live[j] = true; a[j] = f29_from_std<F9>(raw); }
        }
    }
}

// totals[block] = product of the block's non-zero elements (packed internal form)
template <class F, bool PACKED>
__global__ __launch_bounds__(POLY_THREADS) void k_bi_reduce(const fe* v, u64 len, fe* totals) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 sh[POLY_THREADS];
    const u32 t = threadIdx.x;
    f29 a[POLY_K];
    bool live[POLY_K];
    bi_load<F, PACKED>(v, len, (u64)blockIdx.x * POLY_PTILE + t, a, live);
    sh[t] = f29_mul<F9>(f29_mul<F9>(a[0], a[1]), f29_mul<F9>(a[2], a[3]));
    __syncthreads();
    for (u32 d = POLY_THREADS / 2; d >= 1; d >>= 1) {
        if (t < d) sh[t] = f29_mul<F9>(sh[t], sh[t + d]);
        __syncthreads();
    }
    if (t == 0) poly_store_packed<F9>(&totals[blockIdx.x], sh[0]);
}

// TOP: the block inverts its own total (grid of one block); otherwise inv_totals[block] holds it.
template <class F, bool PACKED, bool TOP>
__global__ __launch_bounds__(POLY_THREADS) void k_batch_invert(fe* v, u64 len, const fe* inv_totals) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 pre[POLY_THREADS], suf[POLY_THREADS];
    __shared__ f29 inv_total;
    const u32 t = threadIdx.x;
    const u64 base = (u64)blockIdx.x * POLY_PTILE + t;
    f29 a[POLY_K];
    bool live[POLY_K];
    bi_load<F, PACKED>(v, len, base, a, live);
    const f29 p2 = f29_mul<F9>(a[0], a[1]);
    const f29 p3 = f29_mul<F9>(p2, a[2]);
    const f29 T = f29_mul<F9>(p3, a[3]);
    block_scan_mul<F9, true>(pre, suf, T);
    f29 u;
    if (TOP) {
        if (t < 64) {                                        // one wave, every lane the same value
            f29 inv = f29_inv_safegcd<F9>(pre[POLY_THREADS - 1]);
            if (t == 0) inv_total = inv;
        }
        __syncthreads();
        u = inv_total;
    } else {
        u = poly_load_packed(&inv_totals[blockIdx.x]);
    }
    if (t > 0) u = f29_mul<F9>(u, pre[t - 1]);
    if (t + 1 < POLY_THREADS) u = f29_mul<F9>(u, suf[t + 1]);
    // u = 1 / (a0 a1 a2 a3)
    f29 r[POLY_K];
    r[3] = f29_mul<F9>(u, p3); u = f29_mul<F9>(u, a[3]);
    r[2] = f29_mul<F9>(u, p2); u = f29_mul<F9>(u, a[2]);
    r[1] = f29_mul<F9>(u, a[0]); u = f29_mul<F9>(u, a[1]);
    r[0] = u;
#pragma unroll
    for (int j = 0; j < POLY_K; j++) {
        u64 i = base + (u64)j * POLY_THREADS;
        if (i < len && live[j]) {
            if (PACKED) poly_store_packed<F9>(&v[i], r[j]);
            else poly_store<F9>(&v[i], r[j]);
        }
    }
}

// ---- exclusive prefix product ------------------------------------------------------------------
// pass 1: thread t of block b owns elements 1024 b + 4 t .. + 3; writes the product of everything
// before them inside the block (scratch_e, packed internal form) and the block total.
// If `mul_by` is given the scanned sequence is in[i] * mul_by[i] (grand product: den^-1 * num).
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_pp_block(const fe* in, u64 in_stride, const fe* mul_by, u64 mul_stride, u64 len, fe* scratch_e, fe* totals) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 pre[POLY_THREADS];
    const u32 t = threadIdx.x;
    in += (u64)blockIdx.y * in_stride;                       // blockIdx.y: independent column of a batch
    if (mul_by) mul_by += (u64)blockIdx.y * mul_stride;
    scratch_e += (u64)blockIdx.y * gridDim.x * POLY_THREADS;
    totals += (u64)blockIdx.y * gridDim.x;
    const u64 base = (u64)blockIdx.x * POLY_PTILE + (u64)t * POLY_K;
    f29 T = f29_one<F9>();
    for (int j = 0; j < POLY_K; j++) {      // (rolled: see k_poly_eval_multi)
        if (base + j < len) {
            f29 a = poly_load<F9>(&in[base + j]);
            if (mul_by) a = f29_mul<F9>(a, poly_load<F9>(&mul_by[base + j]));
            T = j == 0 ? a : f29_mul<F9>(T, a);
        }
    }
    block_scan_mul<F9, false>(pre, nullptr, T);
    f29 e = t > 0 ? pre[t - 1] : f29_one<F9>();
    poly_store_packed<F9>(&scratch_e[(u64)blockIdx.x * POLY_THREADS + t], e);
    if (t == POLY_THREADS - 1) poly_store_packed<F9>(&totals[blockIdx.x], pre[t]);
}

// pass 2 (one block): exclusive scan of the block totals, 256 at a time with a running carry
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_pp_top(const fe* totals, u64 nblocks, fe* bprefix) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 pre[POLY_THREADS];
    const u32 t = threadIdx.x;
    totals += (u64)blockIdx.x * nblocks;                     // one block per column
    bprefix += (u64)blockIdx.x * nblocks;
    f29 carry = f29_one<F9>();
    for (u64 c0 = 0; c0 < nblocks; c0 += POLY_THREADS) {
        f29 v = c0 + t < nblocks ? poly_load_packed(&totals[c0 + t]) : f29_one<F9>();
        block_scan_mul<F9, false>(pre, nullptr, v);
        f29 e = t > 0 ? f29_mul<F9>(carry, pre[t - 1]) : carry;
        if (c0 + t < nblocks) poly_store_packed<F9>(&bprefix[c0 + t], e);
        carry = f29_mul<F9>(carry, pre[POLY_THREADS - 1]);
        __syncthreads();
    }
}

// pass 3: out[i] = bprefix[b] * e[b][t] * in[4t] ... in[i-1]
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_pp_apply(const fe* in, u64 in_stride, const fe* mul_by, u64 mul_stride, u64 len, const fe* scratch_e,
                                                           const fe* bprefix, fe* out, u64 out_stride) {
    typedef typename f29_of<F>::type F9;
    const u32 t = threadIdx.x;
    in += (u64)blockIdx.y * in_stride;
    if (mul_by) mul_by += (u64)blockIdx.y * mul_stride;
    out += (u64)blockIdx.y * out_stride;
    scratch_e += (u64)blockIdx.y * gridDim.x * POLY_THREADS;
    bprefix += (u64)blockIdx.y * gridDim.x;
    const u64 base = (u64)blockIdx.x * POLY_PTILE + (u64)t * POLY_K;
    if (base >= len) return;
    // Element by element -- load a_j, store the running product, multiply -- so that no register array is indexed by a loop the unroller refuses (the
    // four-element array of the first version lived in 160 bytes of scratch per lane); a thread owns its POLY_K elements, so `out` may be `in`.
    f29 run = f29_mul<F9>(poly_load_packed(&bprefix[blockIdx.x]), poly_load_packed(&scratch_e[(u64)blockIdx.x * POLY_THREADS + t]));
    for (int j = 0; j < POLY_K; j++) {
        if (base + j >= len) break;
        f29 a = f29_one<F9>();
        if (j + 1 < POLY_K) {
            a = poly_load<F9>(&in[base + j]);
            if (mul_by) a = f29_mul<F9>(a, poly_load<F9>(&mul_by[base + j]));
        }
        poly_store<F9>(&out[base + j], run);
        if (j + 1 < POLY_K) run = f29_mul<F9>(run, a);
    }
}


// ---- linear combination of columns ---------------------------------------------------------------
// out[i] = sum_j coef[j] * cols[j][i]  (+ out[i] if accumulate)  -  sub0 at i = 0.
// [UPSTREAM plonk/prover.rs + poly/kzg/multiopen/gwc/prover.rs: "poly_batch = sum_i v^i poly_i; poly_batch -=
//  eval_batch" before kate_division, and vanishing::Constructed::evaluate's fold of the h pieces with x^n].
// Pointers and coefficients travel as kernel arguments; the block converts the coefficients to the internal
// form once (LDS), columns stay plain integers of their standard form: (c 2^261)(v 2^256) 2^-261 = c v 2^256.
#define POLY_LC_MAX 40
struct LincombArgs {
    const fe* cols[POLY_LC_MAX];
    fe coef[POLY_LC_MAX];
    u32 count;
};
// EPT elements per thread: 4 for long vectors (independent loads and multiplications per column), 1 when that would leave the chip
// a fraction of a wave per SIMD (2^17 coefficients at 4 per thread are 512 waves for 1024 SIMDs)
template <class F, int POLY_LC_EPT>
__global__ __launch_bounds__(POLY_THREADS) void k_lincomb(LincombArgs A, u64 len, fe* out, int accumulate, fe sub0, int has_sub0) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 cf[POLY_LC_MAX];
    const u32 t = threadIdx.x;
    if (t < A.count) cf[t] = f29_from_std<F9>(A.coef[t]);
    __syncthreads();
    const u64 i0 = (u64)blockIdx.x * (POLY_THREADS * POLY_LC_EPT) + t;
    // the thread's POLY_LC_EPT elements advance together: that many independent loads and multiplications per column
    f29 acc[POLY_LC_EPT];
#pragma unroll
    for (int e = 0; e < POLY_LC_EPT; e++) {
        const u64 i = i0 + (u64)e * POLY_THREADS;
        acc[e] = accumulate && i < len ? f29_unpack(f_load(&out[i])) : f29_zero();
    }
    for (u32 j = 0; j < A.count; j++) {
        const fe* col = A.cols[j];
        const f29 cj = cf[j];
#pragma unroll
        for (int e = 0; e < POLY_LC_EPT; e++) {
            const u64 i = i0 + (u64)e * POLY_THREADS;
            if (i < len) acc[e] = f29_norm(f29_add(acc[e], f29_mul<F9>(cj, f29_unpack(f_load(&col[i])))));     // < (1 + 2 * 40) p  << 2^261
        }
    }
#pragma unroll
    for (int e = 0; e < POLY_LC_EPT; e++) {
        const u64 i = i0 + (u64)e * POLY_THREADS;
        if (i >= len) continue;
        f29 a = acc[e];
        if (has_sub0 && i == 0) a = f29_norm(f29_sub(a, f29_unpack(sub0), F9::KM));
        f29 r = f29_mul<F9>(a, f29_one<F9>());                                    // value * 2^261 * 2^256 / 2^261: standard form, < 2p
        f_store(&out[i], f29_pack(f29_cond_sub(r, F9::P)));
    }
}

// ---- element-wise scaling --------------------------------------------------------------------------
// a[i] *= pattern[i mod period] (period a power of two <= 8; vanishing-polynomial division on the extended
// domain: t(X)^-1 takes 2^(extended_k - k) values [UPSTREAM poly/domain.rs divide_by_vanishing_poly]) and / or
// *= *factor, one element read from device memory (the running value carried from one permutation product
// column into the next [UPSTREAM plonk/permutation/prover.rs: "z = vec![last_z]"]).
struct ScaleArgs { fe pattern[8]; u32 period; };
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_scale(fe* a, u64 len, ScaleArgs S, const fe* d_factor) {
    typedef typename f29_of<F>::type F9;
    const u64 i = (u64)blockIdx.x * POLY_THREADS + threadIdx.x;
    if (i >= len) return;
    f29 v = f29_unpack(f_load(&a[i]));                                            // plain integer of the standard form
    if (S.period) v = f29_mul<F9>(v, f29_from_std<F9>(S.pattern[i & (S.period - 1)]));
    if (d_factor) v = f29_mul<F9>(v, f29_from_std<F9>(f_load(d_factor)));
    if (!S.period && !d_factor) return;
    f_store(&a[i], f29_pack(f29_cond_sub(v, F9::P)));                            // a product is < 2p; the element keeps its form (the map is linear)
}

// ---- kate_division -----------------------------------------------------------------------------------
// q = (a(X) - a(z)) / (X - z):  q[i] = a[i+1] + z q[i+1], q has len - 1 coefficients
// [UPSTREAM halo2_proofs/src/arithmetic.rs kate_division: a sequential recurrence from the top coefficient].
// Here E(i) = sum_{j >= i} a[j] z^(j-i) is a suffix scan and q[i-1] = E(i):
//   (1) k_poly_eval gives every 2048-block's S_b = sum_j a[2048 b + j] z^j and z^2048;
//   (2) the carries C_b = E(2048 (b+1)) are the kate division of the S sequence by z^2048 (recursion);
//   (3) k_kate_apply: a thread's 8 coefficients, a 257-slot Hillis-Steele suffix scan in LDS (slot 256 = C_b),
//       then Horner down the thread's coefficients, storing E(i) to q[i-1].
template <class F>
FP_DEV void kate_apply_body(const fe* a, u64 len, const fe& point, const fe* carries, fe* q) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 sh[POLY_THREADS + 1];
    const u32 t = threadIdx.x;
    const f29 x = f29_from_std<F9>(point);
    const u64 base = (u64)blockIdx.x * POLY_EVAL_TILE + (u64)t * POLY_EVAL_EPT;
    f29 c[POLY_EVAL_EPT];
    f29 acc = f29_zero();
#pragma unroll
    for (int j = POLY_EVAL_EPT - 1; j >= 0; j--) {
        c[j] = base + j < len ? f29_unpack(f_load(&a[base + j])) : f29_zero();
        acc = f29_norm(f29_add(f29_mul<F9>(acc, x), c[j]));                      // values carry the factor 2^256 (standard form as integers)
    }
    sh[t] = acc;
    if (t == 0) sh[POLY_THREADS] = carries ? f29_unpack(f_load(&carries[blockIdx.x])) : f29_zero();
    f29 X = f29_sqr<F9>(f29_sqr<F9>(f29_sqr<F9>(x)));                            // x^8
    __syncthreads();
    for (u32 d = 1; d <= POLY_THREADS; d <<= 1) {
        f29 v = sh[t];
        if (t + d <= POLY_THREADS) v = f29_norm(f29_add(v, f29_mul<F9>(sh[t + d], X)));   // grows by < 2p per level
        __syncthreads();
        sh[t] = v;
        X = f29_sqr<F9>(X);
        __syncthreads();
    }
    // E at the end of this thread's coefficients
    f29 e = sh[t + 1];
#pragma unroll
    for (int j = POLY_EVAL_EPT - 1; j >= 0; j--) {
        e = f29_norm(f29_add(f29_mul<F9>(e, x), c[j]));                            // E(base + j)
        const u64 i = base + j;
        if (i >= 1 && i < len) {
            f29 r = f29_mul<F9>(e, f29_one<F9>());
            f_store(&q[i - 1], f29_pack(f29_cond_sub(r, F9::P)));
        }
    }
}
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_kate_apply(const fe* a, u64 len, fe point_val, const fe* point_ptr, const fe* carries, fe* q) {
    kate_apply_body<F>(a, len, point_ptr ? f_load(point_ptr) : point_val, carries, q);
}
// Several (polynomial, point) pairs in one launch (blockIdx.y): ProverGWC opens at up to four points, and every launch of
// this latency-bound chain costs the same whether it carries one division or four.
#define POLY_KATE_MAX 8
struct KateBatch { const fe* a[POLY_KATE_MAX]; fe* q[POLY_KATE_MAX]; fe pt[POLY_KATE_MAX]; };
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_kate_apply_batch(KateBatch B, u64 len, const fe* pt_dev, const fe* carries, u64 carries_stride) {
    const u32 y = blockIdx.y;
    kate_apply_body<F>(B.a[y], len, pt_dev ? f_load(&pt_dev[y]) : B.pt[y], carries ? carries + (u64)y * carries_stride : nullptr, B.q[y]);
}
// block sums S[y][b] = sum_j a_y[2048 b + j] z_y^j and z_y^2048 (the k_poly_eval tree with a point per polynomial)
template <class F>
__global__ __launch_bounds__(POLY_THREADS) void k_kate_sums_batch(KateBatch B, u64 len, fe* S, u64 s_stride, fe* next_pt) {
    typedef typename f29_of<F>::type F9;
    __shared__ f29 sh[POLY_THREADS];
    const u32 t = threadIdx.x, y = blockIdx.y;
    const fe* c = B.a[y];
    const f29 x = f29_from_std<F9>(B.pt[y]);
    const u64 base = (u64)blockIdx.x * POLY_EVAL_TILE + (u64)t * POLY_EVAL_EPT;
    f29 acc = f29_zero();
#pragma unroll
    for (int j = POLY_EVAL_EPT - 1; j >= 0; j--) {
        f29 cj = base + j < len ? f29_unpack(f_load(&c[base + j])) : f29_zero();
        acc = f29_add(f29_mul<F9>(acc, x), cj);
    }
    f29 X = f29_sqr<F9>(f29_sqr<F9>(f29_sqr<F9>(x)));
    sh[t] = f29_norm(acc);
    __syncthreads();
    for (u32 d = 1; d < POLY_THREADS; d <<= 1) {
        if ((t & (2 * d - 1)) == 0) sh[t] = f29_norm(f29_add(sh[t], f29_mul<F9>(sh[t + d], X)));
        X = f29_sqr<F9>(X);
        __syncthreads();
    }
    if (t == 0) {
        f29 r = f29_mul<F9>(sh[0], f29_one<F9>());
        f_store(&S[(u64)y * s_stride + blockIdx.x], f29_pack(f29_cond_sub(r, F9::P)));
        if (blockIdx.x == 0) poly_store<F9>(&next_pt[y], X);
    }
}

// ==========================================================================================
// host drivers (instantiated once per field next to the NTT in ntt_<field>.hip)
// ==========================================================================================
template <class F>
int eval_poly_t(dehalo_ctx* ctx, const fe* d_coeffs, uint64_t len, uint64_t stride, size_t batch, const uint64_t point[4], fe* d_out, hipStream_t s) {
    if (batch == 0) return 0;
    if (len == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, batch * sizeof(fe), s));
        return 0;
    }
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    const uint64_t nb0 = (len + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
    TRY(dh_ensure(ctx, ctx->ws_poly[0], std::max<size_t>(64, batch * nb0 * sizeof(fe))));
    TRY(dh_ensure(ctx, ctx->ws_poly[1], std::max<size_t>(64, batch * ((nb0 + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE) * sizeof(fe))));
    TRY(dh_ensure(ctx, ctx->ws_poly[2], 4 * sizeof(fe)));     // the level points x^(2048^l)
    fe* bufs[2] = {(fe*)ctx->ws_poly[0].p, (fe*)ctx->ws_poly[1].p};
    fe* pts = (fe*)ctx->ws_poly[2].p;
    const fe* cur = d_coeffs;
    uint64_t cur_len = len, cur_stride = stride;
    const fe x0 = fe_from_u64(point);
    const fe* pt_ptr = nullptr;
    int level = 0;
    for (;;) {
        uint64_t nb = (cur_len + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
        fe* dst = nb == 1 ? d_out : bufs[level & 1];
        fe* next_pt = nb == 1 ? nullptr : &pts[level & 3];
        k_poly_eval<F><<<dim3((u32)nb, (u32)batch), POLY_THREADS, 0, s>>>(cur, cur_len, cur_stride, x0, pt_ptr, dst, nb, next_pt);
        HIP_TRY(ctx, hipGetLastError());
        if (nb == 1) break;
        cur = dst; cur_len = nb; cur_stride = nb; pt_ptr = next_pt;
        level++;
    }
    return 0;
}

// out[pt][poly] = poly_j(points[pt]) for `count` polynomials given by device pointers (len coefficients each) and npts <= POLY_MP_MAX
// points: one table kernel for every level, then one pass per level
template <class F>
int eval_poly_multi_t(dehalo_ctx* ctx, const fe* const* d_polys, size_t count, uint64_t len, const uint64_t* points, uint32_t npts, fe* d_out, hipStream_t s,
                      const uint8_t* masks) {
    if (count == 0 || npts == 0) return 0;
    if (len == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, (size_t)npts * count * sizeof(fe), s));
        return 0;
    }
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    uint32_t levels = 1;
    for (uint64_t l = len; l > POLY_EVAL_TILE; l = (l + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE) levels++;
    if (levels > 4) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial_multi: polynomial too long");
    const uint64_t nb0 = (len + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
    const uint64_t nb1 = (nb0 + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
    TRY(dh_ensure(ctx, ctx->ws_poly[0], std::max<size_t>(64, (size_t)npts * count * nb0 * sizeof(fe))));
    TRY(dh_ensure(ctx, ctx->ws_poly[1], std::max<size_t>(64, (size_t)npts * count * nb1 * sizeof(fe))));
    TRY(dh_ensure(ctx, ctx->ws_poly[2], (size_t)4 * (POLY_MP_MAX + POLY_MP_MAX * POLY_THREADS) * sizeof(fe)));
    fe* bufs[2] = {(fe*)ctx->ws_poly[0].p, (fe*)ctx->ws_poly[1].p};
    fe* lvl_pts = (fe*)ctx->ws_poly[2].p;                     // [level][point]
    fe* pw = lvl_pts + 4 * POLY_MP_MAX;                       // [level][point][256]
    MultiPoints P{};
    P.count = npts;
    for (uint32_t i = 0; i < npts; i++) P.pt[i] = fe_from_u64(points + 4 * i);
    k_poly_pow_table<F><<<dim3(npts, levels), POLY_THREADS, 0, s>>>(P, pw, lvl_pts);
    uint64_t cur_len = len;
    const fe* cur = nullptr;
    for (uint32_t level = 0; level < levels; level++) {
        const uint64_t nb = (cur_len + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
        const bool last = level + 1 == levels;
        fe* dst = last ? d_out : bufs[level & 1];
        const uint64_t dst_pstride = last ? count : count * nb;       // d_out is [pt][poly]; scratch is [pt][poly][block]
        const fe* lp = lvl_pts + level * POLY_MP_MAX;
        const fe* lpw = pw + (uint64_t)level * POLY_MP_MAX * POLY_THREADS;
        // (masks: which points each polynomial is wanted at -- a proof asks for 60 of its 47 x 4 values; the others are written as zero)
        for (size_t first = 0; first < count; first += POLY_MP_POLYS) {
            PolyPtrs pp{};
            const size_t cnt = std::min<size_t>(POLY_MP_POLYS, count - first);
            for (size_t j = 0; j < cnt; j++) pp.m[j] = masks ? masks[first + j] : 0xf;
            if (level == 0) {
                for (size_t j = 0; j < cnt; j++) pp.p[j] = d_polys[first + j];
                k_poly_eval_multi<F><<<dim3((u32)nb, (u32)cnt), POLY_THREADS, 0, s>>>(pp, nullptr, cur_len, 0, 0, npts, lp, lpw, dst + first * (last ? 1 : nb), last ? 1 : nb, dst_pstride);
            } else {
                k_poly_eval_multi<F><<<dim3((u32)nb, (u32)cnt), POLY_THREADS, 0, s>>>(pp, cur + first * cur_len, cur_len, cur_len, count * cur_len, npts, lp, lpw, dst + first * (last ? 1 : nb),
                                                                                          last ? 1 : nb, dst_pstride);
            }
        }
        HIP_TRY(ctx, hipGetLastError());
        cur = dst; cur_len = nb;
    }
    return 0;
}

template <class F, bool PACKED>
int batch_invert_level(dehalo_ctx* ctx, fe* d_v, uint64_t len, fe* scratch, hipStream_t s) {
    const uint64_t nb = (len + POLY_PTILE - 1) / POLY_PTILE;
    if (nb == 1) {
        k_batch_invert<F, PACKED, true><<<1, POLY_THREADS, 0, s>>>(d_v, len, nullptr);
        return 0;
    }
    k_bi_reduce<F, PACKED><<<(u32)nb, POLY_THREADS, 0, s>>>(d_v, len, scratch);
    TRY((batch_invert_level<F, true>(ctx, scratch, nb, scratch + nb, s)));
    k_batch_invert<F, PACKED, false><<<(u32)nb, POLY_THREADS, 0, s>>>(d_v, len, scratch);
    return 0;
}

template <class F>
int batch_invert_t(dehalo_ctx* ctx, fe* d_v, uint64_t len, hipStream_t s) {
    if (len == 0) return 0;
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    const uint64_t nb = (len + POLY_PTILE - 1) / POLY_PTILE;
    TRY(dh_ensure(ctx, ctx->ws_poly[4], (nb + nb / 512 + 8) * sizeof(fe)));     // totals of every level
    TRY((batch_invert_level<F, false>(ctx, d_v, len, (fe*)ctx->ws_poly[4].p, s)));
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// out[i] = prod_{j < i} in[j] (* mul_by[j]) for each of `batch` columns;  in == out allowed
template <class F>
int prefix_product_t(dehalo_ctx* ctx, const fe* d_in, uint64_t in_stride, const fe* d_mul_by, uint64_t mul_stride, uint64_t len, size_t batch, fe* d_out,
                     uint64_t out_stride, hipStream_t s) {
    if (len == 0 || batch == 0) return 0;
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    const uint64_t nb = (len + POLY_PTILE - 1) / POLY_PTILE;
    TRY(dh_ensure(ctx, ctx->ws_poly[0], batch * nb * POLY_THREADS * sizeof(fe)));
    TRY(dh_ensure(ctx, ctx->ws_poly[1], batch * nb * sizeof(fe)));
    TRY(dh_ensure(ctx, ctx->ws_poly[2], batch * nb * sizeof(fe)));
    fe* e = (fe*)ctx->ws_poly[0].p;
    fe* totals = (fe*)ctx->ws_poly[1].p;
    fe* bprefix = (fe*)ctx->ws_poly[2].p;
    const dim3 grid((u32)nb, (u32)batch);
    k_pp_block<F><<<grid, POLY_THREADS, 0, s>>>(d_in, in_stride, d_mul_by, mul_stride, len, e, totals);
    k_pp_top<F><<<(u32)batch, POLY_THREADS, 0, s>>>(totals, nb, bprefix);
    k_pp_apply<F><<<grid, POLY_THREADS, 0, s>>>(d_in, in_stride, d_mul_by, mul_stride, len, e, bprefix, d_out, out_stride);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// z[i] = prod_{j < i} num[j] / den[j] for each of `batch` columns (zero denominators are treated like
// upstream's batch_invert treats them: left as zero, so the product collapses to zero from there on).
// All denominators of the batch share ONE batch inversion (one Fermat exponentiation).
template <class F>
int grand_product_t(dehalo_ctx* ctx, const fe* d_num, const fe* d_den, uint64_t len, size_t batch, uint64_t stride, fe* d_z, hipStream_t s) {
    if (len == 0 || batch == 0) return 0;
    TRY(dh_ensure(ctx, ctx->ws_poly[3], batch * len * sizeof(fe)));
    fe* inv = (fe*)ctx->ws_poly[3].p;
    HIP_TRY(ctx, hipMemcpy2DAsync(inv, len * sizeof(fe), d_den, stride * sizeof(fe), len * sizeof(fe), batch, hipMemcpyDeviceToDevice, s));
    TRY(batch_invert_t<F>(ctx, inv, batch * len, s));
    return prefix_product_t<F>(ctx, inv, len, d_num, stride, len, batch, d_z, stride, s);
}


template <class F>
int lincomb_t(dehalo_ctx* ctx, const fe* const* d_cols, const uint64_t* coefs, size_t count, uint64_t len, fe* d_out, const uint64_t* sub0, hipStream_t s) {
    if (len == 0) return 0;
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    const u32 ept = len >= (1ull << 20) ? 4 : 1;
    const u32 grid = (u32)((len + POLY_THREADS * ept - 1) / (POLY_THREADS * ept));
    if (count == 0) HIP_TRY(ctx, hipMemsetAsync(d_out, 0, len * sizeof(fe), s));
    fe s0 = sub0 ? fe_from_u64(sub0) : fe{};
    for (size_t first = 0; first < count || (first == 0 && sub0); first += POLY_LC_MAX) {
        LincombArgs A;
        A.count = (u32)std::min<size_t>(POLY_LC_MAX, count - first);
        for (u32 j = 0; j < A.count; j++) { A.cols[j] = d_cols[first + j]; A.coef[j] = fe_from_u64(coefs + 4 * (first + j)); }
        const bool last = first + POLY_LC_MAX >= count;
        if (ept == 4) k_lincomb<F, 4><<<grid, POLY_THREADS, 0, s>>>(A, len, d_out, first != 0 || count == 0, s0, sub0 && last ? 1 : 0);
        else k_lincomb<F, 1><<<grid, POLY_THREADS, 0, s>>>(A, len, d_out, first != 0 || count == 0, s0, sub0 && last ? 1 : 0);
        if (count == 0) break;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class F>
int scale_t(dehalo_ctx* ctx, fe* d_a, uint64_t len, const uint64_t* pattern, uint32_t period, const fe* d_factor, hipStream_t s) {
    if (len == 0 || (!period && !d_factor)) return 0;
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    ScaleArgs S{};
    S.period = period;
    for (uint32_t i = 0; i < period; i++) S.pattern[i] = fe_from_u64(pattern + 4 * i);
    k_scale<F><<<(u32)((len + POLY_THREADS - 1) / POLY_THREADS), POLY_THREADS, 0, s>>>(d_a, len, S, d_factor);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// level of the kate recursion: a (len coefficients, standard form) / (X - point) -> q (len - 1); scratch holds the
// block sums and carries of every level: [S: nb][C: nb][point: 1] then the next level
template <class F>
int kate_level(dehalo_ctx* ctx, const fe* d_a, uint64_t len, fe point_val, const fe* point_ptr, fe* d_q, fe* scratch, hipStream_t s) {
    const uint64_t nb = (len + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
    if (nb == 1) {
        k_kate_apply<F><<<1, POLY_THREADS, 0, s>>>(d_a, len, point_val, point_ptr, nullptr, d_q);
        return 0;
    }
    fe* S = scratch;
    fe* C = scratch + nb;
    fe* next_pt = scratch + 2 * nb;
    k_poly_eval<F><<<dim3((u32)nb, 1), POLY_THREADS, 0, s>>>(d_a, len, len, point_val, point_ptr, S, nb, next_pt);
    HIP_TRY(ctx, hipMemsetAsync(C, 0, nb * sizeof(fe), s));                        // C[nb - 1] = 0: nothing above the last block
    TRY((kate_level<F>(ctx, S, nb, fe{}, next_pt, C, scratch + 2 * nb + 8, s)));   // C[b] = E_S(b + 1), b < nb - 1
    k_kate_apply<F><<<(u32)nb, POLY_THREADS, 0, s>>>(d_a, len, point_val, point_ptr, C, d_q);
    return 0;
}

// `count` <= POLY_KATE_MAX divisions of equal length in three launches (sums, carries, apply); len <= 2048^2
template <class F>
int kate_division_batch_t(dehalo_ctx* ctx, const fe* const* d_a, uint64_t len, const uint64_t* points, fe* const* d_q, size_t count, hipStream_t s) {
    if (len <= 1 || count == 0) return 0;
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    const uint64_t nb = (len + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
    KateBatch B{};
    for (size_t y = 0; y < count; y++) { B.a[y] = d_a[y]; B.q[y] = d_q[y]; B.pt[y] = fe_from_u64(points + 4 * y); }
    if (nb == 1) {
        k_kate_apply_batch<F><<<dim3(1, (u32)count), POLY_THREADS, 0, s>>>(B, len, nullptr, nullptr, 0);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    TRY(dh_ensure(ctx, ctx->ws_poly[0], (2 * nb + 2) * POLY_KATE_MAX * sizeof(fe) + 4096));
    fe* S = (fe*)ctx->ws_poly[0].p;                       // [y][nb]
    fe* C = S + POLY_KATE_MAX * nb;                       // [y][nb]; C[y][nb - 1] = 0: nothing above the last block
    fe* pt2 = C + POLY_KATE_MAX * nb;                     // [y]: z_y^2048
    k_kate_sums_batch<F><<<dim3((u32)nb, (u32)count), POLY_THREADS, 0, s>>>(B, len, S, nb, pt2);
    HIP_TRY(ctx, hipMemsetAsync(C, 0, POLY_KATE_MAX * nb * sizeof(fe), s));
    KateBatch B2{};
    for (size_t y = 0; y < count; y++) { B2.a[y] = S + y * nb; B2.q[y] = C + y * nb; }
    k_kate_apply_batch<F><<<dim3(1, (u32)count), POLY_THREADS, 0, s>>>(B2, nb, pt2, nullptr, 0);       // C[y][b] = E_S(b + 1), b < nb - 1
    k_kate_apply_batch<F><<<dim3((u32)nb, (u32)count), POLY_THREADS, 0, s>>>(B, len, nullptr, C, nb);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class F>
int kate_division_t(dehalo_ctx* ctx, const fe* d_a, uint64_t len, const uint64_t point[4], fe* d_q, hipStream_t s) {
    if (len <= 1) return 0;
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    const uint64_t nb = (len + POLY_EVAL_TILE - 1) / POLY_EVAL_TILE;
    TRY(dh_ensure(ctx, ctx->ws_poly[0], (2 * nb + 8) * 2 * sizeof(fe) + 4096));
    TRY((kate_level<F>(ctx, d_a, len, fe_from_u64(point), nullptr, d_q, (fe*)ctx->ws_poly[0].p, s)));
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

#define DEFINE_POLY_ENTRY(NAME, F)                                                                                                              \
    int eval_poly_##NAME(dehalo_ctx* ctx, const fe* c, uint64_t len, uint64_t stride, size_t batch, const uint64_t pt[4], fe* out, hipStream_t s) { \
        return eval_poly_t<F>(ctx, c, len, stride, batch, pt, out, s); }                                                                         \
    int eval_poly_multi_##NAME(dehalo_ctx* ctx, const fe* const* polys, size_t count, uint64_t len, const uint64_t* pts, uint32_t npts, fe* out, hipStream_t s, \
                               const uint8_t* masks) {                                                                                                             \
        return eval_poly_multi_t<F>(ctx, polys, count, len, pts, npts, out, s, masks); }                                                     \
    int batch_invert_##NAME(dehalo_ctx* ctx, fe* v, uint64_t len, hipStream_t s) { return batch_invert_t<F>(ctx, v, len, s); }                    \
    int prefix_product_##NAME(dehalo_ctx* ctx, const fe* in, uint64_t len, fe* out, hipStream_t s) {                                             \
        return prefix_product_t<F>(ctx, in, len, nullptr, 0, len, 1, out, len, s); }                                                             \
    int grand_product_##NAME(dehalo_ctx* ctx, const fe* num, const fe* den, uint64_t len, size_t batch, uint64_t stride, fe* z, hipStream_t s) { \
        return grand_product_t<F>(ctx, num, den, len, batch, stride, z, s); }                                                                   \
    int lincomb_##NAME(dehalo_ctx* ctx, const fe* const* cols, const uint64_t* coefs, size_t count, uint64_t len, fe* out, const uint64_t* sub0, hipStream_t s) { \
        return lincomb_t<F>(ctx, cols, coefs, count, len, out, sub0, s); }                                                                       \
    int scale_##NAME(dehalo_ctx* ctx, fe* a, uint64_t len, const uint64_t* pattern, uint32_t period, const fe* d_factor, hipStream_t s) {        \
        return scale_t<F>(ctx, a, len, pattern, period, d_factor, s); }                                                                          \
    int kate_division_##NAME(dehalo_ctx* ctx, const fe* a, uint64_t len, const uint64_t pt[4], fe* q, hipStream_t s) {                           \
        return kate_division_t<F>(ctx, a, len, pt, q, s); }                                                                                      \
    int kate_division_batch_##NAME(dehalo_ctx* ctx, const fe* const* a, uint64_t len, const uint64_t* pts, fe* const* q, size_t count, hipStream_t s) { \
        return kate_division_batch_t<F>(ctx, a, len, pts, q, count, s); }
