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
#pragma unroll
    for (int j = 0; j < POLY_K; j++) {
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
    f29 a[POLY_K];
#pragma unroll
    for (int j = 0; j < POLY_K; j++) {
        a[j] = f29_one<F9>();
        if (base + j < len) {
            a[j] = poly_load<F9>(&in[base + j]);
            if (mul_by) a[j] = f29_mul<F9>(a[j], poly_load<F9>(&mul_by[base + j]));
        }
    }
    f29 run = f29_mul<F9>(poly_load_packed(&bprefix[blockIdx.x]), poly_load_packed(&scratch_e[(u64)blockIdx.x * POLY_THREADS + t]));
#pragma unroll
    for (int j = 0; j < POLY_K; j++) {
        if (base + j < len) {
            poly_store<F9>(&out[base + j], run);
            if (j + 1 < POLY_K) run = f29_mul<F9>(run, a[j]);
        }
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

#define DEFINE_POLY_ENTRY(NAME, F)                                                                                                              \
    int eval_poly_##NAME(dehalo_ctx* ctx, const fe* c, uint64_t len, uint64_t stride, size_t batch, const uint64_t pt[4], fe* out, hipStream_t s) { \
        return eval_poly_t<F>(ctx, c, len, stride, batch, pt, out, s); }                                                                         \
    int batch_invert_##NAME(dehalo_ctx* ctx, fe* v, uint64_t len, hipStream_t s) { return batch_invert_t<F>(ctx, v, len, s); }                    \
    int prefix_product_##NAME(dehalo_ctx* ctx, const fe* in, uint64_t len, fe* out, hipStream_t s) {                                             \
        return prefix_product_t<F>(ctx, in, len, nullptr, 0, len, 1, out, len, s); }                                                             \
    int grand_product_##NAME(dehalo_ctx* ctx, const fe* num, const fe* den, uint64_t len, size_t batch, uint64_t stride, fe* z, hipStream_t s) { \
        return grand_product_t<F>(ctx, num, den, len, batch, stride, z, s); }
