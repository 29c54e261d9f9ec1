// msm.cuh -- Pippenger bucket MSM for gfx950 (replaces upstream
// halo2_proofs::arithmetic::{best_multiexp, multiexp_serial}, halo2_proofs/src/arithmetic.rs
// @ v2023_04_20).  Same group element as best_multiexp; a different
// algorithm, designed around the MI355X memory system rather than around rayon chunks:
//
//  * SRS tables resident in HBM.  dehalo_bases_register may store [2^(c*w)]P_i for every
//    window w (n * W * 64 B; 1 GiB at n = 2^20, c = 16 -- 0.35 % of the 288 GB).  Then all W
//    windows of one MSM share ONE set of 2^(c-1) buckets: no per-window bucket reduction, no
//    doublings between windows.
//  * Signed c-bit digits (buckets halve), digit = 0 skipped (witness columns are mostly
//    small values: SURVEY.md 8(d)).
//  * Counting sort of (bucket, point) pairs with a whole-bucket-range histogram in LDS
//    (2^15 x 4 B = 128 KiB of the CU's 160 KiB): histogram -> scan -> scatter, no digit
//    array in HBM (digits are recomputed from the scalar, ~1 % of the group-add work).
//  * Bucket accumulation split into fixed-length tasks of <= L points regardless of bucket
//    size (a bucket that receives 300 k points of a 0/1 column costs the same per lane as a
//    uniform one), one lane per task, XYZZ mixed additions (8M + 2S); partial sums are merged
//    by two further levels of the same scheme.
//  * Bucket reduction sum_k k*B_k: lanes take 4 consecutive buckets (local running sums),
//    weight their run by the block offset with double-and-add, then a tree of group additions.
//
// Group adds are sequential per lane; a wave64 executes 64 independent bucket chains in
// lock-step.  Integer-only (v_mad_u64_u32); no MFMA (nothing here is a contraction).
#pragma once
#include "ec.cuh"

#define MSM_SORT_THREADS 1024
#define MSM_ACC_THREADS 128
#define MSM_L1 32          // chunk of the first merge level
#define MSM_RED_M 4        // buckets per lane in the bucket reduction
#define MSM_TREE_THREADS 128

struct MsmGeom {
    u32 n;          // scalars per MSM
    u32 table_n;    // registered points (row pitch of the window tables)
    u32 c;          // window bits
    u32 W;          // windows = ceil(256 / c)
    u32 nb;         // buckets per group = 2^(c-1)
    u32 G;          // bucket groups per MSM: 1 (precomputed tables) or W
    u32 batch;      // independent MSMs in this launch
    u32 slices;     // sort blocks per (group, batch)
    u32 L0;         // level-0 task length
};

// ---- scalar -> signed digits -------------------------------------------------------------
// canonical scalar s < 2^255, digits d_w in [-(2^(c-1) - 1), 2^(c-1)], sum d_w 2^(cw) = s.
// Calls f(w, bucket, neg) for every non-zero digit with w in [w_lo, w_hi).
template <class FS, class Fn>
FP_DEV void for_each_digit(const fe& mont_scalar, u32 c, u32 W, u32 w_lo, u32 w_hi, Fn f) {
    fe s = f_from_mont<FS>(mont_scalar);
    const u32 mask = (1u << c) - 1, halfv = 1u << (c - 1);
    u32 carry = 0;
    for (u32 w = 0; w < w_hi; w++) {
        u32 raw = (s.v[0] & mask) + carry;
        // s >>= c   (static indices only: keeps the limbs in registers)
#pragma unroll
        for (int i = 0; i < 7; i++) s.v[i] = (s.v[i] >> c) | (s.v[i + 1] << (32 - c));
        s.v[7] >>= c;
        bool neg = raw > halfv;
        carry = neg ? 1u : 0u;
        u32 mag = neg ? (1u << c) - raw : raw;
        if (mag != 0 && w >= w_lo) f(w, mag - 1, neg);
    }
    (void)W;
}

// ---- sort step 1: per-bucket counts ------------------------------------------------------
// grid (slices, G, batch); dynamic LDS nb * 4 B
template <class FS>
__global__ __launch_bounds__(MSM_SORT_THREADS) void k_msm_hist(MsmGeom g, const fe* scalars, u32* count) {
    extern __shared__ u32 lhist[];
    for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) lhist[b] = 0;
    __syncthreads();
    const u32 grp = blockIdx.y, bat = blockIdx.z;
    const u32 w_lo = g.G == 1 ? 0 : grp, w_hi = g.G == 1 ? g.W : grp + 1;
    const fe* sc = scalars + (u64)bat * g.n;
    const u32 per = (g.n + g.slices - 1) / g.slices;
    const u32 beg = blockIdx.x * per, end = min(beg + per, g.n);
    for (u32 i = beg + threadIdx.x; i < end; i += blockDim.x) {
        fe s = f_load(&sc[i]);
        for_each_digit<FS>(s, g.c, g.W, w_lo, w_hi, [&](u32, u32 bucket, bool) { atomicAdd(&lhist[bucket], 1u); });
    }
    __syncthreads();
    u32* gc = count + ((u64)bat * g.G + grp) * g.nb;
    for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) {
        u32 v = lhist[b];
        if (v) atomicAdd(&gc[b], v);
    }
}

// ---- scans (3 kernels): item offsets and task offsets ------------------------------------
// in: cnt[total]; out: off[total + 1] = exclusive scan of cnt, toff[total + 1] = exclusive scan
// of ceil(cnt / L).  Blocks of SCAN_BLOCK entries.
#define SCAN_THREADS 256
#define SCAN_PER_THREAD 8
#define SCAN_BLOCK (SCAN_THREADS * SCAN_PER_THREAD)

FP_DEV u32 ceil_div_u32(u32 a, u32 b) { return (a + b - 1) / b; }

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const u32* cnt, u32 total, u32 L, u32* bsum_items, u32* bsum_tasks) {
    __shared__ u32 s_i[SCAN_THREADS], s_t[SCAN_THREADS];
    u32 base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    u32 si = 0, st = 0;
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
        u32 idx = base + k;
        u32 v = idx < total ? cnt[idx] : 0;
        si += v; st += ceil_div_u32(v, L);
    }
    s_i[threadIdx.x] = si; s_t[threadIdx.x] = st;
    __syncthreads();
    for (u32 d = SCAN_THREADS / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) { s_i[threadIdx.x] += s_i[threadIdx.x + d]; s_t[threadIdx.x] += s_t[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { bsum_items[blockIdx.x] = s_i[0]; bsum_tasks[blockIdx.x] = s_t[0]; }
}

// single block: exclusive scan of the block sums in place (nblocks <= a few thousand)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_top(u32* bsum_items, u32* bsum_tasks, u32 nblocks) {
    __shared__ u32 s_i[SCAN_THREADS], s_t[SCAN_THREADS];
    __shared__ u32 carry_i, carry_t;
    if (threadIdx.x == 0) { carry_i = 0; carry_t = 0; }
    __syncthreads();
    for (u32 base = 0; base < nblocks; base += SCAN_THREADS) {
        u32 idx = base + threadIdx.x;
        u32 vi = idx < nblocks ? bsum_items[idx] : 0, vt = idx < nblocks ? bsum_tasks[idx] : 0;
        s_i[threadIdx.x] = vi; s_t[threadIdx.x] = vt;
        __syncthreads();
        for (u32 d = 1; d < SCAN_THREADS; d <<= 1) {  // Hillis-Steele inclusive
            u32 ai = threadIdx.x >= d ? s_i[threadIdx.x - d] : 0, at = threadIdx.x >= d ? s_t[threadIdx.x - d] : 0;
            __syncthreads();
            s_i[threadIdx.x] += ai; s_t[threadIdx.x] += at;
            __syncthreads();
        }
        if (idx < nblocks) { bsum_items[idx] = carry_i + s_i[threadIdx.x] - vi; bsum_tasks[idx] = carry_t + s_t[threadIdx.x] - vt; }
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) { carry_i += s_i[threadIdx.x]; carry_t += s_t[threadIdx.x]; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const u32* cnt, u32 total, u32 L, const u32* bsum_items, const u32* bsum_tasks,
                                                             u32* off, u32* toff) {
    __shared__ u32 s_i[SCAN_THREADS], s_t[SCAN_THREADS];
    u32 base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    u32 vi[SCAN_PER_THREAD], vt[SCAN_PER_THREAD];
    u32 si = 0, st = 0;
#pragma unroll
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
        u32 idx = base + k;
        u32 v = idx < total ? cnt[idx] : 0;
        vi[k] = v; vt[k] = ceil_div_u32(v, L);
        si += vi[k]; st += vt[k];
    }
    s_i[threadIdx.x] = si; s_t[threadIdx.x] = st;
    __syncthreads();
    for (u32 d = 1; d < SCAN_THREADS; d <<= 1) {
        u32 ai = threadIdx.x >= d ? s_i[threadIdx.x - d] : 0, at = threadIdx.x >= d ? s_t[threadIdx.x - d] : 0;
        __syncthreads();
        s_i[threadIdx.x] += ai; s_t[threadIdx.x] += at;
        __syncthreads();
    }
    u32 ri = bsum_items[blockIdx.x] + s_i[threadIdx.x] - si;
    u32 rt = bsum_tasks[blockIdx.x] + s_t[threadIdx.x] - st;
#pragma unroll
    for (u32 k = 0; k < SCAN_PER_THREAD; k++) {
        u32 idx = base + k;
        if (idx < total) { off[idx] = ri; toff[idx] = rt; }
        ri += vi[k]; rt += vt[k];
        if (idx + 1 == total) { off[total] = ri; toff[total] = rt; }
    }
}

// cnt_out[b] = toff[b + 1] - toff[b]   (number of partials per bucket = next level's item count)
__global__ void k_diff(const u32* toff, u32 total, u32* cnt_out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) cnt_out[i] = toff[i + 1] - toff[i];
}

// ---- sort step 2: scatter point references into bucket order -----------------------------
// entry = table index | sign << 31.  Same grid / LDS as k_msm_hist.
template <class FS>
__global__ __launch_bounds__(MSM_SORT_THREADS) void k_msm_scatter(MsmGeom g, const fe* scalars, const u32* off, u32* cursor, u32* idx_out) {
    extern __shared__ u32 lhist[];
    for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) lhist[b] = 0;
    __syncthreads();
    const u32 grp = blockIdx.y, bat = blockIdx.z;
    const u32 w_lo = g.G == 1 ? 0 : grp, w_hi = g.G == 1 ? g.W : grp + 1;
    const fe* sc = scalars + (u64)bat * g.n;
    const u32 per = (g.n + g.slices - 1) / g.slices;
    const u32 beg = blockIdx.x * per, end = min(beg + per, g.n);
    for (u32 i = beg + threadIdx.x; i < end; i += blockDim.x) {
        fe s = f_load(&sc[i]);
        for_each_digit<FS>(s, g.c, g.W, w_lo, w_hi, [&](u32, u32 bucket, bool) { atomicAdd(&lhist[bucket], 1u); });
    }
    __syncthreads();
    const u64 gb = ((u64)bat * g.G + grp) * g.nb;
    for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) {
        u32 v = lhist[b];
        if (v) lhist[b] = off[gb + b] + atomicAdd(&cursor[gb + b], v);  // claim a run inside the bucket
    }
    __syncthreads();
    for (u32 i = beg + threadIdx.x; i < end; i += blockDim.x) {
        fe s = f_load(&sc[i]);
        for_each_digit<FS>(s, g.c, g.W, w_lo, w_hi, [&](u32 w, u32 bucket, bool neg) {
            u32 pos = atomicAdd(&lhist[bucket], 1u);
            u32 tidx = g.G == 1 ? w * g.table_n + i : i;
            idx_out[pos] = tidx | (neg ? 0x80000000u : 0u);
        });
    }
}

// largest b in [0, total) with toff[b] <= t   (toff non-decreasing, toff[total] > t)
FP_DEV u32 find_segment(const u32* toff, u32 total, u32 t) {
    u32 lo = 0, hi = total;  // invariant: toff[lo] <= t < toff[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (toff[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- level 0: one lane = one task of <= L0 points of one bucket ---------------------------
template <class CV>
__global__ __launch_bounds__(MSM_ACC_THREADS) void k_msm_accum0(MsmGeom g, u32 total_buckets, const u32* idx, const u32* off, const u32* toff,
                                                               const affine_t* table, xyzz_t* partial) {
    typedef typename CV::Base F;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 ntasks = toff[total_buckets];
    if (t >= ntasks) return;
    u32 b = find_segment(toff, total_buckets, t);
    u32 j = t - toff[b];
    u32 beg = off[b] + j * g.L0;
    u32 end = min(beg + g.L0, off[b + 1]);
    xyzz_t acc = xyzz_identity();
    for (u32 p = beg; p < end; p++) {
        u32 e = idx[p];
        affine_t q = aff_load(&table[e & 0x7fffffffu]);
        if (e >> 31) q.y = f_neg<F>(q.y);
        acc = xyzz_add_mixed<F>(acc, q);
    }
    xyzz_store(&partial[t], acc);
}

// ---- merge levels: segments of XYZZ partials --------------------------------------------
// level 1: task = <= L consecutive partials of one bucket; level 2 (L = 0xffffffff): whole segment
template <class CV>
__global__ __launch_bounds__(MSM_ACC_THREADS) void k_msm_merge(u32 total_buckets, u32 L, const u32* seg_off, const u32* toff, const xyzz_t* in,
                                                              xyzz_t* out, u32 per_bucket) {
    typedef typename CV::Base F;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 b, beg, end;
    if (per_bucket) {
        if (t >= total_buckets) return;
        b = t; beg = seg_off[b]; end = seg_off[b + 1];
    } else {
        u32 ntasks = toff[total_buckets];
        if (t >= ntasks) return;
        b = find_segment(toff, total_buckets, t);
        u32 j = t - toff[b];
        beg = seg_off[b] + j * L;
        end = min(beg + L, seg_off[b + 1]);
    }
    xyzz_t acc = xyzz_identity();
    for (u32 p = beg; p < end; p++) acc = xyzz_add<F>(acc, xyzz_load(&in[p]));
    xyzz_store(&out[t], acc);
}

// ---- bucket reduction: sum_k (k + 1) * B_k per group --------------------------------------
// lane (group, t) takes buckets [t*M, t*M + M): contribution = sum (k - k0 + 1) B_k + k0 * sum B_k
template <class CV>
__global__ __launch_bounds__(MSM_ACC_THREADS) void k_msm_reduce_local(u32 nb, u32 total_groups, const xyzz_t* buckets, xyzz_t* contrib) {
    typedef typename CV::Base F;
    const u32 per_group = (nb + MSM_RED_M - 1) / MSM_RED_M;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= per_group * total_groups) return;
    u32 grp = gid / per_group, t = gid % per_group;
    u32 k0 = t * MSM_RED_M, k1 = min(k0 + MSM_RED_M, nb);
    const xyzz_t* B = buckets + (u64)grp * nb;
    xyzz_t run = xyzz_identity(), acc = xyzz_identity();
    for (u32 k = k1; k-- > k0;) {
        run = xyzz_add<F>(run, xyzz_load(&B[k]));
        acc = xyzz_add<F>(acc, run);
    }
    // k0 * run, MSB-first double-and-add (k0 < 2^15)
    xyzz_t w = xyzz_identity();
    if (k0) {
        int top = 31 - __clz(k0);
        for (int bit = top; bit >= 0; bit--) {
            w = xyzz_double<F>(w);
            if ((k0 >> bit) & 1) w = xyzz_add<F>(w, run);
        }
    }
    xyzz_store(&contrib[gid], xyzz_add<F>(acc, w));
}

// tree sum: in[groups][cnt] -> out[groups][ceil(cnt / (2 * MSM_TREE_THREADS))]
template <class CV>
__global__ __launch_bounds__(MSM_TREE_THREADS) void k_msm_tree_sum(const xyzz_t* in, u32 cnt, xyzz_t* out, u32 out_cnt) {
    typedef typename CV::Base F;
    __shared__ xyzz_t sh[MSM_TREE_THREADS];
    u32 grp = blockIdx.y;
    const xyzz_t* src = in + (u64)grp * cnt;
    u32 i0 = blockIdx.x * (2 * MSM_TREE_THREADS) + threadIdx.x;
    xyzz_t a = i0 < cnt ? xyzz_load(&src[i0]) : xyzz_identity();
    u32 i1 = i0 + MSM_TREE_THREADS;
    if (i1 < cnt) a = xyzz_add<F>(a, xyzz_load(&src[i1]));
    sh[threadIdx.x] = a;
    __syncthreads();
    for (u32 d = MSM_TREE_THREADS / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            xyzz_t x = xyzz_add<F>(sh[threadIdx.x], sh[threadIdx.x + d]);
            sh[threadIdx.x] = x;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) xyzz_store(&out[(u64)grp * out_cnt + blockIdx.x], sh[0]);
}

// ---- final: window sums -> one Jacobian point per MSM -------------------------------------
// G == 1: convert.  G == W: result = sum_w 2^(c*w) S_w; lane w doubles S_w c*w times, then an
// LDS tree (the one-shot, unregistered-bases path only).
template <class CV>
__global__ __launch_bounds__(64) void k_msm_final(MsmGeom g, const xyzz_t* group_sums, jacobian_t* out) {
    typedef typename CV::Base F;
    __shared__ xyzz_t sh[64];
    u32 bat = blockIdx.x;
    u32 w = threadIdx.x;
    xyzz_t s = xyzz_identity();
    if (w < g.G) {
        s = xyzz_load(&group_sums[(u64)bat * g.G + w]);
        if (g.G > 1) {
            u32 nd = g.c * w;
            for (u32 i = 0; i < nd; i++) s = xyzz_double<F>(s);
        }
    }
    sh[w] = s;
    __syncthreads();
    for (u32 d = 32; d > 0; d >>= 1) {
        if (w < d) {
            xyzz_t x = xyzz_add<F>(sh[w], sh[w + d]);
            sh[w] = x;
        }
        __syncthreads();
    }
    if (w == 0) {
        jacobian_t j = xyzz_to_jacobian<F>(sh[0]);
        f_store(&out[bat].x, j.x); f_store(&out[bat].y, j.y); f_store(&out[bat].z, j.z);
    }
}

// ---- SRS table precomputation: table[w][i] = [2^(c*w)] P_i, affine -------------------------
template <class CV>
__global__ __launch_bounds__(128) void k_msm_precompute(affine_t* table, u32 n, u32 c, u32 W) {
    typedef typename CV::Base F;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_t p = aff_load(&table[i]);
    xyzz_t cur = xyzz_from_affine<F>(p);
    for (u32 w = 1; w < W; w++) {
        for (u32 k = 0; k < c; k++) cur = xyzz_double<F>(cur);
        affine_t a = xyzz_to_affine<F>(cur);
        aff_store(&table[(u64)w * n + i], a);
    }
}

// Jacobian -> affine for MSM outputs (dehalo_to_affine)
template <class CV>
__global__ void k_jac_to_affine(const jacobian_t* in, affine_t* out, u32 count) {
    typedef typename CV::Base F;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    fe x = f_load(&in[i].x), y = f_load(&in[i].y), z = f_load(&in[i].z);
    affine_t a;
    if (f_is_zero(z)) { a.x = f_zero(); a.y = f_zero(); }
    else {
        fe zi = f_inv<F>(z);
        fe zi2 = f_sqr<F>(zi);
        a.x = f_mul<F>(x, zi2);
        a.y = f_mul<F>(y, f_mul<F>(zi2, zi));
    }
    aff_store(&out[i], a);
}
