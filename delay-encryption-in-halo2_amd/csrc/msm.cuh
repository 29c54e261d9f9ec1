// msm.cuh -- Pippenger bucket MSM for gfx950 (replaces upstream
// halo2_proofs::arithmetic::{best_multiexp, multiexp_serial}, halo2_proofs/src/arithmetic.rs
// @ v2023_04_20).  Same group element as best_multiexp; a different
// algorithm, designed around the MI355X memory system rather than around rayon chunks:
//
//  * SRS tables resident in HBM.  dehalo_bases_register may store [2^(c*w)]P_i for every
//    window w (n * W * 64 B; 1 GiB at n = 2^20, c = 16 -- 0.35 % of the 288 GB).  Then all W
//    windows of one MSM share ONE set of 2^(c-1) buckets: no per-window bucket reduction, no
//    doublings between windows.  Table entries are kept in the kernels' internal field form
//    (x * 2^261 mod p, canonical, packed in 32 B) so the hot loop never converts.
//  * Signed c-bit digits (buckets halve), digit = 0 skipped (witness columns are mostly
//    small values: SURVEY.md 8(d)).
//  * Stable counting sort of (bucket, point) pairs with a whole-bucket-range histogram in LDS
//    (2^15 x 4 B = 128 KiB of the CU's 160 KiB): per-block histograms -> column scan -> bucket
//    scan -> single-pass scatter; no global atomics, no digit array in HBM (digits are
//    recomputed from the scalar, ~1 % of the group-add work).
//  * Bucket accumulation split into fixed-length tasks of <= L points regardless of bucket
//    size (a bucket that receives 300 k points of a 0/1 column costs the same per lane as a
//    uniform one), one lane per task, XYZZ mixed additions on the carry-free 9 x 29-bit field
//    (fp29.cuh / ec29.cuh); a bucket's partial sums are merged by a lane group sized to their
//    number (1 / 8 / 64 / 1024 lanes).
//  * Bucket reduction sum_k k*B_k: lanes take 4 consecutive buckets (local running sums),
//    weight their run by the block offset with double-and-add, then a tree of group additions.
//
// Group adds are sequential per lane; a wave64 executes 64 independent bucket chains in
// lock-step.  Integer-only (v_mad_u64_u32); no MFMA (nothing here is a contraction).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "ec29.cuh"
#include "internal.hpp"

#define MSM_SORT_THREADS 1024
#define MSM_ACC_THREADS 128        // the accumulation's default block: 2 waves (4 waves x 122 VGPRs fill a SIMD's register file)
#define MSM_ACC_THREADS_MAX 768    // msm_acc_block = 768: ONE block of 12 waves per CU = 3 waves per SIMD and no room for a second block -- a quarter of every
                                   // SIMD's registers (and all of the LDS) stays free for the <= 128-VGPR kernels of the other contexts
// (measured: 2 and 3 resident waves per SIMD give the same k_msm_accum0 time -- the loop is
// VALU-issue-bound -- and capping residency at 2 did not improve multi-stream overlap)
#ifndef MSM_ACC_WAVES_ATTR
#ifdef MSM_ACC_CAP
#define MSM_ACC_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(MSM_ACC_CAP, MSM_ACC_CAP)))
#else
#define MSM_ACC_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))      // the register budget of four waves per SIMD (128), whatever the block size
#endif
#endif
#ifdef MSM_HIST_VGPR64
#define MSM_HIST_ATTR __attribute__((amdgpu_num_vgpr(64)))
#else
#define MSM_HIST_ATTR
#endif
#define MSM_MERGE_COUNTERS 8   // merge-class counters in ws_counters, in front of [L0 | M]

// Experiment, off (round 4): the MSM's latency-bound kernels (sort, scans, merge, bucket reduction) raising their waves' issue priority with s_setprio, so that beside
// another stream's throughput kernels (the side context's NTTs in a proof, a neighbour step's accumulation) the SIMD's arbiter takes their instructions first.
// Measured with -DMSM_TAIL_PRIO=2 against 0 on one box: the kernels ALONE got slower (sort 0.194 -> 0.216 ms, merge + reduction 0.253 -> 0.277 at 2^20), the step, a
// k = 17 proof (7.58-7.60 against 7.48-7.54 ms) and batch mode (154.3-154.9 against 153.9-154.7 proofs/s) did not move.
#ifndef MSM_TAIL_PRIO
#define MSM_TAIL_PRIO 0
#endif
FP_DEV void msm_tail_prio() {
#if MSM_TAIL_PRIO > 0
    __builtin_amdgcn_s_setprio(MSM_TAIL_PRIO);
#endif
}

// classes of k_msm_merge2 (msm_bred.cuh) by the number S of partial sums of a bucket; S > MERGE2_CHUNK: cut into parts of MERGE2_CHUNK records
#define MERGE2_CHUNK 512
FP_DEV u32 merge2_class(u32 S) { return S <= 2 ? 0u : S <= 4 ? 1u : S <= 8 ? 2u : S <= 64 ? 3u : S <= MERGE2_CHUNK ? 4u : 5u; }

#define MSM_IDX_FIRST 0x40000000u     // sorted entry, bit 30: first point of its bucket
#define MSM_IDX_MASK 0x3fffffffu      // table index (precomputed tables: window * table_n + point < 2^30, checked at registration)

struct MsmGeom {
    u32 n;          // scalars per MSM
    u32 table_n;    // registered points (row pitch of the window tables)
    u32 c;          // window bits
    u32 W;          // windows = ceil(256 / c)
    u32 nb;         // buckets per group = 2^(c-1)
    u32 G;          // bucket groups per MSM: 1 (precomputed tables) or W
    u32 batch;      // independent MSMs in this launch
    u32 slices;     // sort blocks per (group, batch)
    u32 L0;         // points per lane of k_msm_accum0
    u32 wb;         // G == W only: windows (= bucket groups) one sort block covers; the sort's grid.y = ceil(G / wb)
};

// ---- scalar -> signed digits -------------------------------------------------------------
// canonical scalar s < 2^255, digits d_w in [-(2^(c-1) - 1), 2^(c-1)], sum d_w 2^(cw) = s.
// Calls f(w, bucket, neg) for every non-zero digit with w in [w_lo, w_hi).
FP_DEV bool fe_is_zero_words(const fe& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0; }

template <class FS, class Fn>
FP_DEV void for_each_digit(const fe& mont_scalar, u32 c, u32 w_lo, u32 w_hi, Fn f) {
    if (fe_is_zero_words(mont_scalar)) return;           // 0 * R = 0: a zero scalar has no digits (most rows of a permuted lookup column)
    fe s = f_from_mont<FS>(mont_scalar);
    // windows that can hold a digit: those covering the scalar's bits, plus one for the last carry (small witness values end after a
    // few windows; computed once per scalar, not tested per digit)
    u32 bits = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) if (s.v[i]) bits = 32u * i + (32u - (u32)__clz((int)s.v[i]));
    const u32 w_end = min(w_hi, (bits + c - 1) / c + 1);
    const u32 mask = (1u << c) - 1, halfv = 1u << (c - 1);
    u32 carry = 0;
    for (u32 w = 0; w < w_end; w++) {
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
}

// The same recoding as for_each_digit, eight digits at a time into arrays with static indices, so a
// caller can issue eight independent LDS atomics back to back instead of waiting for each result.
struct DigitStream {
    fe s;
    u32 carry, c, mask, halfv;
    template <class FS>
    FP_DEV void init(const fe& mont_scalar, u32 c_) {
        s = f_from_mont<FS>(mont_scalar);
        c = c_; mask = (1u << c) - 1; halfv = 1u << (c - 1); carry = 0;
    }
    FP_DEV bool exhausted() const { return carry == 0 && fe_is_zero_words(s); }
    // next digit: bucket index (0xffffffff for a zero digit) and sign
    FP_DEV void next(u32& bucket, bool& neg) {
        u32 raw = (s.v[0] & mask) + carry;
#pragma unroll
        for (int i = 0; i < 7; i++) s.v[i] = (s.v[i] >> c) | (s.v[i + 1] << (32 - c));
        s.v[7] >>= c;
        neg = raw > halfv;
        carry = neg ? 1u : 0u;
        u32 mag = neg ? (1u << c) - raw : raw;
        bucket = mag - 1;   // mag == 0 -> 0xffffffff
    }
};

// second-level split of the bucket index: buckets = partitions x 2^sub sub-buckets
// (17-bit windows: 512 sub-buckets, so that the 2^16 buckets are still 128 partitions -- k_msm_part's lists -- and a k_msm_bucket round is still 2048 pairs)
__host__ __device__ inline u32 msm_sub_bits(u32 c) { return c - 1 < 8 ? c - 1 : (c >= 17 ? 9u : 8u); }

// Groups of equal scalars inside a wave.  A grand-product column over unused rows, a permuted lookup column's runs of equal inputs and a sorted table hand a
// wave ONE value -- or two or three where runs meet; lanes that hold the same scalar have the same digits, so per window one lane per GROUP counts / reserves
// for all of them (64 LDS atomics on one word per digit otherwise: 5 ms of k_msm_hist on the grand products of a k = 20 proof before the one-value case was
// caught in round 2, and still 0.23 ms of k_msm_part on the permuted columns of a k = 17 proof, whose waves straddle two runs, until round 4).
// -> false when the active lanes hold more than `max_groups` distinct values (the general path then); leader / cnt / rank describe the lane's group.
FP_DEV bool wave_value_groups(const fe& s, bool active, u32 lane, u32 max_groups, u32& leader, u32& cnt, u32& rank) {
    unsigned long long rem = __ballot(active);
    leader = lane; cnt = 1; rank = 0;
    for (u32 n = 0; rem; n++) {
        if (n == max_groups) return false;
        const int ld = __ffsll((long long)rem) - 1;
        bool eq = active;
#pragma unroll
        for (int w = 0; w < 8; w++) eq &= s.v[w] == (u32)__shfl((int)s.v[w], ld);
        const unsigned long long m = __ballot(eq);
        if (eq) { leader = (u32)ld; cnt = (u32)__popcll(m); rank = (u32)__popcll(m & ((1ull << lane) - 1)); }
        rem &= ~m;
    }
    return true;
}

// ---- sort step 1: per-block histograms ----------------------------------------------------
// grid (slices, G, batch); dynamic LDS nb * 4 B.  Each block stores its whole local histogram
// (plain coalesced stores): bh[group][slice][bucket].  No global atomics anywhere in the sort:
// 256 blocks claiming runs in the same 2^15 counters with returning atomics cost 0.2 ms.
// PACK16: two 16-bit counters a word -- half the LDS (64 KiB for 2^15 buckets, so that a sort block shares a compute unit with an NTT tile or the merge /
// reduction blocks of another context instead of waiting for a CU with all of its LDS free); the host selects it only when a block's scalars times the window
// count stay below 2^16, so no counter can carry into its neighbour.
template <class FS, bool PACK16>
__global__ __launch_bounds__(MSM_SORT_THREADS) MSM_HIST_ATTR void k_msm_hist(MsmGeom g, const fe* scalars, u32* bh, u32* pc) {
    msm_tail_prio();
    extern __shared__ u32 lhist[];
    // G == 1: the block counts every window into the one bucket set.  G == W: it covers windows [w_lo, w_hi), one bucket set each
    // (a scalar is decoded once per block, not once per window: ceil(W / wb) passes over the scalars instead of W).
    const u32 bat = blockIdx.z;
    const u32 w_lo = g.G == 1 ? 0 : blockIdx.y * g.wb, w_hi = g.G == 1 ? g.W : min(g.W, w_lo + g.wb);
    const u32 ng = g.G == 1 ? 1 : w_hi - w_lo, grp0 = g.G == 1 ? 0 : w_lo;
    for (u32 b = threadIdx.x; b < (PACK16 ? ng * g.nb / 2 : ng * g.nb); b += blockDim.x) lhist[b] = 0;
    __syncthreads();
    const fe* sc = scalars + (u64)bat * g.n;
    const u32 per = (g.n + g.slices - 1) / g.slices;
    const u32 beg = blockIdx.x * per, end = min(beg + per, g.n);
    const bool one = g.G == 1;
    for (u32 i0 = beg + threadIdx.x; i0 < end; i0 += 4 * blockDim.x) {      // four scalar loads in flight per lane
        fe s4[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (i0 + j * blockDim.x < end) s4[j] = f_load(&sc[i0 + j * blockDim.x]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool valid = i0 + j * blockDim.x < end;
            // A wave whose 64 consecutive rows hold ONE value (a grand-product column over unused rows, a permuted column's run of
            // equal inputs) would send 64 atomics to the same LDS word for every digit; one lane adds the count instead.
            // (k = 20 grand products, 7 columns: this kernel 5.2 ms -> see DESIGN.md)
            u32 gl, weight, grank;
            const bool grouped = wave_value_groups(s4[j], valid, threadIdx.x & 63, 4, gl, weight, grank);
            if (!grouped) weight = 1;
            const bool counts = grouped ? valid && (threadIdx.x & 63) == gl : valid;
            if (counts)
                for_each_digit<FS>(s4[j], g.c, w_lo, w_hi, [&](u32 w, u32 bucket, bool) {
                    const u32 slot = (one ? 0u : (w - w_lo) * g.nb) + bucket;
                    if (PACK16) atomicAdd(&lhist[slot >> 1], weight << ((slot & 1u) << 4));
                    else atomicAdd(&lhist[slot], weight);
                });
        }
    }
    __syncthreads();
    const u32 sub = msm_sub_bits(g.c), P = g.nb >> sub;
    for (u32 gi = 0; gi < ng; gi++) {
        const u64 gidx = (u64)bat * g.G + grp0 + gi;
        const u32* lh = lhist + (PACK16 ? gi * g.nb / 2 : gi * g.nb);
        if (PACK16) {      // the packed words as they are: a 16-bit count per bucket in memory (half the bytes written here and read by the column scan)
            u32* out = bh + ((gidx * g.slices + blockIdx.x) * g.nb) / 2;      // (bh: the 16-bit array's base, run_msm_t)
            for (u32 w = threadIdx.x; w < g.nb / 2; w += blockDim.x) out[w] = lh[w];
        } else {
            u32* out = bh + (gidx * g.slices + blockIdx.x) * g.nb;
            for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) out[b] = lh[b];
        }
        // partition = the bucket's top bits; its count for this slice (second sort level, see k_msm_part)
        u32* pout = pc + (gidx * g.slices + blockIdx.x) * P;
        // one wave per partition at a time: its 64 lanes read consecutive LDS words (eight lanes striding through a partition's counters, as until round 6, put
        // all of them on one bank: 14 us of this kernel with the 512 counters of a 17-bit window's partitions)
        const u32 nsub = 1u << sub, lane = threadIdx.x & 63;
        for (u32 q = threadIdx.x >> 6; q < P; q += blockDim.x >> 6) {
            u32 sum = 0;
            if (PACK16) {
                for (u32 w = lane; w < nsub / 2; w += 64) { const u32 x = lh[(q << sub) / 2 + w]; sum += (x & 0xffffu) + (x >> 16); }
            } else {
                for (u32 b = lane; b < nsub; b += 64) sum += lh[(q << sub) + b];
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d);
            if (lane == 0) pout[q] = sum;
        }
    }
}

// column scan: for every bucket, exclusive prefix over the slices (in place) and the total count.  One launch covers two tables:
// blocks [0, blocks_a) the per-bucket histograms (bh, counts out), the rest the per-partition counts (pc, no totals).
// (src16: the counts as 16-bit values, k_msm_hist<PACK16>; null: they sit in `bh` itself and are replaced by their prefixes)
template <bool SRC16>
FP_DEV u32 colscan_one(u32 nb, u32 slices, u32 total_buckets, u32* bh, const unsigned short* src16, u32* count, u32 gb) {
    if (gb >= total_buckets) return 0;
    u32 grp = gb / nb, b = gb - grp * nb;
    u32* col = bh + (u64)grp * slices * nb + b;
    const unsigned short* col16 = SRC16 ? src16 + (u64)grp * slices * nb + b : nullptr;
    u32 run = 0;
    u32 k = 0;
    for (; k + 16 <= slices; k += 16) {   // 16 independent loads in flight per lane, then the prefix
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = SRC16 ? (u32)col16[(u64)(k + j) * nb] : col[(u64)(k + j) * nb];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            col[(u64)(k + j) * nb] = run;
            run += v[j];
        }
    }
    for (; k < slices; k++) {
        u32 v = SRC16 ? (u32)col16[(u64)k * nb] : col[(u64)k * nb];
        col[(u64)k * nb] = run;
        run += v;
    }
    if (count) count[gb] = run;
    return run;
}
// (256 threads.)  A block of the first table also leaves the sums of its 256 buckets -- points and non-empty buckets -- for the scan below:
// bsum_items / bsum_tasks [blockIdx] (until round 4 a launch of its own, k_scan_block_sums).
#define SCAN_THREADS 256
static __global__ __launch_bounds__(SCAN_THREADS) void k_msm_colscan(u32 nb, u32 slices, u32 total_buckets, u32* bh, const unsigned short* bh16, u32* count, u32 blocks_a, u32 P,
                                                                     u32 total_parts, u32* pc, u32* bsum_items, u32* bsum_tasks, u32* merge_counters) {
    msm_tail_prio();
    if (blockIdx.x == 0 && threadIdx.x < MSM_MERGE_COUNTERS) merge_counters[threadIdx.x] = 0;      // (k_scan_offsets counts the merge classes: zero before it starts, no fill launch)
    if (blockIdx.x >= blocks_a) {
        colscan_one<false>(P, slices, total_parts, pc, nullptr, nullptr, (blockIdx.x - blocks_a) * blockDim.x + threadIdx.x);
        return;
    }
    const u32 gb0 = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 v = bh16 ? colscan_one<true>(nb, slices, total_buckets, bh, bh16, count, gb0) : colscan_one<false>(nb, slices, total_buckets, bh, nullptr, count, gb0);
    u32 si = v, st = v ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { si += __shfl_down(si, d); st += __shfl_down(st, d); }
    __shared__ u32 w_i[SCAN_THREADS / 64], w_t[SCAN_THREADS / 64];
    if ((threadIdx.x & 63) == 0) { w_i[threadIdx.x >> 6] = si; w_t[threadIdx.x >> 6] = st; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 a = 0, b = 0;
        for (u32 w = 0; w < SCAN_THREADS / 64; w++) { a += w_i[w]; b += w_t[w]; }
        bsum_items[blockIdx.x] = a; bsum_tasks[blockIdx.x] = b;
    }
}

// ---- the scan: point offsets and record ranges, ONE launch after the column scan (round 4; three before: block sums, a one-block scan of them, apply) --------
// in: cnt[total] and the sums of its blocks of 256 (k_msm_colscan); out: off[total + 1] = exclusive scan of cnt, nrank[total + 1] = exclusive count of non-empty
// buckets, each bucket's range of partial-sum records, and geo.  Every block adds up the block sums before it and all of them itself (a few hundred to a few
// thousand words from the L2: cheaper than a launch that does it once), so every block knows M and derives the same L.
// geo[0] = L fixes the accumulation's points per lane from the ACTUAL number of sorted points M (zero digits are never sorted: a witness column of small values
// has a fraction of batch * n * W): L = ceil(M / (rounds * resident)) with rounds = ceil(M / (resident * lmax)), at least 4; geo[1] = M.  (Sized from the upper
// bound on the host, a sparse column left most lanes idle and the rest with full-length chains: advice columns took half the time of dense ones with a fifth of
// the points.)  off[b] = points before bucket b.  The accumulation cuts the sorted list into ranges of L points, one per lane, regardless of bucket boundaries;
// lane g writes one partial sum per bucket its range touches, at record  g + (number of non-empty buckets before that bucket)  -- consecutive for the lanes of
// one bucket.  rbeg/rend[b] = that bucket's record range (equal when empty).
// It also queues every bucket with S >= 2 partial sums in the list of its merge class (k_msm_merge2; `lists` null: not wanted) -- S follows from the offsets alone, so
// the lists are ready before the accumulation starts instead of in a launch between it and the merge: one lane per bucket, one atomic per wave and class.
//   classes 0 .. 4: lists[class][..] = bucket;   S > MERGE2_CHUNK: list 5 holds one entry (slot, part) per part of MERGE2_CHUNK records, list 6 one entry
//   (bucket, first part, parts, arrival counter) per such bucket.
static __global__ __launch_bounds__(SCAN_THREADS) void k_scan_offsets(const u32* cnt, u32 total, const u32* bsum_items, const u32* bsum_tasks, u32 nblocks, u32* geo, u32 resident,
                                                                      u32 lmax, u32 lcap, u32 kmin, u32* off, u32* nrank, u32* rbeg, u32* rend, u32* lists, u32 cap) {
    msm_tail_prio();
    __shared__ u32 s_i[SCAN_THREADS], s_t[SCAN_THREADS];
    __shared__ u32 w4[4][SCAN_THREADS / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 bi = 0, bt = 0, ti = 0;                      // points / non-empty buckets in the blocks before this one; points in all blocks
    for (u32 j = tid; j < nblocks; j += SCAN_THREADS) {
        const u32 a = bsum_items[j];
        ti += a;
        if (j < blockIdx.x) { bi += a; bt += bsum_tasks[j]; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { bi += __shfl_down(bi, d); bt += __shfl_down(bt, d); ti += __shfl_down(ti, d); }
    if (lane == 0) { w4[0][wave] = bi; w4[1][wave] = bt; w4[2][wave] = ti; }
    const u32 idx = blockIdx.x * SCAN_THREADS + tid;
    const u32 vi = idx < total ? cnt[idx] : 0, vt = vi ? 1u : 0u;
    s_i[tid] = vi; s_t[tid] = vt;
    __syncthreads();
    bi = 0; bt = 0; ti = 0;
    for (u32 w = 0; w < SCAN_THREADS / 64; w++) { bi += w4[0][w]; bt += w4[1][w]; ti += w4[2][w]; }
    const u64 M = ti;
    u64 L;
    if (lcap) {   // layers of one wave per SIMD (`resident` lanes each): kmin, ..., 4 (a full chip), 6, 8, ... until a lane has <= lcap points
        u64 k = kmin;
        while (M > k * resident * lcap) k += k < 4 ? 1 : 2;
        L = max((u64)4, (M + k * resident - 1) / (k * resident));
    } else {
        const u64 rounds = max((u64)1, (M + (u64)resident * lmax - 1) / ((u64)resident * lmax));
        L = (M + rounds * resident - 1) / (rounds * resident);
        L = min((u64)lmax, max((u64)4, L));
    }
    if (blockIdx.x == 0 && tid == 0) { geo[0] = (u32)L; geo[1] = (u32)M; }
    for (u32 d = 1; d < SCAN_THREADS; d <<= 1) {
        const u32 ai = tid >= d ? s_i[tid - d] : 0, at = tid >= d ? s_t[tid - d] : 0;
        __syncthreads();
        s_i[tid] += ai; s_t[tid] += at;
        __syncthreads();
    }
    const u32 ri = bi + s_i[tid] - vi, rt = bt + s_t[tid] - vt;
    u32 S = 0;
    if (idx < total) {
        off[idx] = ri; nrank[idx] = rt;
        const u32 b0 = (u32)(ri / L) + rt;
        const u32 b1 = vi ? (u32)((ri + vi - 1) / L) + rt + 1 : b0;
        rbeg[idx] = b0;
        rend[idx] = b1;
        S = b1 - b0;
        if (idx + 1 == total) { off[total] = ri + vi; nrank[total] = rt + vt; }
    }
    if (!lists) return;
    u32* counters = geo - MSM_MERGE_COUNTERS;      // the merge-class counters sit just below geo (ws_counters); zeroed by k_msm_colscan
    const u32 cls = S <= 1 ? 8u : merge2_class(S);
    // one atomic per BLOCK and class (per wave and class, 256 blocks of a four-column 2^17 launch sent 5,000 atomics to five words: 25 us of this kernel)
    __shared__ u32 wcnt[5][SCAN_THREADS / 64], cbase[5];
    unsigned long long mine = 0;
    for (u32 c = 0; c < 5; c++) {
        const unsigned long long mask = __ballot(cls == c);
        if (cls == c) mine = mask;
        if (lane == 0) wcnt[c][wave] = (u32)__popcll(mask);
    }
    __syncthreads();
    if (tid < 5) {
        u32 tot = 0;
        for (u32 w = 0; w < SCAN_THREADS / 64; w++) tot += wcnt[tid][w];
        cbase[tid] = tot ? atomicAdd(&counters[tid], tot) : 0u;
    }
    __syncthreads();
    if (cls < 5) {
        u32 pos = cbase[cls] + (u32)__popcll(mine & ((1ull << lane) - 1));
        for (u32 w = 0; w < wave; w++) pos += wcnt[cls][w];
        lists[(size_t)cls * cap + pos] = idx;
    }
    if (cls == 5u) {      // rare: its own atomics
        const u32 parts = (S + MERGE2_CHUNK - 1) / MERGE2_CHUNK;
        const u32 slot = atomicAdd(&counters[6], 1u), first = atomicAdd(&counters[5], parts);
        u32* hb = lists + (size_t)6 * cap + 4 * (size_t)slot;
        hb[0] = idx; hb[1] = first; hb[2] = parts; hb[3] = 0;
        for (u32 p = 0; p < parts; p++) {
            lists[(size_t)5 * cap + 2 * (size_t)(first + p)] = slot;
            lists[(size_t)5 * cap + 2 * (size_t)(first + p) + 1] = p;
        }
    }
}

// ---- sort step 2: two-level stable scatter -------------------------------------------------
// A one-level scatter gives every block ~2 items per bucket run: 16.8 M isolated 4-byte stores
// (0.22 ms).  Two levels keep stores sequential per open cache line:
//   k_msm_part   : slice k appends its (sub-bucket, reference) pairs to the run of each PARTITION
//                  (top bits of the bucket; 128 partitions at c = 16): ~512 items per run, one open
//                  line per partition per block;
//   k_msm_bucket : block (partition p, slice k) distributes that run over the partition's 2^8
//                  buckets.  The partition's buckets span one contiguous 0.5 MiB region and all
//                  blocks of a partition are given block ids congruent mod 8 (one XCD under the
//                  observed round-robin placement -- speed only), so that XCD's L2 merges them.
// Positions are known without atomics on global memory: bucket offsets + per-slice prefixes.
// Every WAVE groups the 64 x 8 pairs of a sub-round by partition in its own LDS slice (count, scan,
// place -- wave-synchronous, no block barrier), reserves its share of every partition run with one
// returning atomic per partition, and copies the staged pairs out in staging order: a store
// instruction then touches ~8-16 cache lines instead of 64 (these kernels wait to ISSUE stores).
// LDS of one wave: the staging of 512 pairs (8 B + a 2-B partition tag each) and four words per list (count, first slot, cursor, reserved position); `lists` = 128,
// or 256 for the 256 partitions of a 17-bit window
#define MSM_PART_WAVE_LDS(lists) (4 * (lists) * 4 + 512 * 8 + 512 * 2)
template <class FS>
__global__ __launch_bounds__(MSM_SORT_THREADS) void k_msm_part(MsmGeom g, const fe* scalars, const u32* off, const u32* pc, unsigned long long* pairs, u32 lists) {
    msm_tail_prio();
    extern __shared__ __align__(8) unsigned char part_smem[];
    const u32 bat = blockIdx.z;
    const u32 sub = msm_sub_bits(g.c), P = g.nb >> sub, submask = (1u << sub) - 1;
    // windows [w_lo, w_hi) of this block and their bucket groups (k_msm_hist): the runs of (group, partition) pairs, PT <= lists of them
    const u32 w_lo = g.G == 1 ? 0 : blockIdx.y * g.wb, w_hi = g.G == 1 ? g.W : min(g.W, w_lo + g.wb);
    const u32 ng = g.G == 1 ? 1 : w_hi - w_lo, grp0 = g.G == 1 ? 0 : w_lo;
    const u32 PT = ng * P;
    const bool one = g.G == 1;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    u32* pcur = reinterpret_cast<u32*>(part_smem);                               // block-shared run cursors [lists]
    unsigned char* wbase = part_smem + 4 * lists + (size_t)wave * MSM_PART_WAVE_LDS(lists);  // this wave's slice
    unsigned long long* stage = reinterpret_cast<unsigned long long*>(wbase);   // [512]
    u32* lcnt = reinterpret_cast<u32*>(wbase + 512 * 8);                         // [lists] pairs per partition
    u32* lbase = lcnt + lists;                                                   // first staging slot
    u32* lcur = lbase + lists;                                                   // placement cursor
    u32* gbase = lcur + lists;                                                   // reserved global position
    unsigned short* stageq = reinterpret_cast<unsigned short*>(gbase + lists);   // [512] partition of a staged pair
    for (u32 qq = threadIdx.x; qq < PT; qq += blockDim.x) {
        const u32 gi = qq / P, q = qq - gi * P;
        const u64 gidx = (u64)bat * g.G + grp0 + gi;
        pcur[qq] = off[gidx * g.nb + (q << sub)] + pc[(gidx * g.slices + blockIdx.x) * P + q];
    }
    __syncthreads();
    const fe* sc = scalars + (u64)bat * g.n;
    const u32 per = (g.n + g.slices - 1) / g.slices;
    const u32 beg = blockIdx.x * per, end = min(beg + per, g.n);
    (void)nwaves;
    fe s_next;
    if (beg + wave * 64 + lane < end) s_next = f_load(&sc[beg + wave * 64 + lane]);
    for (u32 i0 = beg + wave * 64; i0 < end; i0 += blockDim.x) {       // every lane of a wave runs the same trip count
        const u32 i = i0 + lane;
        const bool have = i < end;
        const fe s_cur = s_next;
        if (i + blockDim.x < end) s_next = f_load(&sc[i + blockDim.x]);   // the next round's scalar is in flight during this one
        // (uniform per wave) a wave of zero scalars -- the long zero run of a sorted, permuted lookup column -- has nothing to place,
        // and a wave of small witness values runs out of digits after the first windows
        const bool nonzero = have && !fe_is_zero_words(s_cur);
        const unsigned long long actm = __ballot(nonzero);
        if (actm == 0) continue;
        DigitStream ds;
        if (nonzero) ds.template init<FS>(s_cur, g.c);
        {   // few distinct values in the wave (wave_value_groups above): the lanes of a group have the same digits, so a window's pairs of a group go to one
            // partition run -- one reservation per window by the group's first lane, positions by rank inside the group, no staging
            u32 gl, gcnt, grank;
            if (__popcll(actm) > 1 && wave_value_groups(s_cur, nonzero, lane, 4, gl, gcnt, grank)) {
                for (u32 w = 0; w < w_hi; w++) {
                    if (__ballot(nonzero && !ds.exhausted()) == 0) break;
                    u32 bk = 0xffffffffu; bool ng1 = false;
                    if (nonzero) ds.next(bk, ng1);                       // (uniform across a group; an exhausted stream yields no digit)
                    const bool put = nonzero && w >= w_lo && bk != 0xffffffffu;
                    const u32 q = put ? (one ? 0u : (w - w_lo) * P) + (bk >> sub) : 0u;
                    u32 g0 = 0;
                    if (put && lane == gl) g0 = atomicAdd(&pcur[q], gcnt);
                    g0 = (u32)__shfl((int)g0, (int)gl);
                    const u32 tidx = one ? w * g.table_n + i : i;
                    if (put) pairs[g0 + grank] = ((unsigned long long)(bk & submask) << 32) | (tidx | (ng1 ? 0x80000000u : 0u));
                }
                continue;
            }
        }
        for (u32 w0 = 0; w0 < w_hi; w0 += 8) {
            if (__ballot(nonzero && !ds.exhausted()) == 0) break;
            u32 bk[8];
            bool ng8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                bk[j] = 0xffffffffu; ng8[j] = false;
                if (nonzero && w0 + j < w_hi) {
                    ds.next(bk[j], ng8[j]);
                    if (w0 + j < w_lo) bk[j] = 0xffffffffu;
                }
            }
            if (w0 + 8 <= w_lo) continue;                                // (uniform) nothing of this block's windows yet
            // count
            for (u32 q = lane; q < PT; q += 64) lcnt[q] = 0;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (bk[j] != 0xffffffffu) atomicAdd(&lcnt[(one ? 0u : (w0 + j - w_lo) * P) + (bk[j] >> sub)], 1u);
            __builtin_amdgcn_wave_barrier();
            // scan (PT <= 256: up to four entries per lane) and reservation in the partition runs
            u32 run = 0;
            for (u32 q0 = 0; q0 < PT; q0 += 64) {
                const u32 q = q0 + lane;
                const u32 v = q < PT ? lcnt[q] : 0;
                u32 x = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { u32 y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
                if (q < PT) {
                    lbase[q] = run + x - v; lcur[q] = run + x - v;
                    gbase[q] = v ? atomicAdd(&pcur[q], v) : 0u;
                }
                run += __shfl(x, 63);
            }
            __builtin_amdgcn_wave_barrier();
            // place
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (bk[j] != 0xffffffffu) {
                    const u32 q = (one ? 0u : (w0 + j - w_lo) * P) + (bk[j] >> sub);
                    const u32 pos = atomicAdd(&lcur[q], 1u);
                    const u32 tidx = one ? (w0 + j) * g.table_n + i : i;
                    stage[pos] = ((unsigned long long)(bk[j] & submask) << 32) | (tidx | (ng8[j] ? 0x80000000u : 0u));
                    stageq[pos] = (unsigned short)q;
                }
            __builtin_amdgcn_wave_barrier();
            // copy out in staging order
            for (u32 e = lane; e < run; e += 64) {
                const u32 q = stageq[e];
                pairs[gbase[q] + (e - lbase[q])] = stage[e];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// grid (P * ceil(slices / MSM_BUCKET_SLICES), total_groups); 256 threads.  A block walks
// MSM_BUCKET_SLICES consecutive slices of one partition: the sort is stable, so after slice k the
// cursors already stand at slice k + 1's first positions -- no re-initialisation.
#define MSM_BUCKET_SLICES 4
static __global__ __launch_bounds__(256) void k_msm_bucket(u32 nb, u32 c, u32 slices, const u32* off, const u32* bh, const u32* pc,
                                                          const unsigned long long* pairs, u32* idx_out, u32 per_block) {
    msm_tail_prio();
    __shared__ u32 lcur[512], lfirst[512];
    const u32 sub = msm_sub_bits(c), P = nb >> sub, nsub = 1u << sub;      // nsub <= 512
    const u32 chunks = (slices + per_block - 1) / per_block;      // per_block: consecutive slices one block walks (MSM_BUCKET_SLICES, or fewer: run_msm_t)
    const u64 gidx = blockIdx.y;
    u32 p, ch;
    if ((P & 7) == 0) {   // blocks of one partition on one XCD: id = 8 * j + (p mod 8)
        u32 xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        p = xcd + 8 * (j / chunks);
        ch = j % chunks;
    } else {
        p = blockIdx.x / chunks;
        ch = blockIdx.x % chunks;
    }
    const u32 k0 = ch * per_block, k1 = min(k0 + per_block, slices);
    const u32* goff = off + gidx * nb + ((u64)p << sub);
    const u32* rel = bh + (gidx * slices + k0) * nb + ((u64)p << sub);
    for (u32 j = threadIdx.x; j < nsub; j += blockDim.x) { lcur[j] = goff[j] + rel[j]; lfirst[j] = goff[j]; }
    // the run of slices [k0, k1) inside the partition: partition base + prefix over earlier slices
    const u32 pbase = goff[0];
    const u32 beg = pbase + pc[(gidx * slices + k0) * P + p];
    const u32 end = k1 < slices ? pbase + pc[(gidx * slices + k1) * P + p] : goff[nsub];   // off has total + 1 entries
    __syncthreads();
    // Rounds of <= 2048 pairs: grouped by sub-bucket in LDS (count, scan, place), then copied out in
    // staging order -- a store instruction touches ~8 lines (32-byte runs) instead of 64.
    __shared__ u32 cnt[512], sbase[512], scur[512], wsum[4];
    __shared__ u32 stage[2048];
    __shared__ unsigned short stageq[2048];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 r0 = beg; r0 < end; r0 += 2048) {
        const u32 rend = min(r0 + 2048u, end);
        unsigned long long pr[8];
        if (tid < nsub) cnt[tid] = 0;
        if (tid + 256 < nsub) cnt[tid + 256] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u32 i = r0 + tid + j * 256;
            pr[j] = i < rend ? pairs[i] : ~0ull;
            // (a wave whose 64 pairs all go to ONE sub-bucket -- rows of a constant or sorted column -- counts once: 64 atomics on one LDS word otherwise)
            const bool v = pr[j] != ~0ull;
            const u32 sb = (u32)(pr[j] >> 32);
            const unsigned long long act = __ballot(v);
            if (act && __ballot(v && sb != (u32)__builtin_amdgcn_readfirstlane((int)sb)) == 0 && ((act & 1ull) != 0)) {
                if (lane == 0) atomicAdd(&cnt[sb], (u32)__popcll(act));
            } else if (v) atomicAdd(&cnt[sb], 1u);
        }
        __syncthreads();
        if (nsub <= 256) {   // exclusive scan of the (<= 256) counts: one per thread
            u32 v = tid < nsub ? cnt[tid] : 0, x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { u32 y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
            if (lane == 63) wsum[wave] = x;
            __syncthreads();
            u32 before = 0;
            for (u32 w = 0; w < wave; w++) before += wsum[w];
            if (tid < nsub) { sbase[tid] = before + x - v; scur[tid] = before + x - v; }
        } else {             // 512 counts (17-bit windows): two consecutive ones per thread
            const u32 v0 = cnt[2 * tid], v1 = cnt[2 * tid + 1];
            u32 x = v0 + v1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { u32 y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
            if (lane == 63) wsum[wave] = x;
            __syncthreads();
            u32 before = 0;
            for (u32 w = 0; w < wave; w++) before += wsum[w];
            const u32 e0 = before + x - v0 - v1;
            sbase[2 * tid] = e0; scur[2 * tid] = e0;
            sbase[2 * tid + 1] = e0 + v0; scur[2 * tid + 1] = e0 + v0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool v = pr[j] != ~0ull;
            const u32 sb = (u32)(pr[j] >> 32);
            const unsigned long long act = __ballot(v);
            u32 pos = 0;
            if (act && __ballot(v && sb != (u32)__builtin_amdgcn_readfirstlane((int)sb)) == 0 && ((act & 1ull) != 0)) {
                if (lane == 0) pos = atomicAdd(&scur[sb], (u32)__popcll(act));
                pos = (u32)__shfl((int)pos, 0) + (u32)__popcll(act & ((1ull << lane) - 1));
            } else if (v) pos = atomicAdd(&scur[sb], 1u);
            if (v) {
                stage[pos] = (u32)pr[j];
                stageq[pos] = (unsigned short)sb;
            }
        }
        __syncthreads();
        const u32 total = rend - r0;
        for (u32 e = tid; e < total; e += 256) {
            const u32 sb = stageq[e];
            const u32 at = lcur[sb] + (e - sbase[sb]);
            idx_out[at] = stage[e] | (at == lfirst[sb] ? MSM_IDX_FIRST : 0u);     // bit 30: the first point of its bucket (k_msm_accum0 starts a new sum there)
        }
        __syncthreads();
        if (tid < nsub) lcur[tid] += cnt[tid];     // the next round continues where this one ended
        if (tid + 256 < nsub) lcur[tid + 256] += cnt[tid + 256];
    }
}

// largest b in [0, total) with off[b] <= t   (off non-decreasing, off[total] > t): the non-empty segment holding t
FP_DEV u32 find_segment(const u32* off, u32 total, u32 t) {
    u32 lo = 0, hi = total;  // invariant: off[lo] <= t < off[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// table entry -> (affine limbs, is_identity); bit 31 of the reference negates y
template <class F>
FP_DEV aff29 load_point(const affine_t* table, u32 e, bool& is_id) {
    affine_t pk = aff_load(&table[e & MSM_IDX_MASK]);
    is_id = aff_is_identity(pk);
    aff29 q = a29_from_packed(pk);
    if (e >> 31) q.y = f29_sub(f29_zero(), q.y, F::KN);   // 2p - y, limbs < 2^30
    return q;
}

// ---- level 0: one lane = L0 consecutive points of the sorted list -----------------------------
// Every lane gets the same number of points whatever the bucket sizes are (a task-per-bucket
// split leaves 2.1 "rounds" of tasks for 2.0 rounds of lanes at 2^20: measured 10 % of this
// kernel).  A lane whose range crosses a bucket boundary flushes its sum and starts the next
// bucket's; L0 is chosen on the host so that the lanes fill the chip a whole number of times.
template <class CV>
__global__ __launch_bounds__(MSM_ACC_THREADS_MAX) MSM_ACC_WAVES_ATTR void k_msm_accum0(MsmGeom g, u32 total_buckets, const u32* idx, const u32* off, const u32* nrank,
                                                               const affine_t* table, xyzz29_rec* partial, const u32* geo) {
    typedef typename f29_of<typename CV::Base>::type F;
    const u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 M = off[total_buckets];
    const u32 L0 = geo[0];                                      // points per lane, fixed by the scan from the actual M
    const u64 beg64 = (u64)lane * L0;
    if (beg64 >= M) return;
    const u32 beg = (u32)beg64;
    const u32 end = (u32)min((u64)M, beg64 + L0);
    // Sorted entry = table index | bit 31 (negate y) | bit 30 (first point of its bucket, set by k_msm_bucket).  Only non-empty buckets appear in the
    // sorted list, so the record of the next bucket's sum is the next record: a boundary costs one store and no look-up (until round 4 the lane searched
    // `off` for the next non-empty bucket and fetched its record rank: two or three dependent loads in a branch that some lane of the wave takes in
    // a fifth of the iterations at 2^20 and in a third of them at 2^17).
    u32 rec = lane + nrank[find_segment(off, total_buckets, beg)];      // (the bucket holding point `beg`: non-empty)
    // the next point's index is fetched one iteration ahead (dependent idx -> table[idx] chain);
    // measured twice (also after the product-scanning multiplication): prefetching the 64-B point
    // one iteration ahead too changes nothing, although SQ_WAIT_ANY is 24 % of the wave cycles
    u32 e = idx[beg];
    affine_t pk = aff_load(&table[e & MSM_IDX_MASK]);
    bool is_id = aff_is_identity(pk);
    aff29 q = a29_from_packed(pk);
    if (e >> 31) q.y = f29_sub(f29_zero(), q.y, F::KN);
    xyzz29 acc = x29_from_affine<F>(q, is_id);
    if (!is_id) acc.y = f29_norm(acc.y);
    u32 e_next = beg + 1 < end ? idx[beg + 1] : 0;
    for (u32 p = beg + 1; p < end; p++) {
        e = e_next;
        pk = aff_load(&table[e & MSM_IDX_MASK]);
        e_next = p + 1 < end ? idx[p + 1] : 0;
        is_id = aff_is_identity(pk);
        q = a29_from_packed(pk);
        if (e >> 31) q.y = f29_sub(f29_zero(), q.y, F::KN);   // 2p - y, limbs < 2^30
        if (e & MSM_IDX_FIRST) {                              // bucket boundary inside the range
            x29_store(&partial[rec], acc);
            rec++;
            acc = x29_from_affine<F>(q, is_id);
            if (!is_id) acc.y = f29_norm(acc.y);
        } else if (!is_id) acc = x29_add_mixed<F>(acc, q);
    }
    x29_store(&partial[rec], acc);
}

// ---- merging the partial sums of each bucket ---------------------------------------------
// Signed-digit carries of small witness values pile thousands of points into bucket 0 (digit
// +-1), 0/1 columns put half the column into one bucket and a grand-product column that stays
// constant over unused rows repeats one scalar thousands of times, so the number S of partial sums
// per bucket spans 0 .. 10^5.  k_scan_offsets classes the buckets by S while it scans and k_msm_merge2
// (msm_bred.cuh) merges every class in ONE launch with a lane group sized to it.  (The round-3 generation --
// its own classification launch, operands in registers at 172 VGPRs -- is in the history up to round 4.)

// ---- bucket reduction: sum_k (k + 1) * B_k per group: k_msm_bred (msm_bred.cuh).  A group has 2^(c-1) >= 8 buckets: dehalo_bases_register admits windows of 4 .. 16
// bits (upstream's 1- and 3-bit windows for fewer than 32 terms are a CPU economy; the result does not depend on the window).

// An MSM's result leaves as upstream's Jacobian {x, y, z} and / or, for a caller that feeds the transcript (dehalo_msm_device_affine),
// as the affine point: x = X / ZZ, y = Y / ZZZ with one inversion of ZZ * ZZZ -- no second kernel, no detour through Jacobian.
template <class F>
FP_DEV void msm_emit(const xyzz29& p, jacobian_t* out, affine_t* out_affine) {
    if (out) {
        jacobian_t j = x29_to_jacobian_std<F>(p);
        f_store(&out->x, j.x); f_store(&out->y, j.y); f_store(&out->z, j.z);
    }
    if (out_affine) {
        affine_t a;
        if (f29_is_zero_slow<F>(p.zz)) { a.x = f_zero(); a.y = f_zero(); }
        else {
            f29 ti = f29_inv_safegcd<F>(f29_mul<F>(p.zz, p.zzz));
            a.x = f29_to_std<F>(f29_mul<F>(p.x, f29_mul<F>(ti, p.zzz)));
            a.y = f29_to_std<F>(f29_mul<F>(p.y, f29_mul<F>(ti, p.zz)));
        }
        aff_store(out_affine, a);
    }
}

#include "msm_bred.cuh"

// ---- final: window sums -> one Jacobian point per MSM -------------------------------------
// G == 1: convert.  G == W: result = sum_w 2^(c*w) S_w; QUAD w doubles S_w c*w times (a chain of up to c (W - 1) ~ 240 doublings
// whatever the size of the MSM: quad-cooperative, it is 2.4x shorter), then an LDS tree of quad additions (the one-shot,
// unregistered-bases path only).  256 threads = 64 quads >= W.
template <class CV>
__global__ __launch_bounds__(256) void k_msm_final(MsmGeom g, const xyzz29_rec* group_sums, jacobian_t* out, affine_t* out_affine) {
    typedef f29_lat<typename f29_of<typename CV::Base>::type> F;   // short dependent chains at low occupancy: latency schedule
    __shared__ xyzz29_rec sh[64];
    const u32 bat = blockIdx.x;
    const u32 w = threadIdx.x >> 2, role = threadIdx.x & 3;
    if (g.G == 1) {  // precomputed tables: nothing to combine
        if (threadIdx.x == 0) msm_emit<F>(x29_load(&group_sums[bat]), out ? out + bat : nullptr, out_affine ? out_affine + bat : nullptr);
        return;
    }
    xyzz29 s = x29_identity();
    if (w < g.G) s = x29_load(&group_sums[(u64)bat * g.G + w]);
    const u32 nd = w < g.G ? g.c * w : 0;
    for (u32 i = 0; i < nd; i++) s = x29_double_quad<F>(s);        // (uniform within a quad)
    if (role == 0) x29_store(&sh[w], s);
    __syncthreads();
    for (u32 d = 32; d > 0; d >>= 1) {
        if (w < d) {
            xyzz29 x = x29_add_quad<F>(x29_load(&sh[w]), x29_load(&sh[w + d]));
            if (role == 0) x29_store(&sh[w], x);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) msm_emit<F>(x29_load(&sh[0]), out ? out + bat : nullptr, out_affine ? out_affine + bat : nullptr);
}

// ---- SRS table: table[w][i] = [2^(c*w)] P_i, affine, internal canonical packed form ---------
// (one-time per SRS; window rows by repeated doubling on the 32-bit reference arithmetic of
// ec.cuh, one Fermat inversion per row entry)
template <class CV>
__global__ __launch_bounds__(128) void k_msm_build_table(const affine_t* std_points, affine_t* table, u32 n, u32 c, u32 rows) {
    typedef typename CV::Base F;
    typedef typename f29_of<typename CV::Base>::type F9;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_t p = aff_load(&std_points[i]);
    xyzz_t cur = xyzz_from_affine<F>(p);
    for (u32 w = 0; w < rows; w++) {
        affine_t a = p;
        if (w > 0) {
            for (u32 k = 0; k < c; k++) cur = xyzz_double<F>(cur);
            a = xyzz_to_affine<F>(cur);
        }
        affine_t o;
        if (aff_is_identity(a)) { o.x = f_zero(); o.y = f_zero(); }
        else {
            o.x = f29_to_packed_canon<F9>(f29_from_std<F9>(a.x));
            o.y = f29_to_packed_canon<F9>(f29_from_std<F9>(a.y));
        }
        aff_store(&table[(u64)w * n + i], o);
    }
}

// Jacobian -> affine for MSM outputs (dehalo_to_affine), standard form in and out
template <class CV>
__global__ void k_jac_to_affine(const jacobian_t* in, affine_t* out, u32 count) {
    typedef typename CV::Base F;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    fe x = f_load(&in[i].x), y = f_load(&in[i].y), z = f_load(&in[i].z);
    affine_t a;
    if (f_is_zero(z)) { a.x = f_zero(); a.y = f_zero(); }
    else {
        typedef typename f29_of<F>::type F9;
        fe zi = f29_to_std<F9>(f29_inv_safegcd<F9>(f29_from_std<F9>(z)));   // ~6x fewer instructions than the Fermat chain
        fe zi2 = f_sqr<F>(zi);
        a.x = f_mul<F>(x, zi2);
        a.y = f_mul<F>(y, f_mul<F>(zi2, zi));
    }
    aff_store(&out[i], a);
}

// ==========================================================================================
// host driver (instantiated once per curve in msm_<curve>.hip)
// ==========================================================================================

template <class CV>
int run_msm_t(dehalo_ctx* ctx, const dehalo_bases* bases, const fe* d_scalars, size_t len, size_t batch, jacobian_t* d_out, hipStream_t s) {
    typedef typename CV::Scalar FS;
    if (batch == 0) return 0;
    if (len == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, batch * sizeof(jacobian_t), s));
        return 0;
    }
    MsmGeom g;
    g.n = (u32)len; g.table_n = (u32)bases->n; g.c = bases->c; g.W = bases->W; g.nb = 1u << (g.c - 1);
    g.G = bases->precomp ? 1 : g.W;
    g.batch = (u32)batch;
    // sort blocks (k_msm_hist, k_msm_part): msm_sort_block threads, two scalars a thread until the grid has 256 K threads, then longer slices.  1024-thread
    // blocks (the default) take 128 and 115 KiB of LDS and so a compute unit to themselves; 512-thread blocks take 64 KiB (packed 16-bit histogram) and 58 KiB
    // (eight waves' staging) and start beside an NTT tile / merge / reduction block of another context -- which removes the sort's waiting and nothing else:
    // the chip is throughput-bound, the time moves to the kernels the sort now shares a CU with, and twice the per-slice histograms cost 26 us of the lone sort
    // (profiles/r06_sort_block_ab.txt; DESIGN.md sections 4 and 8)
    const u32 sort_threads = ctx->msm_sort_block == 512 ? 512u : 1024u;
    g.slices = (u32)std::min<size_t>(256 * (1024 / sort_threads), std::max<size_t>(1, len / (2 * sort_threads)));
    {   // small precomputed-table launches: 2048 scalars a sort block leave a 2^14 column 8 blocks and a 2^11 column ONE for k_msm_hist / k_msm_part (21 + 34 us of
        // latency where the work is 2); down to 256 scalars a block until ~128 blocks are there (DEHALO_MSM_SMALL_SLICES=0: the A/B)
        static const bool small_slices = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_SMALL_SLICES"); return !(e && e[0] == '0'); }();
        if (small_slices && g.G == 1 && (size_t)g.slices * batch < 128)
            g.slices = (u32)std::max<size_t>(g.slices, std::min<size_t>(std::max<size_t>(1, len / 256), (128 + batch - 1) / batch));
    }
    if ((size_t)g.nb * 4 > 128 * 1024) {      // a 17-bit window (2^16 buckets: precomputed tables only): the histogram fits the LDS only as packed 16-bit counters --
        if (g.G != 1) return dh_fail(ctx, DEHALO_ERR_INVALID, "a 17-bit window needs a precomputed table");
        g.slices = (u32)std::max<uint64_t>(g.slices, ((uint64_t)len * g.W + 65534) / 65535);      // -- so a slice's scalars times the windows stay below 2^16
    }
    {   // single-row tables: windows per sort block -- the block's histograms fit 128 KiB of LDS and its (group, partition) runs the 128 staging lists
        const u32 sb = msm_sub_bits(g.c), Pg = g.nb >> sb;
        const u32 fit = std::min<u32>(std::min<u32>(g.W, 128 / Pg), (128u * 1024 / 4) / g.nb);
        const u32 fill = (u32)((uint64_t)g.W * g.slices * batch / 256);        // ... while the grid keeps >= 256 blocks (2^14: one window per block as before)
        g.wb = g.G == 1 ? g.W : std::max<u32>(1, std::min<u32>(fit, fill));
    }
    const uint64_t total_groups = (uint64_t)batch * g.G;
    const uint64_t total_buckets = total_groups * g.nb;
    const uint64_t Mmax = (uint64_t)batch * len * g.W;
    if (Mmax >= (1ull << 32) || total_buckets >= (1ull << 31))
        return dh_fail(ctx, DEHALO_ERR_INVALID, "batch * len * windows too large for one launch");
    // points per lane: the lanes fill the chip (msm_acc_waves waves per SIMD of k_msm_accum0) a whole number of times.  The value is
    // fixed ON THE DEVICE from the number of points actually sorted (k_scan_offsets); the host only bounds the lane count.
    const uint32_t lcap = (uint32_t)ctx->msm_acc_points;
    const uint64_t resident = (uint64_t)ctx->num_cus * 4 * (lcap ? 1 : (uint64_t)ctx->msm_acc_waves) * 64;
    const uint64_t lmax = 64 * 4 / (uint64_t)ctx->msm_acc_waves;
    g.L0 = 0;
    const uint32_t acc_block = ctx->msm_acc_block == MSM_ACC_THREADS_MAX ? MSM_ACC_THREADS_MAX : MSM_ACC_THREADS;
    uint64_t lanes_max;
    if (lcap) {
        uint64_t k = 4;
        while (Mmax > k * resident * lcap) k += 2;
        lanes_max = k * resident + acc_block;
    } else {
        const uint64_t rounds_max = std::max<uint64_t>(1, (Mmax + resident * lmax - 1) / (resident * lmax));
        lanes_max = rounds_max * resident + acc_block;
    }
    const uint64_t nt0_max = lanes_max + total_buckets;        // records: one per lane + one per non-empty bucket (upper bound)
    const size_t REC = sizeof(xyzz29_rec);

    TRY(dh_ensure(ctx, ctx->ws_count, total_buckets * 4));
    TRY(dh_ensure(ctx, ctx->ws_counters, 64));                                   // 8 merge-class counters | L0 | M
    TRY(dh_ensure(ctx, ctx->ws_bhist, total_buckets * (size_t)g.slices * 6));   // per-block histograms: the prefixes (u32), and behind them the packed 16-bit counts when k_msm_hist writes those
    const u32 sub_bits = msm_sub_bits(g.c), P = g.nb >> sub_bits;
    TRY(dh_ensure(ctx, ctx->ws_pcount, total_groups * (size_t)g.slices * P * 4));   // per-(slice, partition) counts
    TRY(dh_ensure(ctx, ctx->ws_pairs, Mmax * 8));                                   // partition-sorted (sub-bucket, reference) pairs
    TRY(dh_ensure(ctx, ctx->ws_off, (total_buckets + 1) * 4));
    TRY(dh_ensure(ctx, ctx->ws_records, (total_buckets + 1) * 4 * 3));           // nrank | rbeg | rend
    // merge-class lists: `cap` words per class -- a bucket index per listed bucket, or (classes 5 / 6 of k_msm_merge2) 2 words per 512-record part and 4 per heavy bucket
    const u32 merge_cap = (u32)std::max<uint64_t>(std::min<uint64_t>(total_buckets, nt0_max / 2 + 1), 4 * (nt0_max / MERGE2_CHUNK + 2));
    TRY(dh_ensure(ctx, ctx->ws_merge_parts, (2 * (nt0_max / MERGE2_CHUNK) + 4) * sizeof(xyzz29_rec)));   // part sums of the heavy buckets
    TRY(dh_ensure(ctx, ctx->ws_merge_lists, (size_t)merge_cap * MSM_MERGE_COUNTERS * 4));   // merge-class lists
    TRY(dh_ensure(ctx, ctx->ws_idx, Mmax * 4));
    TRY(dh_ensure(ctx, ctx->ws_partial0, nt0_max * REC));
    TRY(dh_ensure(ctx, ctx->ws_buckets, total_buckets * REC));
    // (the radix-2 bucket reduction keeps its node vectors in the same two buffers: BRED_VMAX records per block of 128 / 256 buckets / per cluster of 16 blocks)
    const size_t bred_blocks = std::max<size_t>(1, g.nb / BRED_BLOCK_BUCKETS_MIN);
    TRY(dh_ensure(ctx, ctx->ws_contrib, total_groups * bred_blocks * BRED_VMAX * REC));
    TRY(dh_ensure(ctx, ctx->ws_tree, total_groups * 16 * BRED_VMAX * REC));
    if ((size_t)total_groups * BRED_CNT_PER_GROUP * 4 > ctx->ws_bred_cnt.cap) {      // cluster / group arrival counters: zero when allocated, left zero by every launch
        TRY(dh_ensure(ctx, ctx->ws_bred_cnt, (size_t)total_groups * BRED_CNT_PER_GROUP * 4));
        HIP_TRY(ctx, hipMemsetAsync(ctx->ws_bred_cnt.p, 0, ctx->ws_bred_cnt.cap, s));
    }
    TRY(dh_ensure(ctx, ctx->ws_gsums, total_groups * REC));
    u32* count = (u32*)ctx->ws_count.p;
    u32* cursor = (u32*)ctx->ws_counters.p;
    u32* off = (u32*)ctx->ws_off.p;
    u32* nrank = (u32*)ctx->ws_records.p;
    u32* rbeg = nrank + (total_buckets + 1);
    u32* rend = rbeg + (total_buckets + 1);
    u32* merge_lists = (u32*)ctx->ws_merge_lists.p;
    u32* merge_counters = cursor;
    u32* bh = (u32*)ctx->ws_bhist.p;
    u32* pc = (u32*)ctx->ws_pcount.p;
    unsigned long long* pairs = (unsigned long long*)ctx->ws_pairs.p;
    u32* idx = (u32*)ctx->ws_idx.p;
    xyzz29_rec* partial0 = (xyzz29_rec*)ctx->ws_partial0.p;
    xyzz29_rec* buckets = (xyzz29_rec*)ctx->ws_buckets.p;
    xyzz29_rec* gsums = (xyzz29_rec*)ctx->ws_gsums.p;

    // packed 16-bit counters when no bucket of a block can be counted 2^16 times: every scalar of the slice in every window of the block (all windows for G == 1)
    const u32 per_slice = (u32)((len + g.slices - 1) / g.slices);
    const bool pack16 = g.nb >= 2 && (uint64_t)per_slice * (g.G == 1 ? g.W : g.wb) <= 65535;
    const size_t lds_hist = (size_t)g.nb * (pack16 ? 2 : 4) * (g.G == 1 ? 1 : g.wb);
    const void* hist_fn = pack16 ? (const void*)k_msm_hist<FS, true> : (const void*)k_msm_hist<FS, false>;
    if (lds_hist > 48 * 1024) {
        HIP_TRY(ctx, dh_func_lds(ctx, hist_fn, (int)lds_hist));
    }
    // block sizes of the two scalar-decoding sort kernels (DEHALO_MSM_HIST_THREADS / DEHALO_MSM_PART_THREADS, 64 .. 1024): smaller blocks fit beside a
    // resident accumulation of another context (msm_acc_block = 768 leaves one 128-VGPR wave slot per SIMD: 512 threads x 62 VGPRs, 256 x 77)
    static const u32 hist_threads_env = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_HIST_THREADS"); const int v = e ? atoi(e) : 0; return (u32)std::max(0, std::min(MSM_SORT_THREADS, v & ~63)); }();
    static const u32 part_threads_env = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_PART_THREADS"); const int v = e ? atoi(e) : 0; return (u32)std::max(0, std::min(MSM_SORT_THREADS, v & ~63)); }();
    const u32 hist_threads = hist_threads_env ? hist_threads_env : sort_threads, part_threads = part_threads_env ? part_threads_env : sort_threads;
    const u32 part_lists = (g.G == 1 ? P : 128u) > 128u ? 256u : 128u;      // (G == W: wb windows x Pg partitions <= 128 by the choice of wb above)
    const size_t lds_part = dh_co_lds_pad(0, 4 * (size_t)part_lists + (size_t)(part_threads / 64) * MSM_PART_WAVE_LDS(part_lists));
    HIP_TRY(ctx, dh_func_lds(ctx, (const void*)k_msm_part<FS>, (int)lds_part));
    const u32 tb = (u32)total_buckets;
    {
        ScopedTimer t(ctx, s, DEHALO_K_MSM_SORT);
        dim3 grid(g.slices, g.G == 1 ? 1 : (g.G + g.wb - 1) / g.wb, (u32)batch);
        unsigned short* bh16 = reinterpret_cast<unsigned short*>(bh + total_buckets * (size_t)g.slices);
        if (pack16) k_msm_hist<FS, true><<<grid, hist_threads, lds_hist, s>>>(g, d_scalars, reinterpret_cast<u32*>(bh16), pc);
        else k_msm_hist<FS, false><<<grid, hist_threads, lds_hist, s>>>(g, d_scalars, bh, pc);
        const u32 cs_a = (tb + 255) / 256, cs_b = ((u32)total_groups * P + 255) / 256;
        TRY(dh_ensure(ctx, ctx->ws_bsum, (size_t)cs_a * 2 * sizeof(u32)));
        u32* bs_i = (u32*)ctx->ws_bsum.p;
        u32* bs_t = bs_i + cs_a;
        k_msm_colscan<<<cs_a + cs_b, SCAN_THREADS, 0, s>>>(g.nb, g.slices, tb, bh, pack16 ? bh16 : nullptr, count, cs_a, P, (u32)total_groups * P, pc, bs_i, bs_t, merge_counters);
        k_scan_offsets<<<cs_a, SCAN_THREADS, 0, s>>>(count, tb, bs_i, bs_t, cs_a, cursor + MSM_MERGE_COUNTERS, (u32)resident, (u32)lmax, lcap, (u32)ctx->msm_acc_min_layers, off, nrank, rbeg, rend,
                                                     merge_lists, merge_cap);
        HIP_TRY(ctx, hipGetLastError());
        k_msm_part<FS><<<grid, part_threads, lds_part, s>>>(g, d_scalars, off, pc, pairs, part_lists);
        const size_t lds_bk = dh_co_lds_pad(22 * 1024, 0);
        TRY(dh_co_lds_attr(ctx, (const void*)k_msm_bucket, lds_bk));
        // slices per block: 4 (measured best on dense columns) unless DEHALO_MSM_BUCKET_SLICES says otherwise (1 / 2 / 4 / 8: A/B measurements on the
        // skewed columns of a proof, where a block's run can be 17 windows x 4 slices of ONE value)
        static const u32 bucket_slices = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_BUCKET_SLICES"); const int v = e ? atoi(e) : MSM_BUCKET_SLICES; return (u32)(v >= 1 && v <= 16 ? v : MSM_BUCKET_SLICES); }();
        // ... and fewer while the grid would not give every CU a block (2^14: 16 partitions x 8 slices -- 32 blocks of 10 k pairs each took 49 us a column, round 4)
        u32 bslices = bucket_slices;
        static const bool bucket_fill = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_BUCKET_FILL"); return !(e && e[0] == '0'); }();
        while (bucket_fill && bslices > 1 && (uint64_t)P * ((g.slices + bslices - 1) / bslices) * total_groups < (uint64_t)ctx->num_cus) bslices >>= 1;
        k_msm_bucket<<<dim3(P * ((g.slices + bslices - 1) / bslices), (u32)total_groups), 256, lds_bk, s>>>(g.nb, g.c, g.slices, off, bh, pc, pairs, idx, bslices);
        HIP_TRY(ctx, hipGetLastError());
    }
    {
        ScopedTimer t(ctx, s, DEHALO_K_MSM_ACCUMULATE);
        u32 blocks = (u32)((lanes_max + acc_block - 1) / acc_block);
        // DEHALO_MSM_ACC_LDS (bytes of dynamic LDS per block, unused by the kernel): caps the accumulation's resident blocks per CU so that
        // wave slots and registers stay free for the kernels of other contexts (tuning experiments; results never depend on it)
        static const unsigned acc_lds = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_ACC_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
        k_msm_accum0<CV><<<blocks, acc_block, acc_lds, s>>>(g, tb, idx, off, nrank, bases->table, partial0, cursor + MSM_MERGE_COUNTERS);
        HIP_TRY(ctx, hipGetLastError());
    }
    {
        ScopedTimer t(ctx, s, DEHALO_K_MSM_REDUCE);
        // partial sums -> one point per bucket: every size class in one launch (k_msm_merge2, msm_bred.cuh: operands in LDS, < 128 VGPRs, quads throughout)
        {
            const size_t lds_m = dh_co_lds_pad(18 * 1024, 0);
            TRY(dh_co_lds_attr(ctx, (const void*)k_msm_merge2<CV>, lds_m));
#ifdef DEHALO_EXPERIMENTS
            static const bool merge_stamps = getenv("DEHALO_MSM_MERGE_STAMPS") != nullptr;
            static const int merge_q3 = [] { const char* e = getenv("DEHALO_MSM_MERGE_Q3"); return e ? atoi(e) : 0; }();
            static bool merge_q3_set = false;
            if (merge_q3 && !merge_q3_set) { HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_merge2_q3), &merge_q3, sizeof(int))); merge_q3_set = true; }
            if (merge_stamps) { const int on = 1; HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_merge2_stamps_on), &on, sizeof(on))); }
#endif
            k_msm_merge2<CV><<<MERGE2_GRID, 256, lds_m, s>>>(rbeg, rend, partial0, buckets, merge_counters, merge_lists, merge_cap, (xyzz29_rec*)ctx->ws_merge_parts.p, tb);
#ifdef DEHALO_EXPERIMENTS
            if (merge_stamps) TRY(merge2_report_stamps(ctx, tb, s));
#endif
        }
        // bucket reduction: ONE launch of the radix-2 recursion (msm_bred.cuh: 2 additions per bucket, operands in LDS, < 128 VGPRs; the last block of a group
        // weights, sums and writes the result)
        bool emitted = false;
        if (g.nb < 8) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: window below 4 bits");      // (unreachable through dehalo_bases_register: c >= 4)
        {
            // buckets per block: 128 up to 2^13 buckets (the kernel alone 119 -> 107 us at 4096 buckets, 134 -> 125 at 16384, 153 -> 153 at 32768 where the third level costs
            // what the shorter first one saves); 256 above: measured on k = 17 proofs the 128-bucket blocks -- twice as many, beside the side context's transforms -- cost
            // 0.1 ms (profiles/r04_bred_block_buckets.txt).
            static const u32 bred_bb_env = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_BRED_BLOCK"); const int v = e ? atoi(e) : 0; return v == 128 || v == 256 ? (u32)v : 0u; }();
            const u32 bred_bb = bred_bb_env ? bred_bb_env : (g.nb <= 8192 ? 128u : 256u);
            const u32 nblk = std::max<u32>(1, g.nb / bred_bb);
            const bool fin = g.G == 1;
            const size_t lds_b = dh_co_lds_pad(41 * 1024, 0);
            TRY(dh_co_lds_attr(ctx, (const void*)k_msm_bred<CV>, lds_b));
#ifdef DEHALO_EXPERIMENTS
            static const bool bred_stamps = getenv("DEHALO_MSM_BRED_STAMPS") != nullptr;
            if (bred_stamps) {
                const int on = 1; unsigned long long init[12] = {0, 0, 0, 0, 0, 0, 0, ~0ull, 0, 0, 0, 0};
                HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_bred_stamps_on), &on, sizeof(on)));
                HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_bred_stamps), init, sizeof(init)));
            }
#endif
            k_msm_bred<CV><<<dim3(nblk, (u32)total_groups), BRED_THREADS, lds_b, s>>>(g.nb, bred_bb, buckets, (xyzz29_rec*)ctx->ws_contrib.p, (xyzz29_rec*)ctx->ws_tree.p, (u32*)ctx->ws_bred_cnt.p, gsums,
                                                                                 fin ? d_out : nullptr, fin ? ctx->msm_affine_out : nullptr);
            emitted = fin;
#ifdef DEHALO_EXPERIMENTS
            if (bred_stamps) {
                unsigned long long st[12];
                HIP_TRY(ctx, hipStreamSynchronize(s));
                HIP_TRY(ctx, hipMemcpyFromSymbol(st, HIP_SYMBOL(g_bred_stamps), sizeof(st)));
                auto us = [&](int i) { return (double)(st[i] - st[7]) / 100.0; };
                fprintf(stderr, "k_msm_bred nb %u groups %u, us after the first block's start (the block that finishes group 0): its start %.1f | phase 0 tree done %.1f | phase 1 %.1f | phase 2 %.1f | "
                        "doublings done %.1f | final tree %.1f | result written %.1f || hand-offs: phase 1 last arrival known %.1f, siblings in LDS %.1f | phase 2 %.1f, %.1f\n", g.nb, (unsigned)total_groups,
                        us(0), us(1), st[2] ? us(2) : 0.0, st[3] ? us(3) : 0.0, us(5), us(4), us(6), st[8] ? us(8) : 0.0, st[9] ? us(9) : 0.0, st[10] ? us(10) : 0.0, st[11] ? us(11) : 0.0);
            }
#endif
        }
        if (!emitted) {
            k_msm_final<CV><<<(u32)batch, 256, 0, s>>>(g, gsums, d_out, ctx->msm_affine_out);
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    return 0;
}

template <class CV>
int build_table_t(dehalo_ctx* ctx, dehalo_bases* b, const affine_t* d_std_points, hipStream_t s) {
    u32 rows = b->precomp ? b->W : 1;
    k_msm_build_table<CV><<<(u32)((b->n + 127) / 128), 128, 0, s>>>(d_std_points, b->table, (u32)b->n, b->c, rows);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// sum of `count` Jacobian points (standard form) -> one Jacobian point: the combine step of an MSM that
// was split by point range over several GPUs: one wave, strided lane sums, shuffle tree
template <class CV>
__global__ __launch_bounds__(64) void k_point_sum(const jacobian_t* in, u32 count, jacobian_t* out) {
    typedef f29_lat<typename f29_of<typename CV::Base>::type> F;
    const u32 lane = threadIdx.x;
    xyzz29 acc = x29_identity();
    for (u32 i = lane; i < count; i += 64) {
        fe z = f_load(&in[i].z);
        if (f_is_zero(z)) continue;
        xyzz29 p;
        f29 z29 = f29_from_std<F>(z);
        p.x = f29_from_std<F>(f_load(&in[i].x));
        p.y = f29_from_std<F>(f_load(&in[i].y));
        p.zz = f29_sqr<F>(z29);                    // Jacobian (X, Y, Z) is XYZZ (X, Y, Z^2, Z^3)
        p.zzz = f29_mul<F>(p.zz, z29);
        acc = x29_add<F>(acc, p);
    }
    acc = x29_group_reduce<F, 64>(acc);
    if (lane == 0) {
        jacobian_t j = x29_to_jacobian_std<F>(acc);
        f_store(&out->x, j.x); f_store(&out->y, j.y); f_store(&out->z, j.z);
    }
}

template <class CV>
int point_sum_t(dehalo_ctx* ctx, const jacobian_t* d_in, uint32_t count, jacobian_t* d_out, hipStream_t s) {
    k_point_sum<CV><<<1, 64, 0, s>>>(d_in, count, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class CV>
int to_affine_t(dehalo_ctx* ctx, const jacobian_t* d_in, affine_t* d_out, uint32_t count, hipStream_t s) {
    k_jac_to_affine<CV><<<(count + 63) / 64, 64, 0, s>>>(d_in, d_out, count);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

#define DEFINE_MSM_ENTRY(NAME, CV)                                                                                                        \
    int run_msm_##NAME(dehalo_ctx* ctx, const dehalo_bases* bases, const fe* d_scalars, size_t len, size_t batch, jacobian_t* d_out,      \
                       hipStream_t s) { return run_msm_t<CV>(ctx, bases, d_scalars, len, batch, d_out, s); }                              \
    int build_table_##NAME(dehalo_ctx* ctx, dehalo_bases* b, const affine_t* d_std_points, hipStream_t s) {                               \
        return build_table_t<CV>(ctx, b, d_std_points, s); }                                                                              \
    int to_affine_##NAME(dehalo_ctx* ctx, const jacobian_t* d_in, affine_t* d_out, uint32_t count, hipStream_t s) {                       \
        return to_affine_t<CV>(ctx, d_in, d_out, count, s); }                                                                             \
    int point_sum_##NAME(dehalo_ctx* ctx, const jacobian_t* d_in, uint32_t count, jacobian_t* d_out, hipStream_t s) {                     \
        return point_sum_t<CV>(ctx, d_in, count, d_out, s); }
