// lookup_permute.hip -- the lookup argument's permuted (input, table) pair on the device
// (SURVEY.md 8(f) row 2).  Replaces [UPSTREAM halo2_proofs/src/plonk/lookup/prover.rs
// permute_expression_pair @ v2023_04_20]:
//
//   A' = the first `usable` input values, sorted ascending (Ord on the canonical integer);
//   S'[row] = A'[row] wherever A'[row] differs from A'[row-1] (first occurrences), each taking one
//             copy of that value out of the table's multiset -- an input value that is not in the
//             table is upstream's Error::ConstraintSystemFailure;
//   the remaining table values, ascending, fill the repeated rows taken from the END
//             (upstream pops `repeated_input_rows`), so the smallest leftover lands in the highest
//             repeated row.
// The blinding rows upstream appends are the caller's (they are random).
//
// Upstream: Vec::sort + BTreeMap on one thread.  Here only the TABLE is sorted -- once per distinct table column of a call (the five
// range lookups of the delay-encryption circuit share one) -- by a merge sort on the full 256-bit keys (LDS bitonic tiles, then
// merge-path passes: no digit plan, no fallback, nothing read back).  The inputs are never sorted: every input value finds its table
// position by binary search (a miss is the error), positions are counted, and A' / S' are WRITTEN from the counts -- A' is the sorted
// table with every entry repeated as often as it was looked up, S' the first occurrences plus the unused entries from the end.
// Everything is hand-written here (round 3 replaced the rocPRIM radix sort and scans this file used to call).
//
// Tables of fixed columns (round 4): when every table expression of a lookup reads fixed columns only, WHICH rows of the table are equal does not depend on
// theta -- only the order of the compressed values does.  The prover hands over, per table, one representative row of every distinct tuple and its
// multiplicity (computed once per proving key: the range tables of the delay-encryption circuit have 339 distinct rows among 131,066).  Then only the
// distinct keys are sorted -- one LDS tile instead of 64 tiles and six merge passes --, the full sorted table is written out from (key, multiplicity) and the
// inputs search the distinct keys.  Same A' and S' (tests/test_gpu_parity.py compares both paths with the CPU restatement).
#include <cstring>
#include <string>
#include <vector>

#include "fp.cuh"
#include "field_constants.h"
#include "internal.hpp"

namespace {

constexpr u32 LP_TILE = 2048, LP_THREADS = 256, LP_PER = LP_TILE / LP_THREADS;      // 8 keys per thread (scan / emit kernels)
constexpr u32 LP_SORT_THREADS = 1024;                                               // tile sort, rank: one compare-exchange per thread per stage
constexpr u32 LP_MERGE_OUT = 512, LP_MERGE_THREADS = 128, LP_MERGE_PER = LP_MERGE_OUT / LP_MERGE_THREADS;      // a merge block's outputs
constexpr u32 LP_MAX_BATCH = 16;

struct LpCols {      // per lookup: its input column, its outputs, which sorted table it uses; per distinct table: the column
    const fe* in[LP_MAX_BATCH];
    fe* out_in[LP_MAX_BATCH];
    fe* out_tab[LP_MAX_BATCH];
    const fe* tab[LP_MAX_BATCH];      // distinct tables
    u32 table_of[LP_MAX_BATCH];
    const u32* rep[LP_MAX_BATCH];     // per distinct table, or null: a representative row of every distinct tuple (< LP_TILE of them) ...
    const u32* mult[LP_MAX_BATCH];    // ... and how many of the usable rows hold it
    u32 ndist[LP_MAX_BATCH];
};

FP_DEV int lp_cmp(const fe& a, const fe& b) {          // canonical integers, most significant word first
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
    }
    return 0;
}
// keys in LDS as eight planes of LP_TILE words (word w of key i at lds[w * LP_TILE + i]): conflict-free for consecutive i
FP_DEV fe lds_key(const u32* lds, u32 i) {
    fe r;
#pragma unroll
    for (int w = 0; w < 8; w++) r.v[w] = lds[w * LP_TILE + i];
    return r;
}
FP_DEV void lds_put(u32* lds, u32 i, const fe& k) {
#pragma unroll
    for (int w = 0; w < 8; w++) lds[w * LP_TILE + i] = k.v[w];
}
FP_DEV bool lds_less(const u32* lds, u32 i, u32 j) {      // key[i] < key[j]
    for (int w = 7; w >= 0; w--) {
        const u32 a = lds[w * LP_TILE + i], b = lds[w * LP_TILE + j];
        if (a != b) return a < b;
    }
    return false;
}

// canonical keys of the distinct tables, padded to whole tiles with +infinity (all ones: above every canonical value)
template <class F>
__global__ void k_lp_canon_tables(LpCols c, u64 n, u64 npad, fe* keys) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = blockIdx.y;
    if (i >= npad) return;
    fe k;
    if (i < n) k = f_from_mont<F>(f_load(&c.tab[t][i]));
    else {
#pragma unroll
        for (int w = 0; w < 8; w++) k.v[w] = 0xFFFFFFFFu;
    }
    f_store(&keys[(u64)t * npad + i], k);
}

// one tile of LP_TILE keys sorted in LDS (bitonic network on the full keys)
__global__ void __launch_bounds__(LP_SORT_THREADS) k_lp_tile_sort(const fe* in, fe* out, u64 npad) {
    extern __shared__ u32 lds[];
    const u64 base = (u64)blockIdx.y * npad + (u64)blockIdx.x * LP_TILE;
    for (u32 i = threadIdx.x; i < LP_TILE; i += LP_SORT_THREADS) lds_put(lds, i, f_load(&in[base + i]));
    __syncthreads();
    const u32 t = threadIdx.x;
    for (u32 k = 2; k <= LP_TILE; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            // both operands are loaded in full before they are compared (an early-exit compare on LDS words is a chain of eight dependent
            // round trips when the keys are equal, and table columns are mostly equal keys)
            const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // 2 j (t / j) + t % j, j a power of two
            const fe a = lds_key(lds, i), b = lds_key(lds, i + j);
            const int cmp = lp_cmp(a, b);
            if (((i & k) == 0) ? cmp > 0 : cmp < 0) {
                lds_put(lds, i, b);
                lds_put(lds, i + j, a);
            }
            __syncthreads();
        }
    }
    for (u32 i = threadIdx.x; i < LP_TILE; i += LP_SORT_THREADS) f_store(&out[base + i], lds_key(lds, i));
}

// merge path: the number of elements taken from A among the first d of merge(A, B) (A first on ties).  One wave searches 64 candidates
// per round (three rounds cover 2^18 positions) instead of a lane walking a 17-step dependent chain of global loads.
__device__ u64 lp_merge_path_wave(const fe* A, u64 la, const fe* B, u64 lb, u64 d) {
    u64 lo = d > lb ? d - lb : 0, hi = d < la ? d : la;      // the answer lies in [lo, hi]
    const u32 lane = threadIdx.x & 63;
    while (lo < hi) {
        const u64 span = hi - lo;                            // candidates mid in [lo, hi): "take A[mid]" iff A[mid] <= B[d - 1 - mid]
        const u64 step = (span + 63) / 64;
        const u64 mid = lo + (u64)lane * step;
        bool take = false;
        if (mid < hi) take = lp_cmp(f_load(&A[mid]), f_load(&B[d - 1 - mid])) <= 0;
        const u64 mask = __ballot(take);                     // monotone: true ... true false ... false
        const u32 cnt = (u32)__popcll(mask);
        // the answer is > the last true candidate and <= the first false one
        const u64 new_lo = cnt ? lo + (u64)(cnt - 1) * step + 1 : lo;
        const u64 first_false = lo + (u64)cnt * step;
        const u64 new_hi = first_false < hi ? first_false : hi;
        lo = new_lo;
        hi = new_hi;
    }
    return lo;
}

// runs of `run` keys (sorted) -> runs of 2 * run: every block writes LP_MERGE_OUT consecutive outputs of one pair of runs (small blocks:
// a column of 2^17 keys is 256 of them, one per compute unit)
__global__ void __launch_bounds__(LP_MERGE_THREADS) k_lp_merge(const fe* in, fe* out, u64 npad, u64 run) {
    __shared__ u32 lds[8 * LP_MERGE_OUT];                    // LP_MERGE_OUT keys, eight planes
    __shared__ u64 part[2];
    constexpr u32 PITCH = LP_MERGE_OUT;
    const u64 col = (u64)blockIdx.y * npad;
    const u64 o = (u64)blockIdx.x * LP_MERGE_OUT;            // first output of this block, within the column
    const u64 pair = o / (2 * run), within = o % (2 * run);
    const u64 a0 = pair * 2 * run;
    const u64 la = run < npad - a0 ? run : npad - a0;
    const u64 b0 = a0 + la;
    const u64 lb = b0 < npad ? (run < npad - b0 ? run : npad - b0) : 0;
    const fe* A = in + col + a0;
    const fe* B = in + col + b0;
    const u64 d0 = within, d1 = (within + LP_MERGE_OUT < la + lb) ? within + LP_MERGE_OUT : la + lb;
    const u32 wave = threadIdx.x >> 6;
    {
        const u64 r = lp_merge_path_wave(A, la, B, lb, wave == 0 ? d0 : d1);
        if ((threadIdx.x & 63) == 0) part[wave] = r;
    }
    __syncthreads();
    const u64 ai0 = part[0], ai1 = part[1], bi0 = d0 - ai0, bi1 = d1 - ai1;
    const u32 na = (u32)(ai1 - ai0), nb = (u32)(bi1 - bi0);   // na + nb <= LP_MERGE_OUT
    for (u32 i = threadIdx.x; i < na + nb; i += LP_MERGE_THREADS) {
        const fe k = i < na ? f_load(&A[ai0 + i]) : f_load(&B[bi0 + (i - na)]);
#pragma unroll
        for (int w = 0; w < 8; w++) lds[w * PITCH + i] = k.v[w];
    }
    __syncthreads();
    auto key = [&](u32 i) {
        fe r;
#pragma unroll
        for (int w = 0; w < 8; w++) r.v[w] = lds[w * PITCH + i];
        return r;
    };
    // every thread merges LP_MERGE_PER outputs starting at its own diagonal
    const u32 total = na + nb;
    const u32 d = threadIdx.x * LP_MERGE_PER < total ? threadIdx.x * LP_MERGE_PER : total;
    u32 lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (lp_cmp(key(mid), key(na + (d - 1 - mid))) <= 0) lo = mid + 1;      // A[mid] <= B[d - 1 - mid]
        else hi = mid;
    }
    u32 ai = lo, bi = d - lo;
    fe ka, kb;
    bool ha = ai < na, hb = bi < nb;
    if (ha) ka = key(ai);
    if (hb) kb = key(na + bi);
    for (u32 k = 0; k < LP_MERGE_PER && d + k < total; k++) {
        const bool from_a = ha && (!hb || lp_cmp(ka, kb) <= 0);
        f_store(&out[col + o + d + k], from_a ? ka : kb);
        if (from_a) {
            ai++;
            ha = ai < na;
            if (ha) ka = key(ai);
        } else {
            bi++;
            hb = bi < nb;
            if (hb) kb = key(na + bi);
        }
    }
}

// ---- tables given as distinct rows + multiplicities -------------------------------------------------------------------------------------
// inclusive scan over the 1024 threads of a block: shuffles inside a wave, one LDS round trip across the 16 waves (a Hillis-Steele scan in LDS is ten steps of
// two barriers each: 50 us for a kernel whose whole work is this scan)
FP_DEV u32 lp_block_scan_1024(u32 v, u32* wave_sums /* 16 words of LDS */) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 x = __shfl_up(v, d); if ((int)lane >= d) v += x; }
    if (lane == 63) wave_sums[wave] = v;
    __syncthreads();
    u32 before = 0;
    for (u32 w = 0; w < wave; w++) before += wave_sums[w];
    __syncthreads();
    return v + before;
}
// up to 512 keys sorted by counting, for every key, the keys below it (ties by position): one pass instead of the bitonic network's 66 stages of two barriers
// each.  Only the table's `ndist` keys take part (the +infinity padding behind them stays where it is).  NO LDS: key j is the same address
// for every lane, so the loop reads the keys through the scalar cache -- a block that needs 16 KB of LDS (let alone the bitonic tile's 64 KB) waits until a CU
// whose LDS is held by two NTT tiles of the side context gives one up: 160-270 us in a proof for 5 us of work.  (Nor do the top 64 bits alone order the keys:
// the compressed values theta * tag + value of one tag differ in their LOW bits only.)
// (Round 5: one 512-thread block per table did this in 112-125 us -- every thread walked all `len` keys, eight waves on one CU, 4.8 % of a k = 17 proof's lookup
// phase for 339 keys.  Now a QUAD per key, each lane counting over a quarter of the keys, on one-wave blocks spread over the chip: 16 keys a block.)
#define LP_SS_THREADS 64
__global__ void __launch_bounds__(LP_SS_THREADS) k_lp_small_sort(LpCols c, const fe* in, fe* out) {
    const u32 tb = blockIdx.y, len = c.ndist[tb];
    const fe* K = in + (u64)tb * LP_TILE;
    fe* O = out + (u64)tb * LP_TILE;
    const u32 gt = blockIdx.x * LP_SS_THREADS + threadIdx.x;
    for (u32 i = len + gt; i < LP_TILE; i += gridDim.x * LP_SS_THREADS) f_store(&O[i], f_load(&K[i]));      // the +infinity padding stays where it is
    const u32 t = gt >> 2, part = gt & 3;
    const bool live = t < len;                            // (a whole quad is live or not: its lanes stay together for the shuffles below)
    const fe mine = f_load(&K[live ? t : 0]);
    const u32 q = (len + 3) >> 2, j_beg = part * q, j_end = min(len, j_beg + q);
    u32 rank = 0;
    for (u32 j0 = j_beg; j0 < j_end; j0 += 4) {           // four keys' loads in flight at a time
        fe cur[4];
#pragma unroll
        for (int x = 0; x < 4; x++) cur[x] = f_load(&K[j0 + x < j_end ? j0 + x : j_end - 1]);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            int cmp = 0;                                  // all eight words, no early exit (an early-exit compare turns into eight dependent loads)
#pragma unroll
            for (int w = 7; w >= 0; w--) {
                const int d = cur[x].v[w] < mine.v[w] ? -1 : (cur[x].v[w] > mine.v[w] ? 1 : 0);
                cmp = cmp ? cmp : d;
            }
            if (j0 + x < j_end) rank += cmp < 0 || (cmp == 0 && j0 + x < t);      // equal keys: by position
        }
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    if (live && part == 0) f_store(&O[rank], mine);
}
// canonical keys of the distinct rows of table t, one tile per table, padded with +infinity
template <class F>
__global__ void k_lp_distinct_keys(LpCols c, u64 n, fe* keys) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (i >= LP_TILE) return;
    fe k;
    if (i < c.ndist[t]) k = f_from_mont<F>(f_load(&c.tab[t][min((u64)c.rep[t][i], n - 1)]));      // (a row index is the caller's: never read past the column)
    else {
#pragma unroll
        for (int w = 0; w < 8; w++) k.v[w] = 0xFFFFFFFFu;
    }
    f_store(&keys[(u64)t * LP_TILE + i], k);
}
FP_DEV u32 lp_lower_bound_tile(const fe* T, const fe& a) {      // first index in the sorted tile with T[index] >= a (the +infinity padding is above every value)
    u32 lo = 0;
    for (u32 len = LP_TILE; len > 1;) {
        const u32 half = len >> 1;
        if (lp_cmp(f_load(&T[lo + half - 1]), a) < 0) lo += half;
        len -= half;
    }
    if (lp_cmp(f_load(&T[lo]), a) < 0) lo++;
    return lo;
}
// one block per table: the multiplicity of every SORTED distinct key (a tuple finds its key's first position; tuples whose compressed values coincide add up
// there, as equal values of the multiset do) and its exclusive scan = the key's first position in the full sorted table: dstart[t][0 .. LP_TILE]
template <class F>
__global__ void __launch_bounds__(1024) k_lp_distinct_starts(LpCols c, u64 n, const fe* dsorted, u32* dstart) {
    __shared__ u32 sm[LP_TILE], part[16];
    const u32 t = blockIdx.x, tid = threadIdx.x;
    const fe* T = dsorted + (u64)t * LP_TILE;
    for (u32 i = tid; i < LP_TILE; i += 1024) sm[i] = 0;
    __syncthreads();
    for (u32 d = tid; d < c.ndist[t]; d += 1024) {
        const fe key = f_from_mont<F>(f_load(&c.tab[t][min((u64)c.rep[t][d], n - 1)]));
        atomicAdd(&sm[min(lp_lower_bound_tile(T, key), LP_TILE - 1)], c.mult[t][d]);
    }
    __syncthreads();
    constexpr u32 PER = LP_TILE / 1024;
    u32 v[PER], sum = 0;
    for (u32 k = 0; k < PER; k++) { v[k] = sm[tid * PER + k]; sum += v[k]; }
    u32 run = lp_block_scan_1024(sum, part) - sum;
    for (u32 k = 0; k < PER; k++) { dstart[(u64)t * (LP_TILE + 1) + tid * PER + k] = run; run += v[k]; }
    if (tid == 1023) dstart[(u64)t * (LP_TILE + 1) + LP_TILE] = run;
}
// Every input value's distinct key: counted, cnt[y][position among the sorted distinct keys].  The block keeps the table's sorted distinct keys in LDS as eight
// planes of dlen words (16 KB for up to 512 keys: nine search steps on LDS words instead of seventeen 32-byte loads from a 4 MB table) and a histogram of its
// 2048 inputs; a wave whose 64 inputs are one value (the zero rows of an unused region) counts once.
template <class F>
__global__ void __launch_bounds__(LP_SORT_THREADS) k_lp_rank_distinct(LpCols c, const fe* dsorted, u64 n, u32* cnt, int* err, u32 dlen) {      // dlen: a power of two >= every table's distinct rows
    extern __shared__ u32 lds[];                      // [8 planes of dlen key words | dlen counters]
    u32* hist = lds + 8 * dlen;
    constexpr u32 RP = LP_TILE / LP_SORT_THREADS;     // 2 inputs per thread
    const u32 y = blockIdx.y, t = c.table_of[y], nd = c.ndist[t];
    const fe* T = dsorted + (u64)t * LP_TILE;
    for (u32 i = threadIdx.x; i < dlen; i += LP_SORT_THREADS) {
        const fe k = f_load(&T[i]);
#pragma unroll
        for (int w = 0; w < 8; w++) lds[w * dlen + i] = k.v[w];
        hist[i] = 0;
    }
    __syncthreads();
    auto cmp_at = [&](u32 m, const fe& a) {           // key[m] against a, words from the top
        int cmp = 0;
#pragma unroll
        for (int w = 7; w >= 0; w--) {
            const u32 x = lds[w * dlen + m];
            if (cmp == 0 && x != a.v[w]) cmp = x < a.v[w] ? -1 : 1;
        }
        return cmp;
    };
    const u64 base = (u64)blockIdx.x * LP_TILE;
    bool miss = false;
    for (u32 k = 0; k < RP; k++) {
        const u64 i = base + threadIdx.x + k * LP_SORT_THREADS;
        u32 pos = 0xFFFFFFFFu;
        if (i < n) {
            const fe a = f_from_mont<F>(f_load(&c.in[y][i]));
            u32 l = 0;
            for (u32 len = dlen; len > 1;) {          // lower bound among the first dlen keys of the tile (the +infinity padding is above every value)
                const u32 half = len >> 1;
                if (cmp_at(l + half - 1, a) < 0) l += half;
                len -= half;
            }
            if (cmp_at(l, a) < 0) l++;
            if (l < nd && l < dlen && cmp_at(l, a) == 0) pos = l; else miss = true;
        }
        const bool count = pos != 0xFFFFFFFFu;
        const unsigned long long act = __ballot(count);
        if (act && __ballot(count && pos != (u32)__builtin_amdgcn_readfirstlane((int)pos)) == 0 && (act & 1ull)) {
            if ((threadIdx.x & 63) == 0) atomicAdd(&hist[pos], (u32)__popcll(act));
        } else if (count) atomicAdd(&hist[pos], 1u);
    }
    if (miss) atomicExch(&err[y], 1);
    __syncthreads();
    for (u32 i = threadIdx.x; i < nd; i += LP_SORT_THREADS)
        if (hist[i]) atomicAdd(&cnt[(u64)y * LP_TILE + i], hist[i]);
}
// one block per lookup, over the <= LP_TILE sorted distinct keys p of its table: exclusive scans of the looked-up counts (-> rowstart[p], the first output row
// of key p), of "looked up at all" (-> usedbefore[p]: first-occurrence rows before key p's) and of the leftover copies  multiplicity - looked up at all
// (-> leftstart[p]: the key's rank among the leftovers, ascending).  Each array has LP_TILE + 1 entries (the last: the total).
__global__ void __launch_bounds__(1024) k_lp_scan_distinct(LpCols c, const u32* cnt, const u32* dstart, u32* rowstart, u32* usedbefore, u32* leftstart) {
    __shared__ u32 part[3][16];
    constexpr u32 PER = LP_TILE / 1024;
    const u32 y = blockIdx.x, t = c.table_of[y], tid = threadIdx.x;
    u32 v[3][PER], sum[3] = {0, 0, 0};
    for (u32 k = 0; k < PER; k++) {
        const u32 p = tid * PER + k;
        const u32 cn = cnt[(u64)y * LP_TILE + p], mult = dstart[(u64)t * (LP_TILE + 1) + p + 1] - dstart[(u64)t * (LP_TILE + 1) + p];
        v[0][k] = cn; v[1][k] = cn ? 1u : 0u; v[2][k] = mult - (cn ? 1u : 0u);
        for (int a = 0; a < 3; a++) sum[a] += v[a][k];
    }
    u32 incl[3];
    for (int a = 0; a < 3; a++) incl[a] = lp_block_scan_1024(sum[a], part[a]);
    u32* out[3] = {rowstart + (u64)y * (LP_TILE + 1), usedbefore + (u64)y * (LP_TILE + 1), leftstart + (u64)y * (LP_TILE + 1)};
    for (int a = 0; a < 3; a++) {
        u32 run = incl[a] - sum[a];
        for (u32 k = 0; k < PER; k++) { out[a][tid * PER + k] = run; run += v[a][k]; }
        if (tid == 1023) out[a][LP_TILE] = run;
    }
}
FP_DEV u32 lp_last_le_lds(const u32* a, u32 x) {      // the last index in [0, LP_TILE) with a[index] <= x (a non-decreasing, a[0] = 0)
    u32 lo = 0, hi = LP_TILE;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}
// output row i of lookup y: A'[i] = the key p whose rows [rowstart[p], rowstart[p + 1]) hold i; on the key's first row S' = A'; otherwise the row is the r-th
// repeated one, r = i - (first-occurrence rows up to i), and takes the (m - 1 - r)-th leftover, ascending (m = leftovers = repeated rows).
template <class F>
__global__ void __launch_bounds__(256) k_lp_emit_distinct(LpCols c, const fe* dsorted, u64 n, const u32* rowstart, const u32* usedbefore, const u32* leftstart, const int* err) {
    __shared__ u32 rs[LP_TILE + 1], ls[LP_TILE + 1];
    const u32 y = blockIdx.y;
    if (err[y]) return;                                   // a missing input value: the call fails, nothing meaningful to write
    for (u32 i = threadIdx.x; i <= LP_TILE; i += 256) { rs[i] = rowstart[(u64)y * (LP_TILE + 1) + i]; ls[i] = leftstart[(u64)y * (LP_TILE + 1) + i]; }
    __syncthreads();
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const fe* T = dsorted + (u64)c.table_of[y] * LP_TILE;
    const u32 p = lp_last_le_lds(rs, (u32)i);
    const fe a = f_to_mont<F>(f_load(&T[p]));
    f_store(&c.out_in[y][i], a);
    if (rs[p] == (u32)i) {
        f_store(&c.out_tab[y][i], a);
        return;
    }
    const u32 r = (u32)i - (usedbefore[(u64)y * (LP_TILE + 1) + p] + 1), m = ls[LP_TILE];
    f_store(&c.out_tab[y][i], f_to_mont<F>(f_load(&T[lp_last_le_lds(ls, m - 1 - r)])));
}

// Every input value's position in its sorted table (first copy): counted.  A tile's positions are sorted in LDS and run-length encoded,
// so that a value looked up thousands of times (zero, in the unused rows) costs one atomic per tile instead of one per row.
template <class F>
__global__ void __launch_bounds__(LP_SORT_THREADS) k_lp_rank(LpCols c, const fe* sorted, u64 n, u64 npad, u32* cnt, int* err) {
    __shared__ u32 pos[LP_TILE];
    constexpr u32 RP = LP_TILE / LP_SORT_THREADS;      // 2 inputs per thread
    const u32 y = blockIdx.y;
    const fe* T = sorted + (u64)c.table_of[y] * npad;
    const u64 base = (u64)blockIdx.x * LP_TILE;
    bool miss = false;
    {
        // the thread's lower-bound searches advance together: each step issues all its 32-byte loads before comparing
        fe a[RP];
        u64 lo[RP];
#pragma unroll
        for (u32 k = 0; k < RP; k++) {
            const u64 i = base + threadIdx.x + k * LP_SORT_THREADS;
            a[k] = f_from_mont<F>(f_load(&c.in[y][i < n ? i : n - 1]));
            lo[k] = 0;
        }
        for (u64 len = n; len > 1;) {                              // invariant: the lower bound lies in [lo, lo + len]
            const u64 half = len >> 1;
            fe t[RP];
#pragma unroll
            for (u32 k = 0; k < RP; k++) t[k] = f_load(&T[lo[k] + half - 1]);
#pragma unroll
            for (u32 k = 0; k < RP; k++)
                if (lp_cmp(t[k], a[k]) < 0) lo[k] += half;
            len -= half;
        }
#pragma unroll
        for (u32 k = 0; k < RP; k++) {
            const u32 li = threadIdx.x + k * LP_SORT_THREADS;
            const u64 i = base + li;
            u32 p = 0xFFFFFFFFu;
            if (i < n) {
                u64 l = lo[k];
                if (lp_cmp(f_load(&T[l]), a[k]) < 0) l++;          // the one remaining candidate
                if (l < n && lp_cmp(f_load(&T[l]), a[k]) == 0) p = (u32)l;
                else miss = true;
            }
            pos[li] = p;
        }
    }
    if (miss) atomicExch(&err[y], 1);
    __syncthreads();
    for (u32 k = 2; k <= LP_TILE; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            const u32 t = threadIdx.x;
            const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i + j;
            const u32 a = pos[i], b = pos[x];
            if (((i & k) == 0) ? (b < a) : (a < b)) {
                pos[i] = b;
                pos[x] = a;
            }
            __syncthreads();
        }
    }
    for (u32 k = 0; k < RP; k++) {
        const u32 li = threadIdx.x * RP + k;
        const u32 v = pos[li];
        if (v == 0xFFFFFFFFu || (li > 0 && pos[li - 1] == v)) continue;      // padding / misses sort to the end; only run heads count
        u32 lo = li + 1, hi = LP_TILE;                                       // end of the run: first position with a larger value
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (pos[mid] <= v) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&cnt[(u64)y * n + v], lo - li);
    }
}

// exclusive scans over the n table positions of every lookup, of cnt (-> the first output row of each table entry) and of "never looked
// up" flags (-> the entry's rank among the leftovers), written COMPACTED: used entries (first row, position) and leftover positions;
// three launches: tile sums, their scan, apply
__global__ void __launch_bounds__(LP_THREADS) k_lp_scan_tiles(const u32* cnt, u64 n, u32 tiles, u32* tile_sums) {
    __shared__ u32 red[2][LP_THREADS / 64];
    const u32 y = blockIdx.y;
    const u64 base = (u64)blockIdx.x * LP_TILE;
    u32 s0 = 0, s1 = 0;
    for (u32 k = 0; k < LP_PER; k++) {
        const u64 i = base + threadIdx.x + k * LP_THREADS;
        if (i < n) {
            const u32 v = cnt[(u64)y * n + i];
            s0 += v;
            s1 += v == 0;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d);
        s1 += __shfl_xor(s1, d);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 a = 0, b = 0;
        for (u32 w = 0; w < LP_THREADS / 64; w++) {
            a += red[0][w];
            b += red[1][w];
        }
        tile_sums[((u64)y * tiles + blockIdx.x) * 2] = a;
        tile_sums[((u64)y * tiles + blockIdx.x) * 2 + 1] = b;
    }
}
// one block per lookup: exclusive scan of its tile sums in place; totals[y] = (rows counted, leftovers)
__global__ void __launch_bounds__(1024) k_lp_scan_top(u32* tile_sums, u32 tiles, u32* totals) {
    __shared__ u32 sh[2][1024];
    const u32 y = blockIdx.x;
    u32* ts = tile_sums + (u64)y * tiles * 2;
    u32 carry0 = 0, carry1 = 0;
    for (u32 base = 0; base < tiles; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v0 = i < tiles ? ts[2 * i] : 0, v1 = i < tiles ? ts[2 * i + 1] : 0;
        sh[0][threadIdx.x] = v0;
        sh[1][threadIdx.x] = v1;
        __syncthreads();
        for (u32 d = 1; d < 1024; d <<= 1) {
            const u32 a = threadIdx.x >= d ? sh[0][threadIdx.x - d] : 0, b = threadIdx.x >= d ? sh[1][threadIdx.x - d] : 0;
            __syncthreads();
            sh[0][threadIdx.x] += a;
            sh[1][threadIdx.x] += b;
            __syncthreads();
        }
        if (i < tiles) {
            ts[2 * i] = carry0 + sh[0][threadIdx.x] - v0;
            ts[2 * i + 1] = carry1 + sh[1][threadIdx.x] - v1;
        }
        carry0 += sh[0][1023];
        carry1 += sh[1][1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[2 * y] = carry0;
        totals[2 * y + 1] = carry1;
    }
}
__global__ void __launch_bounds__(LP_THREADS) k_lp_scan_apply(const u32* cnt, u64 n, u32 tiles, const u32* tile_sums, u32* used_start, u32* used_idx, u32* lsrc) {
    __shared__ u32 sh[2][LP_THREADS];
    const u32 y = blockIdx.y;
    const u64 base = (u64)blockIdx.x * LP_TILE + (u64)threadIdx.x * LP_PER;      // a thread owns LP_PER consecutive positions
    u32 v[LP_PER], s0 = 0, s1 = 0;
    for (u32 k = 0; k < LP_PER; k++) {
        v[k] = base + k < n ? cnt[(u64)y * n + base + k] : 0;
        s0 += v[k];
        s1 += (base + k < n) && v[k] == 0;
    }
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    __syncthreads();
    for (u32 d = 1; d < LP_THREADS; d <<= 1) {
        const u32 a = threadIdx.x >= d ? sh[0][threadIdx.x - d] : 0, b = threadIdx.x >= d ? sh[1][threadIdx.x - d] : 0;
        __syncthreads();
        sh[0][threadIdx.x] += a;
        sh[1][threadIdx.x] += b;
        __syncthreads();
    }
    u32 r0 = tile_sums[((u64)y * tiles + blockIdx.x) * 2] + sh[0][threadIdx.x] - s0;      // rows before this entry
    u32 r1 = tile_sums[((u64)y * tiles + blockIdx.x) * 2 + 1] + sh[1][threadIdx.x] - s1;  // leftovers before this entry
    for (u32 k = 0; k < LP_PER; k++) {
        const u64 j = base + k;
        if (j >= n) break;
        if (v[k]) {      // the (j - r1)-th USED entry: its rows start at r0
            used_start[(u64)y * n + (j - r1)] = r0;
            used_idx[(u64)y * n + (j - r1)] = (u32)j;
        } else {
            lsrc[(u64)y * n + r1] = (u32)j;      // the r1-th leftover (ascending)
        }
        r0 += v[k];
        r1 += v[k] == 0;
    }
}

FP_DEV u64 lp_last_le(const u32* a, u64 n, u32 x) {      // the last index with a[index] <= x (a non-decreasing, a[0] <= x)
    u64 lo = 0, hi = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}
// output row i of lookup y.  The row belongs to the k-th USED table entry (the last one whose rows start at or before i); on the first
// row of the entry S' = A'; otherwise the row is the r-th repeated one and takes the (m - 1 - r)-th leftover (ascending), m = leftovers =
// repeated rows.  The search runs over the used entries only (a few hundred for a range table, all of them cache-resident).
template <class F>
__global__ void k_lp_emit(LpCols c, const fe* sorted, u64 n, u64 npad, const u32* used_start, const u32* used_idx, const u32* lsrc, const u32* totals) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 y = blockIdx.y;
    if (i >= n) return;
    if (totals[2 * y] != (u32)n) return;                  // a missing input value: the call fails, nothing meaningful to write
    const fe* T = sorted + (u64)c.table_of[y] * npad;
    const u32 m = totals[2 * y + 1];
    const u64 k = lp_last_le(used_start + (u64)y * n, n - m, (u32)i);
    const fe a = f_to_mont<F>(f_load(&T[used_idx[(u64)y * n + k]]));
    f_store(&c.out_in[y][i], a);
    if (used_start[(u64)y * n + k] == (u32)i) {
        f_store(&c.out_tab[y][i], a);
        return;
    }
    const u32 r = (u32)i - (u32)(k + 1);
    f_store(&c.out_tab[y][i], f_to_mont<F>(f_load(&T[lsrc[(u64)y * n + (m - 1 - r)]])));
}

template <class F>
int lp_run(dehalo_ctx* ctx, const LpCols& c, u32 B, u32 U, u64 n, hipStream_t s, int* d_status) {
    const u64 tiles = (n + LP_TILE - 1) / LP_TILE, npad = tiles * LP_TILE;
    const size_t pad = 256;
    const size_t bytes = 2 * ((size_t)U * npad * sizeof(fe) + pad) + 4 * ((size_t)B * std::max<size_t>(n, LP_TILE + 1) * 4 + pad) + ((size_t)B * tiles * 8 + pad) + 2 * ((size_t)B * 8 + pad) + 4096 +
                         2 * ((size_t)U * LP_TILE * sizeof(fe) + pad) + ((size_t)U * (LP_TILE + 1) * 4 + pad);
    TRY(dh_ensure(ctx, ctx->ws_lookup, bytes));
    char* p = (char*)ctx->ws_lookup.p;
    auto take = [&](size_t b) { char* r = p; p += (b + 255) & ~(size_t)255; return r; };
    fe* k0 = (fe*)take((size_t)U * npad * sizeof(fe));
    fe* k1 = (fe*)take((size_t)U * npad * sizeof(fe));
    const size_t per_lookup = std::max<size_t>(n, LP_TILE + 1);      // (the distinct-rows path keeps LP_TILE + 1 words per lookup in these, whatever n is)
    u32* cnt = (u32*)take((size_t)B * per_lookup * 4);
    u32* used_start = (u32*)take((size_t)B * per_lookup * 4);
    u32* used_idx = (u32*)take((size_t)B * per_lookup * 4);
    u32* lsrc = (u32*)take((size_t)B * per_lookup * 4);
    u32* tile_sums = (u32*)take((size_t)B * tiles * 8);
    u32* totals = (u32*)take((size_t)B * 8);
    int* err = (int*)take((size_t)B * 4);
    fe* dk0 = (fe*)take((size_t)U * LP_TILE * sizeof(fe));
    fe* dk1 = (fe*)take((size_t)U * LP_TILE * sizeof(fe));
    u32* dstart = (u32*)take((size_t)U * (LP_TILE + 1) * 4);
    const int lds_keys = LP_TILE * 32;
    HIP_TRY(ctx, dh_func_lds(ctx, (const void*)k_lp_tile_sort, lds_keys));
    bool distinct = true;
    for (u32 t = 0; t < U; t++) distinct = distinct && c.rep[t] && c.mult[t] && c.ndist[t] > 0 && c.ndist[t] <= LP_TILE;
    HIP_TRY(ctx, hipMemsetAsync(err, 0, (size_t)B * 4, s));
    if (distinct) {
        // tables given as distinct rows: everything after the (one-tile) sort runs over the <= LP_TILE distinct keys instead of the n table positions
        u32* dcnt = cnt;                                   // [B][LP_TILE]
        u32* rowstart = used_start;                        // [B][LP_TILE + 1] each (the buffers hold max(n, LP_TILE + 1) words per lookup)
        u32* usedbefore = used_idx;
        u32* leftstart = lsrc;
        u32 dlen = 64;                                     // a power of two >= every table's distinct rows: what the sort and the searches run over
        for (u32 t = 0; t < U; t++) while (dlen < c.ndist[t]) dlen <<= 1;
        k_lp_distinct_keys<F><<<dim3(LP_TILE / 256, U), 256, 0, s>>>(c, n, dk0);
        if (dlen <= 512) {
            k_lp_small_sort<<<dim3((4 * dlen + LP_SS_THREADS - 1) / LP_SS_THREADS, U), LP_SS_THREADS, 0, s>>>(c, dk0, dk1);      // a quad per key
        } else k_lp_tile_sort<<<dim3(1, U), LP_SORT_THREADS, lds_keys, s>>>(dk0, dk1, LP_TILE);
        k_lp_distinct_starts<F><<<U, 1024, 0, s>>>(c, n, dk1, dstart);
        HIP_TRY(ctx, hipMemsetAsync(dcnt, 0, (size_t)B * LP_TILE * 4, s));
        const int lds_rank = (int)(36 * dlen);
        if (lds_rank > 48 * 1024) HIP_TRY(ctx, dh_func_lds(ctx, (const void*)k_lp_rank_distinct<F>, lds_rank));
        k_lp_rank_distinct<F><<<dim3((u32)tiles, B), LP_SORT_THREADS, lds_rank, s>>>(c, dk1, n, dcnt, err, dlen);
        k_lp_scan_distinct<<<B, 1024, 0, s>>>(c, dcnt, dstart, rowstart, usedbefore, leftstart);
        k_lp_emit_distinct<F><<<dim3((u32)((n + 255) / 256), B), 256, 0, s>>>(c, dk1, n, rowstart, usedbefore, leftstart, err);
    } else {
        fe *src = k1, *dst = k0;
        k_lp_canon_tables<F><<<dim3((u32)((npad + 255) / 256), U), 256, 0, s>>>(c, n, npad, k0);
        k_lp_tile_sort<<<dim3((u32)tiles, U), LP_SORT_THREADS, lds_keys, s>>>(k0, k1, npad);
        for (u64 run = LP_TILE; run < npad; run <<= 1) {
            k_lp_merge<<<dim3((u32)(npad / LP_MERGE_OUT), U), LP_MERGE_THREADS, 0, s>>>(src, dst, npad, run);
            std::swap(src, dst);
        }
        HIP_TRY(ctx, hipMemsetAsync(cnt, 0, (size_t)B * n * 4, s));
        k_lp_rank<F><<<dim3((u32)tiles, B), LP_SORT_THREADS, 0, s>>>(c, src, n, npad, cnt, err);
        k_lp_scan_tiles<<<dim3((u32)tiles, B), LP_THREADS, 0, s>>>(cnt, n, (u32)tiles, tile_sums);
        k_lp_scan_top<<<B, 1024, 0, s>>>(tile_sums, (u32)tiles, totals);
        k_lp_scan_apply<<<dim3((u32)tiles, B), LP_THREADS, 0, s>>>(cnt, n, (u32)tiles, tile_sums, used_start, used_idx, lsrc);
        k_lp_emit<F><<<dim3((u32)((n + 255) / 256), B), 256, 0, s>>>(c, src, n, npad, used_start, used_idx, lsrc, totals);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (d_status) {      // deferred: the flags stay on the device for the caller to read with whatever it reads back next; no synchronisation here
        HIP_TRY(ctx, hipMemcpyAsync(d_status, err, B * sizeof(int), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    int host_err[LP_MAX_BATCH];
    TRY(dh_d2h(ctx, host_err, err, B * sizeof(int), s));
    HIP_TRY(ctx, hipStreamSynchronize(s));     // upstream returns Err(ConstraintSystemFailure) from this call: so must we
    for (u32 y = 0; y < B; y++)
        if (host_err[y])
            return dh_fail(ctx, DEHALO_ERR_NOT_IN_TABLE, "permute_expression_pair: an input value of lookup " + std::to_string(y) + " of the call is not in the table (ConstraintSystemFailure)");
    return 0;
}

}  // namespace

// `batch` lookups at once, given as pointer lists: lookups whose table pointers are equal share one sort.
int lookup_permute_ptrs(dehalo_ctx* ctx, int field, const fe* const* d_inputs, const fe* const* d_tables, uint64_t n, size_t batch, fe* const* d_out_inputs,
                        fe* const* d_out_tables, hipStream_t s, int* d_status, const LookupDistinct* distinct) {
    if (n == 0 || batch == 0) {
        if (d_status && batch) HIP_TRY(ctx, hipMemsetAsync(d_status, 0, batch * sizeof(int), s));
        return 0;
    }
    if (n >= (1ull << 31)) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: too many rows");
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    for (size_t first = 0; first < batch; first += LP_MAX_BATCH) {
        const u32 B = (u32)std::min<size_t>(LP_MAX_BATCH, batch - first);
        LpCols c{};
        u32 U = 0;
        for (u32 y = 0; y < B; y++) {
            c.in[y] = d_inputs[first + y];
            c.out_in[y] = d_out_inputs[first + y];
            c.out_tab[y] = d_out_tables[first + y];
            u32 t = 0;
            while (t < U && c.tab[t] != d_tables[first + y]) t++;
            if (t == U) {
                if (distinct) { c.rep[U] = distinct[first + y].d_rep_rows; c.mult[U] = distinct[first + y].d_mult; c.ndist[U] = distinct[first + y].count; }
                c.tab[U++] = d_tables[first + y];
            }
            c.table_of[y] = t;
        }
        int rc;
        switch (field) {
            case DEHALO_FIELD_BN254_FR: rc = lp_run<Bn254Fr>(ctx, c, B, U, n, s, d_status ? d_status + first : nullptr); break;
            case DEHALO_FIELD_BN254_FQ: rc = lp_run<Bn254Fq>(ctx, c, B, U, n, s, d_status ? d_status + first : nullptr); break;
            case DEHALO_FIELD_PASTA_FP: rc = lp_run<PastaFp>(ctx, c, B, U, n, s, d_status ? d_status + first : nullptr); break;
            case DEHALO_FIELD_PASTA_FQ: rc = lp_run<PastaFq>(ctx, c, B, U, n, s, d_status ? d_status + first : nullptr); break;
            default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
        }
        if (rc) return rc;
    }
    return 0;
}

// column y of inputs / tables / outputs starts y * stride elements in
int lookup_permute_impl(dehalo_ctx* ctx, int field, const fe* d_inputs, const fe* d_tables, uint64_t n, size_t batch, uint64_t stride, fe* d_out_inputs,
                        fe* d_out_tables, hipStream_t s) {
    std::vector<const fe*> in(batch), tab(batch);
    std::vector<fe*> oi(batch), ot(batch);
    for (size_t y = 0; y < batch; y++) {
        in[y] = d_inputs + y * stride;
        tab[y] = d_tables + y * stride;
        oi[y] = d_out_inputs + y * stride;
        ot[y] = d_out_tables + y * stride;
    }
    return lookup_permute_ptrs(ctx, field, in.data(), tab.data(), n, batch, oi.data(), ot.data(), s);
}
