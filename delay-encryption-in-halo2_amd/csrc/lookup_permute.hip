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
// Upstream: Vec::sort + BTreeMap on one thread.  Here: canonical keys -> one pass of rocPRIM's 64-bit radix sort on the leading
// bits + a verification of the full order (fallback: four stable LSD passes), carrying a permutation (rocPRIM is header-only, compiled in; the
// sort is not the prover's hot loop and a hand-written 256-bit radix sort would be the same
// algorithm), then flag / binary-search / scan / scatter kernels.  Outputs are gathered from the
// ORIGINAL Montgomery elements, so no value is ever re-encoded.
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "fp.cuh"
#include "field_constants.h"
#include "internal.hpp"

namespace {

// Batched layout: 2B key columns of n rows each, inputs first then tables, at canon[y * n + i].
template <class F>
__global__ void k_lp_canon(const fe* inputs, const fe* tables, u64 stride, u32 B, u64 n, fe* canon) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 y = blockIdx.y;
    if (i >= n) return;
    const fe* src = y < B ? inputs + (u64)y * stride : tables + (u64)(y - B) * stride;
    f_store(&canon[(u64)y * n + i], f_from_mont<F>(f_load(&src[i])));
}
__global__ void k_lp_iota(u32* p, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (u32)i;
}
// keys[i] = 64-bit limb `limb` of canon[perm[i]]   (limb == 4: the column id perm[i] / n)
__global__ void k_lp_limb(const fe* canon, const u32* perm, u32 limb, u64 n_col, u64 total, unsigned long long* keys) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u32 src = perm[i];
    if (limb == 4) { keys[i] = src / n_col; return; }
    const u32* w = canon[src].v;
    keys[i] = (unsigned long long)w[2 * limb] | ((unsigned long long)w[2 * limb + 1] << 32);
}
// ors[l] |= every key's 64-bit limb l: limbs that are zero everywhere need no sort pass, and the
// highest set bit bounds the digits of the others (range tables hold small values)
__global__ void k_lp_limb_or(const fe* canon, u64 total, unsigned long long* ors) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v[4] = {0, 0, 0, 0};
    if (i < total) {
        const u32* w = canon[i].v;
#pragma unroll
        for (int l = 0; l < 4; l++) v[l] = (unsigned long long)w[2 * l] | ((unsigned long long)w[2 * l + 1] << 32);
    }
#pragma unroll
    for (int l = 0; l < 4; l++) {
        unsigned long long x = v[l];
        for (int d = 32; d >= 1; d >>= 1) x |= __shfl_xor(x, d);
        // only a wave that would ADD bits touches the shared word (4 hot addresses otherwise serialise ~10^5 atomics)
        if ((threadIdx.x & 63) == 0 && (x & ~__atomic_load_n(&ors[l], __ATOMIC_RELAXED)) != 0) atomicOr(&ors[l], x);
    }
}
__global__ void k_lp_gather(const fe* canon, const u32* perm, u64 total, fe* sorted) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) f_store(&sorted[i], f_load(&canon[perm[i]]));
}
FP_DEV int lp_cmp(const fe& a, const fe& b) {          // canonical integers, most significant word first
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
    }
    return 0;
}
// lookup y: A = S[y], T = S[B + y] (both sorted).  repeated[i] = A[i] == A[i-1]; every first
// occurrence looks its value up in T and marks the FIRST copy there as consumed; a miss raises err[y].
__global__ void k_lp_flags(const fe* S, u32 B, u64 n, u32* repeated, u32* consumed, int* err) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 y = blockIdx.y;
    if (i >= n) return;
    const fe* A = S + (u64)y * n;
    const fe* T = S + (u64)(B + y) * n;
    const fe a = f_load(&A[i]);
    bool first = i == 0 || lp_cmp(a, f_load(&A[i - 1])) != 0;
    repeated[(u64)y * n + i] = first ? 0u : 1u;
    if (!first) return;
    u64 lo = 0, hi = n;                                   // lower bound of a in T
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (lp_cmp(f_load(&T[mid]), a) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < n && lp_cmp(f_load(&T[lo]), a) == 0) consumed[(u64)y * n + lo] = 1u;
    else atomicExch(&err[y], 1);
}
// bad += 1 for every adjacent pair of one key column that is out of order under the FULL 256-bit comparison (the fast path sorted
// on the leading bits only)
__global__ void k_lp_check_sorted(const fe* S, u64 n, u64 total, u32* bad) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || i % n == 0) return;
    if (lp_cmp(f_load(&S[i - 1]), f_load(&S[i])) > 0) atomicAdd(bad, 1u);
}
__global__ void k_lp_not(const u32* consumed, u64 total, u32* leftover) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) leftover[i] = consumed[i] ? 0u : 1u;
}
// lsrc[y][q] = original table row of lookup y's q-th leftover (ascending); ranks are global scans,
// made column-local by subtracting the rank at the column's first row
__global__ void k_lp_compact(const u32* leftover, const u32* lrank, const u32* perm, u32 B, u64 n, u32* lsrc) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 y = blockIdx.y;
    if (j >= n) return;
    const u64 base = (u64)y * n;
    if (leftover[base + j]) lsrc[base + (lrank[base + j] - lrank[base])] = perm[(u64)(B + y) * n + j] - (u32)((u64)(B + y) * n);
}
__global__ void k_lp_emit(const fe* inputs, const fe* tables, u64 stride, const u32* perm, const u32* repeated, const u32* rrank, const u32* lsrc, u32 B, u64 n,
                          fe* out_inputs, fe* out_tables) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 y = blockIdx.y;
    if (i >= n) return;
    const u64 base = (u64)y * n;
    const fe* input = inputs + (u64)y * stride;
    const fe* table = tables + (u64)y * stride;
    const fe a = f_load(&input[perm[base + i] - (u32)base]);
    f_store(&out_inputs[(u64)y * stride + i], a);
    if (!repeated[base + i]) { f_store(&out_tables[(u64)y * stride + i], a); return; }
    const u32 m = rrank[base + n - 1] + repeated[base + n - 1] - rrank[base];   // repeated rows of this lookup = its leftovers
    const u32 r = rrank[base + i] - rrank[base];
    f_store(&out_tables[(u64)y * stride + i], f_load(&table[lsrc[base + (m - 1 - r)]]));
}

template <class F>
void launch_canon(const fe* in, const fe* tab, u64 stride, u32 B, u64 n, fe* out, hipStream_t s) {
    k_lp_canon<F><<<dim3((u32)((n + 255) / 256), 2 * B), 256, 0, s>>>(in, tab, stride, B, n, out);
}
int canon_dispatch(dehalo_ctx* ctx, int field, const fe* in, const fe* tab, u64 stride, u32 B, u64 n, fe* out, hipStream_t s) {
    switch (field) {
        case DEHALO_FIELD_BN254_FR: launch_canon<Bn254Fr>(in, tab, stride, B, n, out, s); break;
        case DEHALO_FIELD_BN254_FQ: launch_canon<Bn254Fq>(in, tab, stride, B, n, out, s); break;
        case DEHALO_FIELD_PASTA_FP: launch_canon<PastaFp>(in, tab, stride, B, n, out, s); break;
        case DEHALO_FIELD_PASTA_FQ: launch_canon<PastaFq>(in, tab, stride, B, n, out, s); break;
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    }
    return 0;
}

struct Carve {
    char* p;
    template <class T> T* take(size_t count) {
        T* r = (T*)p;
        p += (count * sizeof(T) + 255) & ~(size_t)255;
        return r;
    }
};

}  // namespace

// `batch` lookups at once: column y of inputs / tables / outputs starts y * stride elements in.
// All 2 * batch key columns go through the SAME global sort passes (value limbs, then a last
// stable pass on the column id), so the number of launches does not grow with the batch.
int lookup_permute_impl(dehalo_ctx* ctx, int field, const fe* d_inputs, const fe* d_tables, uint64_t n, size_t batch, uint64_t stride, fe* d_out_inputs,
                        fe* d_out_tables, hipStream_t s) {
    if (n == 0 || batch == 0) return 0;
    const u32 B = (u32)batch;
    const u64 total = 2ull * B * n, half = (u64)B * n;
    if (total >= (1ull << 31)) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: too many rows");
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    size_t sort_tmp = 0, scan_tmp = 0;
    HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, sort_tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (u32*)nullptr, (u32*)nullptr, total, 0, 64, s));
    HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, scan_tmp, (u32*)nullptr, (u32*)nullptr, 0u, half, rocprim::plus<u32>(), s));
    const size_t tmp_bytes = std::max(sort_tmp, scan_tmp);
    const size_t pad = 256;
    size_t bytes_total = 2 * (total * sizeof(fe) + pad) + 2 * (total * 8 + pad) + 2 * (total * 4 + pad) + 6 * (half * 4 + pad) + (B * 4 + pad) + 1024 + tmp_bytes + pad;
    TRY(dh_ensure(ctx, ctx->ws_lookup, bytes_total));
    Carve c{(char*)ctx->ws_lookup.p};
    fe* canon = c.take<fe>(total); fe* S = c.take<fe>(total);
    unsigned long long* keys_a = c.take<unsigned long long>(total); unsigned long long* keys_b = c.take<unsigned long long>(total);
    u32* p0 = c.take<u32>(total); u32* p1 = c.take<u32>(total);
    u32* repeated = c.take<u32>(half); u32* consumed = c.take<u32>(half); u32* leftover = c.take<u32>(half); u32* rrank = c.take<u32>(half);
    u32* lrank = c.take<u32>(half); u32* lsrc = c.take<u32>(half);
    int* err = c.take<int>(B);
    unsigned long long* ors = c.take<unsigned long long>(4);
    void* tmp = c.take<char>(tmp_bytes);

    const u32 blocks_n = (u32)((n + 255) / 256), blocks_total = (u32)((total + 255) / 256), blocks_half = (u32)((half + 255) / 256);
    TRY(canon_dispatch(ctx, field, d_inputs, d_tables, stride, B, n, canon, s));
    u32 *pin = p0, *pout = p1;
    const unsigned col_bits = 2 * B > 1 ? 32 - (unsigned)__builtin_clz(2 * B - 1) : 0;
    auto sort_pass = [&](u32 limb, unsigned begin_bit, unsigned end_bit) -> int {
        k_lp_limb<<<blocks_total, 256, 0, s>>>(canon, pin, limb, n, total, keys_a);
        size_t bytes = tmp_bytes;
        HIP_TRY(ctx, rocprim::radix_sort_pairs(tmp, bytes, keys_a, keys_b, pin, pout, total, begin_bit, end_bit, s));
        std::swap(pin, pout);
        return 0;
    };
    u32* bad = (u32*)ors;                                      // (first word of the limb-OR area: the two uses never overlap)
    std::vector<int> host_err(B);
    // Fast path: a theta-compressed lookup value is tag * theta + value -- pseudo-random leading bits per tag, a small integer
    // added at the bottom (a range table's values are below 2^16) -- so the low 24 bits and the 40 bits below the modulus' top bit order
    // them: two short value passes + the stable column pass (~13 launches instead of ~40).  The plan does not depend on the data, so nothing is read back before the
    // end: the order is verified on the device with the full comparison (k_lp_check_sorted) while the permutation is already being
    // built from it, and ONE read-back at the end returns the lookup errors and that verdict.  Any violation (values that differ
    // only in the bits in between, e.g. a table of plain integers above 2^24) repeats the call with the full
    // least-significant-limb-first sort, whose pass plan is read from the data.
    const unsigned mod_bits = (field == DEHALO_FIELD_BN254_FR || field == DEHALO_FIELD_BN254_FQ) ? 254 : 255;
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool fast = attempt == 0;
        pin = p0; pout = p1;
        k_lp_iota<<<blocks_total, 256, 0, s>>>(p0, total);
        if (fast) {
            TRY(sort_pass(0, 0, 24));                                      // (three + five 8-bit radix passes)
            TRY(sort_pass(3, mod_bits - 40 - 192, mod_bits - 192));
            if (col_bits) TRY(sort_pass(4, 0, col_bits));
        } else {
            HIP_TRY(ctx, hipMemsetAsync(ors, 0, 4 * sizeof(unsigned long long), s));
            k_lp_limb_or<<<blocks_total, 256, 0, s>>>(canon, total, ors);
            unsigned long long host_ors[4];
            HIP_TRY(ctx, hipMemcpyAsync(host_ors, ors, sizeof(host_ors), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));          // the pass plan depends on the data
            for (u32 limb = 0; limb <= 4; limb++) {
                unsigned bits;
                if (limb < 4) bits = host_ors[limb] ? 64 - (unsigned)__builtin_clzll(host_ors[limb]) : 0;
                else bits = col_bits;                                 // last: the column id, stable
                if (bits == 0) continue;
                TRY(sort_pass(limb, 0, bits));
            }
        }
        k_lp_gather<<<blocks_total, 256, 0, s>>>(canon, pin, total, S);
        HIP_TRY(ctx, hipMemsetAsync(bad, 0, sizeof(u32), s));
        if (fast) k_lp_check_sorted<<<blocks_total, 256, 0, s>>>(S, n, total, bad);
        HIP_TRY(ctx, hipMemsetAsync(consumed, 0, half * 4, s));
        HIP_TRY(ctx, hipMemsetAsync(err, 0, B * sizeof(int), s));
        HIP_TRY(ctx, hipMemsetAsync(lsrc, 0, half * 4, s));       // a failed lookup leaves gaps: keep every index in range
        k_lp_flags<<<dim3(blocks_n, B), 256, 0, s>>>(S, B, n, repeated, consumed, err);
        k_lp_not<<<blocks_half, 256, 0, s>>>(consumed, half, leftover);
        size_t bytes = tmp_bytes;
        HIP_TRY(ctx, rocprim::exclusive_scan(tmp, bytes, repeated, rrank, 0u, half, rocprim::plus<u32>(), s));
        bytes = tmp_bytes;
        HIP_TRY(ctx, rocprim::exclusive_scan(tmp, bytes, leftover, lrank, 0u, half, rocprim::plus<u32>(), s));
        k_lp_compact<<<dim3(blocks_n, B), 256, 0, s>>>(leftover, lrank, pin, B, n, lsrc);
        k_lp_emit<<<dim3(blocks_n, B), 256, 0, s>>>(d_inputs, d_tables, stride, pin, repeated, rrank, lsrc, B, n, d_out_inputs, d_out_tables);
        HIP_TRY(ctx, hipGetLastError());
        u32 host_bad = 0;
        HIP_TRY(ctx, hipMemcpyAsync(host_err.data(), err, B * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpyAsync(&host_bad, bad, sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));     // upstream returns Err(ConstraintSystemFailure) from this call: so must we
        if (host_bad == 0) break;                  // (the full sort never sets it)
    }
    for (u32 y = 0; y < B; y++)
        if (host_err[y])
            return dh_fail(ctx, DEHALO_ERR_NOT_IN_TABLE, "permute_expression_pair: lookup " + std::to_string(y) + ": an input value is not in the table (ConstraintSystemFailure)");
    return 0;
}
