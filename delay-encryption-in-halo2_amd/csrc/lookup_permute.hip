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
// Upstream: Vec::sort + BTreeMap on one thread.  Here: canonical keys -> four stable LSD passes of
// rocPRIM's 64-bit radix sort carrying a permutation (rocPRIM is header-only, compiled in; the
// sort is not the prover's hot loop and a hand-written 256-bit radix sort would be the same
// algorithm), then flag / binary-search / scan / scatter kernels.  Outputs are gathered from the
// ORIGINAL Montgomery elements, so no value is ever re-encoded.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "fp.cuh"
#include "field_constants.h"
#include "internal.hpp"

namespace {

template <class F>
__global__ void k_lp_canon(const fe* in, u64 n, fe* canon) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f_store(&canon[i], f_from_mont<F>(f_load(&in[i])));
}
__global__ void k_lp_iota(u32* p, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (u32)i;
}
// keys[i] = 64-bit limb `limb` of canon[perm[i]]
__global__ void k_lp_limb(const fe* canon, const u32* perm, u32 limb, u64 n, unsigned long long* keys) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* w = canon[perm[i]].v;
    keys[i] = (unsigned long long)w[2 * limb] | ((unsigned long long)w[2 * limb + 1] << 32);
}
// ors[l] |= every key's 64-bit limb l: limbs that are zero everywhere need no sort pass, and the
// highest set bit bounds the digits of the others (range tables hold small values)
__global__ void k_lp_limb_or(const fe* canon, u64 n, unsigned long long* ors) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v[4] = {0, 0, 0, 0};
    if (i < n) {
        const u32* w = canon[i].v;
#pragma unroll
        for (int l = 0; l < 4; l++) v[l] = (unsigned long long)w[2 * l] | ((unsigned long long)w[2 * l + 1] << 32);
    }
#pragma unroll
    for (int l = 0; l < 4; l++) {
        unsigned long long x = v[l];
        for (int d = 32; d >= 1; d >>= 1) x |= __shfl_xor(x, d);
        if ((threadIdx.x & 63) == 0 && x) atomicOr(&ors[l], x);
    }
}
__global__ void k_lp_gather(const fe* canon, const u32* perm, u64 n, fe* sorted) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f_store(&sorted[i], f_load(&canon[perm[i]]));
}
FP_DEV int lp_cmp(const fe& a, const fe& b) {          // canonical integers, most significant word first
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
    }
    return 0;
}
// repeated[i] = A[i] == A[i-1]; every first occurrence looks its value up in T (sorted) and marks
// the FIRST copy there as consumed; a miss raises *err.
__global__ void k_lp_flags(const fe* A, const fe* T, u64 n, u32* repeated, u32* consumed, int* err) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe a = f_load(&A[i]);
    bool first = i == 0 || lp_cmp(a, f_load(&A[i - 1])) != 0;
    repeated[i] = first ? 0u : 1u;
    if (!first) return;
    u64 lo = 0, hi = n;                                   // lower bound of a in T
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (lp_cmp(f_load(&T[mid]), a) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < n && lp_cmp(f_load(&T[lo]), a) == 0) consumed[lo] = 1u;
    else atomicExch(err, 1);
}
__global__ void k_lp_not(const u32* consumed, u64 n, u32* leftover) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leftover[i] = consumed[i] ? 0u : 1u;
}
// lsrc[q] = original table row of the q-th leftover (ascending)
__global__ void k_lp_compact(const u32* leftover, const u32* lrank, const u32* perm_t, u64 n, u32* lsrc) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && leftover[j]) lsrc[lrank[j]] = perm_t[j];
}
__global__ void k_lp_emit(const fe* input, const fe* table, const u32* perm_a, const u32* repeated, const u32* rrank, const u32* lsrc, u64 n, fe* out_input,
                          fe* out_table) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe a = f_load(&input[perm_a[i]]);
    f_store(&out_input[i], a);
    if (!repeated[i]) { f_store(&out_table[i], a); return; }
    const u32 m = rrank[n - 1] + repeated[n - 1];         // number of repeated rows = number of leftovers
    f_store(&out_table[i], f_load(&table[lsrc[m - 1 - rrank[i]]]));
}

int canon_dispatch(dehalo_ctx* ctx, int field, const fe* in, u64 n, fe* out, hipStream_t s) {
    const u32 blocks = (u32)((n + 255) / 256);
    switch (field) {
        case DEHALO_FIELD_BN254_FR: k_lp_canon<Bn254Fr><<<blocks, 256, 0, s>>>(in, n, out); break;
        case DEHALO_FIELD_BN254_FQ: k_lp_canon<Bn254Fq><<<blocks, 256, 0, s>>>(in, n, out); break;
        case DEHALO_FIELD_PASTA_FP: k_lp_canon<PastaFp><<<blocks, 256, 0, s>>>(in, n, out); break;
        case DEHALO_FIELD_PASTA_FQ: k_lp_canon<PastaFq><<<blocks, 256, 0, s>>>(in, n, out); break;
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

// canonical keys sorted ascending; perm_out[i] = original row of the i-th smallest.  bits[l] = bit
// length of the OR of limb l over the column (0: the pass is skipped).
int sort_column(dehalo_ctx* ctx, const fe* canon, u64 n, const unsigned bits[4], fe* sorted, u32* perm_a, u32* perm_b, unsigned long long* keys_a,
                unsigned long long* keys_b, void* tmp, size_t tmp_bytes, u32** perm_out, hipStream_t s) {
    const u32 blocks = (u32)((n + 255) / 256);
    k_lp_iota<<<blocks, 256, 0, s>>>(perm_a, n);
    u32 *pin = perm_a, *pout = perm_b;
    for (u32 limb = 0; limb < 4; limb++) {
        if (bits[limb] == 0) continue;
        k_lp_limb<<<blocks, 256, 0, s>>>(canon, pin, limb, n, keys_a);
        size_t bytes = tmp_bytes;
        HIP_TRY(ctx, rocprim::radix_sort_pairs(tmp, bytes, keys_a, keys_b, pin, pout, n, 0, bits[limb], s));
        std::swap(pin, pout);
    }
    k_lp_gather<<<blocks, 256, 0, s>>>(canon, pin, n, sorted);
    HIP_TRY(ctx, hipGetLastError());
    *perm_out = pin;
    return 0;
}

}  // namespace

int lookup_permute_impl(dehalo_ctx* ctx, int field, const fe* d_input, const fe* d_table, uint64_t n, fe* d_out_input, fe* d_out_table, hipStream_t s) {
    if (n == 0) return 0;
    if (n >= (1ull << 31)) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: too many rows");
    ScopedTimer timer(ctx, s, DEHALO_K_POLY);
    size_t sort_tmp = 0, scan_tmp = 0;
    HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, sort_tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (u32*)nullptr, (u32*)nullptr, n, 0, 64, s));
    HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, scan_tmp, (u32*)nullptr, (u32*)nullptr, 0u, n, rocprim::plus<u32>(), s));
    const size_t tmp_bytes = std::max(sort_tmp, scan_tmp);
    const size_t pad = 256;
    size_t total = 4 * (n * sizeof(fe) + pad) + 2 * (n * 8 + pad) + 4 * (n * 4 + pad) + 6 * (n * 4 + pad) + tmp_bytes + pad + 1024;
    TRY(dh_ensure(ctx, ctx->ws_lookup, total));
    Carve c{(char*)ctx->ws_lookup.p};
    fe* canon_a = c.take<fe>(n); fe* canon_t = c.take<fe>(n); fe* A = c.take<fe>(n); fe* T = c.take<fe>(n);
    unsigned long long* keys_a = c.take<unsigned long long>(n); unsigned long long* keys_b = c.take<unsigned long long>(n);
    u32* pa0 = c.take<u32>(n); u32* pa1 = c.take<u32>(n); u32* pt0 = c.take<u32>(n); u32* pt1 = c.take<u32>(n);
    u32* repeated = c.take<u32>(n); u32* consumed = c.take<u32>(n); u32* leftover = c.take<u32>(n); u32* rrank = c.take<u32>(n); u32* lrank = c.take<u32>(n);
    u32* lsrc = c.take<u32>(n);
    int* err = c.take<int>(1);
    unsigned long long* ors = c.take<unsigned long long>(8);
    void* tmp = c.take<char>(tmp_bytes);
    u32 *perm_a = nullptr, *perm_t = nullptr;
    const u32 blocks = (u32)((n + 255) / 256);
    TRY(canon_dispatch(ctx, field, d_input, n, canon_a, s));
    TRY(canon_dispatch(ctx, field, d_table, n, canon_t, s));
    HIP_TRY(ctx, hipMemsetAsync(ors, 0, 8 * sizeof(unsigned long long), s));
    k_lp_limb_or<<<blocks, 256, 0, s>>>(canon_a, n, ors);
    k_lp_limb_or<<<blocks, 256, 0, s>>>(canon_t, n, ors + 4);
    unsigned long long host_ors[8];
    HIP_TRY(ctx, hipMemcpyAsync(host_ors, ors, sizeof(host_ors), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));          // the pass plan depends on the data (the call synchronises at its end anyway)
    unsigned bits_a[4], bits_t[4];
    for (int l = 0; l < 4; l++) {
        bits_a[l] = host_ors[l] ? 64 - (unsigned)__builtin_clzll(host_ors[l]) : 0;
        bits_t[l] = host_ors[4 + l] ? 64 - (unsigned)__builtin_clzll(host_ors[4 + l]) : 0;
    }
    TRY(sort_column(ctx, canon_a, n, bits_a, A, pa0, pa1, keys_a, keys_b, tmp, tmp_bytes, &perm_a, s));
    TRY(sort_column(ctx, canon_t, n, bits_t, T, pt0, pt1, keys_a, keys_b, tmp, tmp_bytes, &perm_t, s));
    HIP_TRY(ctx, hipMemsetAsync(consumed, 0, n * 4, s));
    HIP_TRY(ctx, hipMemsetAsync(err, 0, sizeof(int), s));
    HIP_TRY(ctx, hipMemsetAsync(lsrc, 0, n * 4, s));          // a failed lookup leaves gaps: keep every index in range
    k_lp_flags<<<blocks, 256, 0, s>>>(A, T, n, repeated, consumed, err);
    k_lp_not<<<blocks, 256, 0, s>>>(consumed, n, leftover);
    size_t bytes = tmp_bytes;
    HIP_TRY(ctx, rocprim::exclusive_scan(tmp, bytes, repeated, rrank, 0u, n, rocprim::plus<u32>(), s));
    bytes = tmp_bytes;
    HIP_TRY(ctx, rocprim::exclusive_scan(tmp, bytes, leftover, lrank, 0u, n, rocprim::plus<u32>(), s));
    k_lp_compact<<<blocks, 256, 0, s>>>(leftover, lrank, perm_t, n, lsrc);
    k_lp_emit<<<blocks, 256, 0, s>>>(d_input, d_table, perm_a, repeated, rrank, lsrc, n, d_out_input, d_out_table);
    HIP_TRY(ctx, hipGetLastError());
    int host_err = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&host_err, err, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));     // upstream returns Err(ConstraintSystemFailure) from this call: so must we
    if (host_err) return dh_fail(ctx, DEHALO_ERR_NOT_IN_TABLE, "permute_expression_pair: an input value is not in the table (ConstraintSystemFailure)");
    return 0;
}
