// pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of k_msm_accum0 (MI355X_MICROARCH.md, HBM section:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Pattern: every lane gathers one 64-byte record (4 x global_load_dwordx4) at a random index of a table far larger than the
// 256 MiB Infinity Cache, each record exactly once; index array read coalesced.  Known bytes per launch:
//   records * 64 (table) + records * 4 (indices) read, records * 4 written.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

struct alignas(16) rec64 { uint4 a, b, c, d; };

__global__ void k_gather64(const rec64* table, const uint32_t* idx, uint32_t n, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rec64* r = &table[idx[i]];
    uint4 a = r->a, b = r->b, c = r->c, d = r->d;
    out[i] = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
}
// the same bytes as a coalesced stream (16 B per lane, consecutive): the case the guide calibrated (FETCH_SIZE reports 1/2)
__global__ void k_stream16(const uint4* table, uint64_t n16, uint32_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t j = i; j < n16; j += (uint64_t)gridDim.x * blockDim.x) { uint4 v = table[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    out[i] = acc;
}

// read + write of the same bytes (16 B per lane, grid-stride): the copy ceiling SURVEY.md 8(d) asks to be recorded
__global__ void k_copy16(const uint4* src, uint4* dst, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t j = i; j < n16; j += (uint64_t)gridDim.x * blockDim.x) dst[j] = src[j];
}

int main() {
    const uint32_t n = 1u << 24;                         // 16.8 M records = 1 GiB
    rec64* table; uint32_t *idx, *out;
    hipMalloc(&table, (size_t)n * sizeof(rec64)); hipMalloc(&idx, (size_t)n * 4); hipMalloc(&out, (size_t)n * 4);
    hipMemset(table, 1, (size_t)n * sizeof(rec64));
    std::vector<uint32_t> h(n);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(12345);
    std::shuffle(h.begin(), h.end(), rng);
    hipMemcpy(idx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        k_gather64<<<n / 256, 256>>>(table, idx, n, out);
        k_stream16<<<4096, 256>>>((const uint4*)table, (uint64_t)n * 4, out);
    }
    hipDeviceSynchronize();
    {   // durations (HIP events, best of 5): read-only stream and copy of 1 GiB -> profiles/copy_ceiling.json
        rec64* dst; hipMalloc(&dst, (size_t)n * sizeof(rec64));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best_s = 1e9f, best_c = 1e9f, best_g = 1e9f, ms;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(a); k_stream16<<<4096, 256>>>((const uint4*)table, (uint64_t)n * 4, out); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b); if (ms < best_s) best_s = ms;
            hipEventRecord(a); k_copy16<<<8192, 256>>>((const uint4*)table, (uint4*)dst, (uint64_t)n * 4); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b); if (ms < best_c) best_c = ms;
            hipEventRecord(a); k_gather64<<<n / 256, 256>>>(table, idx, n, out); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b); if (ms < best_g) best_g = ms;
        }
        const double gib = (double)n * 64;
        printf("{\"copy_ceiling\": {\"k_stream16_ms\": %.4f, \"k_stream16_read_GBps\": %.1f, \"k_copy16_ms\": %.4f, \"k_copy16_GBps_read_plus_write\": %.1f, "
               "\"k_gather64_ms\": %.4f, \"k_gather64_GBps\": %.1f, \"bytes\": %.0f, \"note\": \"1 GiB table; HIP events, best of 5\"}}\n",
               best_s, gib / best_s / 1e6, best_c, 2 * gib / best_c / 1e6, best_g, (double)n * 68 / best_g / 1e6, gib);
        hipFree(dst);
    }
    printf("k_gather64: logical read bytes per launch = %llu (table) + %llu (indices); written = %llu\n", (unsigned long long)n * 64, (unsigned long long)n * 4, (unsigned long long)n * 4);
    printf("k_stream16: logical read bytes per launch = %llu\n", (unsigned long long)n * 64);
    return 0;
}
