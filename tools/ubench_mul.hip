// ubench_mul.hip -- compares f29 multiplication schedules on gfx950 (dev tool).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I delay-encryption-in-halo2_amd/csrc tools/ubench_mul.hip -o gpurun_out/ubench_mul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ec29.cuh"

// product scanning: the carry of column k-1 is the initial addend of column k's multiply-add chain
template <class F, bool OPAQUE>
FP_DEV f29 mul_ps(const f29& a, const f29& b) {
    u32 m[9];
    u32 P[9];
#pragma unroll
    for (int j = 0; j < 9; j++) { P[j] = F::P[j]; if (OPAQUE && F::P[j] != 0) asm("" : "+s"(P[j])); }
    f29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) if (k - i >= 0 && k - i < 9) { acc = mad_wide(a.v[i], b.v[k - i], acc); asm("" : "+v"(acc)); }
#pragma unroll
        for (int i = 0; i < 9; i++) if (i < k && k - i < 9 && F::P[k - i] != 0) { acc = mad_wide(m[i], P[k - i], acc); asm("" : "+v"(acc)); }
        if (k < 9) {
            if (F::INV == F29_MASK) m[k] = (0u - (u32)acc) & F29_MASK; else m[k] = ((u32)acc * F::INV) & F29_MASK;
            acc = mad_wide(m[k], P[0], acc); asm("" : "+v"(acc));
        } else r.v[k - 9] = (u32)acc & F29_MASK;
        acc >>= F29_BITS;
    }
    r.v[8] = (u32)acc;
    return r;
}

template <class F9, int V, int CH>
__global__ void k_mul(f29* out, const fe* in, int iters) {
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    f29 m = f29_unpack(f_load(&in[gid]));
    m.v[8] &= 0xffff;
    f29 x[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { x[c] = m; x[c].v[0] ^= c + 1; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (V == 0) x[c] = f29_mul<F9>(x[c], m);
            else if (V == 1) x[c] = mul_ps<F9, false>(x[c], m);
            else x[c] = mul_ps<F9, true>(x[c], m);
        }
    }
    f29 r = x[0];
#pragma unroll
    for (int c = 1; c < CH; c++) r = f29_add(r, x[c]);
    out[gid] = r;
}

template <class K, class... A>
float time_kernel(K k, dim3 g, dim3 b, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, g, b, 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, g, b, 0, 0, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    const int CUS = p.multiProcessorCount;
    const int fit = 1000;
    for (int wps : {2, 4, 8}) {   // waves per SIMD
        const int threads = 256, blocks = CUS * wps;      // 256 threads = 4 waves = 1 per SIMD per block
        size_t nel = (size_t)blocks * threads;
        fe* fin; f29* fo; hipMalloc(&fin, nel * 32); hipMalloc(&fo, nel * sizeof(f29));
        std::vector<u32> h(nel * 8);
        for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffff : 0xffffffff);
        hipMemcpy(fin, h.data(), nel * 32, hipMemcpyHostToDevice);
        f29 ref[2]; f29 got;
#define RUN(F, V, CH, name) { float ms = time_kernel(k_mul<F, V, CH>, dim3(blocks), dim3(threads), fo, fin, fit); \
        hipMemcpy(&got, fo + 12345 % nel, sizeof(f29), hipMemcpyDeviceToHost); \
        printf("waves/SIMD %d  %-34s %8.3f ms %8.2f Gmul/s  chk %08x %08x\n", wps, name, ms, (double)nel * fit * CH / ms / 1e6, got.v[0], got.v[8]); }
        RUN(PastaFp29, 0, 1, "pasta operand-scan (current)");
        RUN(PastaFp29, 1, 1, "pasta product-scan");
        RUN(PastaFp29, 2, 1, "pasta product-scan opaque P");
        RUN(PastaFp29, 0, 2, "pasta operand-scan x2 chains");
        RUN(PastaFp29, 2, 2, "pasta product-scan opaque x2 chains");
        RUN(Bn254Fq29, 0, 1, "bn254 operand-scan (current)");
        RUN(Bn254Fq29, 1, 1, "bn254 product-scan");
        RUN(Bn254Fq29, 2, 2, "bn254 product-scan opaque x2 chains");
        hipFree(fin); hipFree(fo);
    }
    return 0;
}
