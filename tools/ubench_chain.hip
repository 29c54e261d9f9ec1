// Latency of the group operations the MSM's reduction tails are chains of (k_msm_merge_all, k_msm_reduce_local, k_msm_tree_sum), measured the
// way those kernels meet them: ONE wave per SIMD (1024 waves on the chip), each lane / quad running a dependent chain of K operations.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I delay-encryption-in-halo2_amd/csrc -o tools/ubench_chain tools/ubench_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec29.cuh"

// the merge kernels' inner loop: a quad walks a strided list of 144-byte records in global memory, one quad-cooperative addition per record
template <class CV, class F, int PREFETCH>
__global__ __launch_bounds__(64) void k_walk(int K, const xyzz29_rec* in, u32 n_in, xyzz29_rec* out) {
    const u32 q = (blockIdx.x * 64 + threadIdx.x) >> 2, Q = gridDim.x * 16;
    xyzz29 acc = x29_load(&in[q % n_in]);
    if (PREFETCH) {
        xyzz29 nxt = x29_load(&in[(q + Q) % n_in]);
        for (int i = 1; i < K; i++) {
            xyzz29 cur = nxt;
            nxt = x29_load(&in[(q + (u32)(i + 1) * Q) % n_in]);
            acc = x29_add_quad<F>(acc, cur);
        }
    } else {
        for (int i = 1; i < K; i++) acc = x29_add_quad<F>(acc, x29_load(&in[(q + (u32)i * Q) % n_in]));
    }
    x29_store(&out[blockIdx.x * 64 + threadIdx.x], acc);
}
template <class CV>
__global__ void k_fill(xyzz29_rec* in, u32 n) {
    typedef typename f29_of<typename CV::Base>::type F0;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe gx, gy;
    for (int j = 0; j < 8; j++) { gx.v[j] = CV::GX_M[j]; gy.v[j] = CV::GY_M[j]; }
    xyzz29 g; g.x = f29_from_std<F0>(gx); g.y = f29_from_std<F0>(gy); g.zz = f29_one<F0>(); g.zzz = f29_one<F0>();
    xyzz29 a = x29_double<F0>(g);
    for (u32 k = 0; k < (i & 7); k++) a = x29_add<F0>(a, g);
    x29_store(&in[i], a);
}
template <class CV, class F, int PREFETCH>
static void run_walk(const char* name, int K, int blocks) {
    const u32 n_in = 1u << 20;     // 151 MB of records: every load misses the L2
    xyzz29_rec *in, *out;
    hipMalloc(&in, sizeof(xyzz29_rec) * (size_t)n_in); hipMalloc(&out, sizeof(xyzz29_rec) * 64 * blocks);
    k_fill<CV><<<n_in / 256, 256>>>(in, n_in);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_walk<CV, F, PREFETCH><<<blocks, 64>>>(K, in, n_in, out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k_walk<CV, F, PREFETCH><<<blocks, 64>>>(K, in, n_in, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-62s %8.3f us per record   (%d waves)\n", name, 1e3 * best / K, blocks);
    hipFree(in); hipFree(out);
}

// the same chain in blocks of T threads: where does the dispatcher put the waves of B blocks?
template <class CV, class F, int T>
__global__ __launch_bounds__(T) void k_shape(int K, xyzz29_rec* out) {
    typedef typename f29_of<typename CV::Base>::type F0;
    fe gx, gy;
    for (int i = 0; i < 8; i++) { gx.v[i] = CV::GX_M[i]; gy.v[i] = CV::GY_M[i]; }
    xyzz29 g; g.x = f29_from_std<F0>(gx); g.y = f29_from_std<F0>(gy); g.zz = f29_one<F0>(); g.zzz = f29_one<F0>();
    xyzz29 a = x29_add<F0>(x29_double<F0>(g), g), b = x29_double<F0>(a);
    for (int i = 0; i < K; i++) a = x29_add_quad<F>(a, b);
    x29_store(&out[blockIdx.x * T + threadIdx.x], a);
}
template <class CV, class F, int T>
static void run_shape(int blocks, int K) {
    xyzz29_rec* out;
    hipMalloc(&out, sizeof(xyzz29_rec) * (size_t)T * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_shape<CV, F, T><<<blocks, T>>>(K, out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k_shape<CV, F, T><<<blocks, T>>>(K, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("  %5d blocks x %3d threads = %5d waves: %8.3f us per quad-cooperative addition\n", blocks, T, blocks * T / 64, 1e3 * best / K);
    hipFree(out);
}

template <class CV, class F, int OP>
__global__ __launch_bounds__(64) void k_chain(int K, xyzz29_rec* out, unsigned long long* cyc) {
    typedef typename f29_of<typename CV::Base>::type F0;
    fe gx, gy;
    for (int i = 0; i < 8; i++) { gx.v[i] = CV::GX_M[i]; gy.v[i] = CV::GY_M[i]; }
    xyzz29 g; g.x = f29_from_std<F0>(gx); g.y = f29_from_std<F0>(gy); g.zz = f29_one<F0>(); g.zzz = f29_one<F0>();
    xyzz29 a = x29_double<F0>(g);
    a = x29_add<F0>(a, g);                       // 3G
    xyzz29 b = x29_double<F0>(a);                // 6G
    aff29 q; q.x = g.x; q.y = g.y;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < K; i++) {
        if (OP == 0) a = x29_add_quad<F>(a, b);
        else if (OP == 1) a = x29_double_quad<F>(a);
        else if (OP == 2) a = x29_add<F>(a, b);
        else if (OP == 3) a = x29_double<F>(a);
        else if (OP == 4) a = x29_add_mixed<F>(a, q);
        else { a.x = f29_mul<F>(a.x, b.x); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    x29_store(&out[blockIdx.x * 64 + threadIdx.x], a);
}

template <class CV, class F, int OP>
static void run(const char* name, int K) {
    const int blocks = 1024;
    xyzz29_rec* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(xyzz29_rec) * 64 * blocks); hipMalloc(&cyc, 8 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_chain<CV, F, OP><<<blocks, 64>>>(K, out, cyc);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k_chain<CV, F, OP><<<blocks, 64>>>(K, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long h[1024]; hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int i = 0; i < blocks; i++) if (h[i] > mx) mx = h[i];
    printf("%-52s %8.3f us per operation   (%6.0f shader-clock ticks of s_memtime)\n", name, 1e3 * best / K, (double)mx / K);
    hipFree(out); hipFree(cyc);
}

int main() {
    typedef f29_of<Bn254Fq>::type F;
    typedef f29_lat<F> FL;
    const int K = 400;
    printf("bn256 G1, one wave per SIMD (1024 waves of 64 lanes), chain of %d dependent operations per lane / quad\n", K);
    run<CurveBn254, FL, 0>("quad-cooperative addition, latency schedule", K);
    run<CurveBn254, F, 0>("quad-cooperative addition, throughput schedule", K);
    run<CurveBn254, FL, 1>("quad-cooperative doubling, latency schedule", K);
    run<CurveBn254, F, 1>("quad-cooperative doubling, throughput schedule", K);
    run<CurveBn254, FL, 2>("full addition on one lane, latency schedule", K);
    run<CurveBn254, F, 2>("full addition on one lane, throughput schedule", K);
    run<CurveBn254, FL, 3>("doubling on one lane, latency schedule", K);
    run<CurveBn254, F, 4>("mixed addition on one lane, throughput schedule", K);
    run<CurveBn254, FL, 5>("one multiplication, latency schedule", K);
    run<CurveBn254, F, 5>("one multiplication, throughput schedule", K);
    printf("a quad walking records in global memory (the merge kernels' loop), latency schedule, chain of %d\n", 200);
    run_walk<CurveBn254, FL, 0>("load, then add (as the kernels do)", 200, 1024);
    run_walk<CurveBn254, FL, 1>("next record loaded before the addition", 200, 1024);
    run_walk<CurveBn254, FL, 0>("load, then add; 64 waves on the chip", 200, 64);
    run_walk<CurveBn254, FL, 1>("next record loaded before the addition; 64 waves", 200, 64);
    run_walk<CurveBn254, FL, 0>("load, then add; 2 waves per SIMD", 200, 2048);
    printf("placement: the same chain of quad-cooperative additions (latency schedule) by block shape\n");
    run_shape<CurveBn254, FL, 64>(256, 200); run_shape<CurveBn254, FL, 64>(512, 200); run_shape<CurveBn254, FL, 64>(1024, 200); run_shape<CurveBn254, FL, 64>(2048, 200);
    run_shape<CurveBn254, FL, 128>(128, 200); run_shape<CurveBn254, FL, 128>(256, 200); run_shape<CurveBn254, FL, 128>(512, 200); run_shape<CurveBn254, FL, 128>(1024, 200);
    run_shape<CurveBn254, FL, 256>(64, 200); run_shape<CurveBn254, FL, 256>(128, 200); run_shape<CurveBn254, FL, 256>(256, 200); run_shape<CurveBn254, FL, 256>(512, 200);
    run_shape<CurveBn254, FL, 512>(64, 200); run_shape<CurveBn254, FL, 512>(128, 200); run_shape<CurveBn254, FL, 512>(256, 200);
    return 0;
}
