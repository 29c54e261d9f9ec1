// Latency of the LDS-operand quad-cooperative group operations of msm_bred.cuh (x29q_add_mem / x29q_double_mem) beside the register-operand ones of
// ec29.cuh, one wave per SIMD (1024 waves), each quad running a dependent chain of K operations -- and the same with 2 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I delay-encryption-in-halo2_amd/csrc -o tools/ubench_qmem tools/ubench_qmem.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec29.cuh"
template <class F> __device__ void msm_emit(const xyzz29&, jacobian_t*, affine_t*) {}
__device__ __forceinline__ void msm_tail_prio() {}
#include "msm_bred.cuh"

template <class CV>
__global__ void k_setup(xyzz29_rec* ab) {
    typedef typename f29_of<typename CV::Base>::type F0;
    fe gx, gy;
    for (int i = 0; i < 8; i++) { gx.v[i] = CV::GX_M[i]; gy.v[i] = CV::GY_M[i]; }
    xyzz29 g; g.x = f29_from_std<F0>(gx); g.y = f29_from_std<F0>(gy); g.zz = f29_one<F0>(); g.zzz = f29_one<F0>();
    xyzz29 a = x29_double<F0>(g);
    a = x29_add<F0>(a, g);
    xyzz29 b = x29_double<F0>(a);
    x29_store(&ab[0], a); x29_store(&ab[1], b);
}
template <class CV, class F, int OP, int T>
__global__ __launch_bounds__(T) void k_chain(int K, const xyzz29_rec* ab, xyzz29_rec* out) {
    __shared__ __align__(16) u32 lds[(T / 4) * 2 * 36];
    const u32 quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    u32* ra = lds + 36 * (2 * quad); u32* rb = ra + 36;
    q_copy_in(ra, &ab[0], role); q_copy_in(rb, &ab[1], role);
    __syncthreads();
    if (OP < 2) {
        for (int i = 0; i < K; i++) {
            if (OP == 0) x29q_add_mem<F>(ra, ra, rb);
            else x29q_double_mem<F>(ra, ra);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        q_copy_out(&out[blockIdx.x * (T / 4) + quad], ra, role);
    } else {
        xyzz29 a = x29_load(&ab[0]), b = x29_load(&ab[1]);
        for (int i = 0; i < K; i++) {
            if (OP == 2) a = x29_add_quad<F>(a, b);
            else a = x29_double_quad<F>(a);
        }
        if (role == 0) x29_store(&out[blockIdx.x * (T / 4) + quad], a);
    }
}
template <class CV, class F, int OP, int T>
static void run(const char* name, int K, int blocks) {
    xyzz29_rec *out, *ab;
    hipMalloc(&out, sizeof(xyzz29_rec) * (size_t)T * blocks); hipMalloc(&ab, 2 * sizeof(xyzz29_rec));
    k_setup<CV><<<1, 1>>>(ab);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_chain<CV, F, OP, T><<<blocks, T>>>(K, ab, out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k_chain<CV, F, OP, T><<<blocks, T>>>(K, ab, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    xyzz29_rec h0; hipMemcpy(&h0, out, sizeof(h0), hipMemcpyDeviceToHost);
    printf("%-72s %8.3f us per operation  (%d blocks x %d threads; x limb 0 = %08x)\n", name, 1e3 * best / K, blocks, T, h0.w[0]);
    hipFree(out); hipFree(ab);
}
int main() {
    typedef f29_of<Bn254Fq>::type F;
    typedef f29_lat<F> FL;
    const int K = 300;
    printf("bn256 G1, chains of %d dependent quad-cooperative operations\n", K);
    run<CurveBn254, FL, 2, 256>("register operands (x29_add_quad), latency schedule", K, 256);
    run<CurveBn254, F, 2, 256>("register operands (x29_add_quad), throughput schedule", K, 256);
    run<CurveBn254, FL, 0, 256>("LDS operands (x29q_add_mem), latency schedule", K, 256);
    run<CurveBn254, F, 0, 256>("LDS operands (x29q_add_mem), throughput schedule", K, 256);
    run<CurveBn254, FL, 3, 256>("doubling, register operands, latency schedule", K, 256);
    run<CurveBn254, FL, 1, 256>("doubling, LDS operands, latency schedule", K, 256);
    run<CurveBn254, F, 1, 256>("doubling, LDS operands, throughput schedule", K, 256);
    run<CurveBn254, FL, 0, 256>("LDS operands, latency schedule, 2 waves per SIMD", K, 512);
    run<CurveBn254, F, 0, 256>("LDS operands, throughput schedule, 2 waves per SIMD", K, 512);
    typedef f29_of<PastaFp>::type P;
    typedef f29_lat<P> PL;
    run<CurvePallas, PL, 2, 256>("Pallas: register operands, latency schedule", K, 256);
    run<CurvePallas, PL, 0, 256>("Pallas: LDS operands, latency schedule", K, 256);
    run<CurvePallas, P, 0, 256>("Pallas: LDS operands, throughput schedule", K, 256);
    return 0;
}
