// ubench.hip -- gfx950 instruction-rate and field-arithmetic microbenchmarks (dev tool).
// Build: hipcc -O3 --offload-arch=gfx950 -I delay-encryption-in-halo2_amd/csrc tools/ubench.hip -o gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ec29.cuh"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

#define DEF_INT_BENCH(NAME, ASM)                                                            \
__global__ void NAME(u32* out, u32 seed) {                                                  \
    u32 a = seed + threadIdx.x, b = seed * 3 + 1;                                           \
    u64 r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7; \
    for (int i = 0; i < ITERS; i++) {                                                       \
        ASM(r0) ASM(r1) ASM(r2) ASM(r3) ASM(r4) ASM(r5) ASM(r6) ASM(r7)                     \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7); \
}
#define MAD64(r) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(a), "v"(b) : "vcc");
#define MULLO(r) { u32 t = (u32)r; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(b)); r = t; }
#define MULHI(r) { u32 t = (u32)r; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t) : "v"(b)); r = t; }
#define MAD24(r) { u32 t = (u32)r; asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(t) : "v"(b), "v"(a)); r = t; }
#define ADD32(r) { u32 t = (u32)r; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(b)); r = t; }
#define ADD64(r) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"((u64)b));
#define ADDC(r) { u32 lo = (u32)r, hi = (u32)(r >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc"); r = ((u64)hi << 32) | lo; }
#define SHR64(r) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(r));
#define SHL64(r) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(r));
#define ALIGNB(r) { u32 t = (u32)r; asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(t) : "v"(b)); r = t; }
#define AND32(r) { u32 t = (u32)r; asm volatile("v_and_b32 %0, %0, %1" : "+v"(t) : "v"(b)); r = t; }
#define MOV32(r) { u32 t = (u32)r; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(b)); r = t; }
#define LSHLADD64S(r) asm volatile("v_lshl_add_u64 %0, %1, 3, %0" : "+v"(r) : "v"((u64)b));
DEF_INT_BENCH(k_shr64, SHR64)
DEF_INT_BENCH(k_shl64, SHL64)
DEF_INT_BENCH(k_alignb, ALIGNB)
DEF_INT_BENCH(k_and32, AND32)
DEF_INT_BENCH(k_lshladd3, LSHLADD64S)
DEF_INT_BENCH(k_mad64, MAD64)
DEF_INT_BENCH(k_mullo, MULLO)
DEF_INT_BENCH(k_mulhi, MULHI)
DEF_INT_BENCH(k_mad24, MAD24)
DEF_INT_BENCH(k_add32, ADD32)
DEF_INT_BENCH(k_add64, ADD64)
DEF_INT_BENCH(k_addc, ADDC)

__global__ void k_fma64(double* out, double seed) {
    double a = seed + threadIdx.x, b = 1.0000001;
    double r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    for (int i = 0; i < ITERS; i++) {
#define FMA(r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(a));
        FMA(r0) FMA(r1) FMA(r2) FMA(r3) FMA(r4) FMA(r5) FMA(r6) FMA(r7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

template <class F, int CH>
__global__ void k_fmul(fe* out, const fe* in, int iters) {
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    fe x[CH];
    fe m = f_load(&in[gid]);
#pragma unroll
    for (int c = 0; c < CH; c++) { x[c] = m; x[c].v[0] += c; x[c].v[7] &= 0x0fffffff; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = f_mul<F>(x[c], m);
    }
    fe r = x[0];
#pragma unroll
    for (int c = 1; c < CH; c++) r = f_add<F>(r, x[c]);
    f_store(&out[gid], r);
}

template <class F9, int CH>
__global__ void k_fmul29(f29* out, const fe* in, int iters) {
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    f29 x[CH];
    f29 m = f29_unpack(f_load(&in[gid]));
    m.v[8] &= 0xffff;
#pragma unroll
    for (int c = 0; c < CH; c++) { x[c] = m; x[c].v[0] ^= c; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = f29_mul<F9>(x[c], m);
    }
    f29 r = x[0];
#pragma unroll
    for (int c = 1; c < CH; c++) r = f29_add(r, x[c]);
    out[gid] = r;
}
template <class F9>
__global__ void k_fsqr29(f29* out, const fe* in, int iters) {
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    f29 x = f29_unpack(f_load(&in[gid]));
    x.v[8] &= 0xffff;
    for (int i = 0; i < iters; i++) x = f29_sqr<F9>(x);
    out[gid] = x;
}

template <class F9>
__global__ void k_madd29(xyzz29_rec* out, const affine_t* pts, int iters, u32 npts) {
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 idx = gid % npts;
    affine_t pk = aff_load(&pts[idx]);
    aff29 q = a29_from_packed(pk);
    xyzz29 acc; acc.x = q.x; acc.y = q.y; acc.zz = f29_one<F9>(); acc.zzz = f29_one<F9>();
    for (int i = 0; i < iters; i++) {
        idx = (idx * 1664525u + 1013904223u) % npts;
        pk = aff_load(&pts[idx]);
        q = a29_from_packed(pk);
        acc = x29_add_mixed<F9>(acc, q);
    }
    x29_store(&out[gid], acc);
}

template <class F>
__global__ void k_madd(xyzz_t* out, const affine_t* pts, int iters, u32 npts) {
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    xyzz_t acc = xyzz_identity();
    u32 idx = gid % npts;
    for (int i = 0; i < iters; i++) {
        affine_t q = aff_load(&pts[idx]);
        acc = xyzz_add_mixed<F>(acc, q);
        idx = (idx * 1664525u + 1013904223u) % npts;
    }
    xyzz_store(&out[gid], acc);
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
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int CUS = p.multiProcessorCount;
    const int blocks = CUS * 8, threads = 256;  // 8 waves/SIMD
    u32* d; CHECK(hipMalloc(&d, (size_t)blocks * threads * 8));
    double ops = (double)blocks * threads * ITERS * 8;
    struct { const char* n; float ms; } rows[] = {
        {"v_mad_u64_u32", time_kernel(k_mad64, dim3(blocks), dim3(threads), d, 7u)},
        {"v_mul_lo_u32", time_kernel(k_mullo, dim3(blocks), dim3(threads), d, 7u)},
        {"v_mul_hi_u32", time_kernel(k_mulhi, dim3(blocks), dim3(threads), d, 7u)},
        {"v_mad_u32_u24", time_kernel(k_mad24, dim3(blocks), dim3(threads), d, 7u)},
        {"v_add_u32", time_kernel(k_add32, dim3(blocks), dim3(threads), d, 7u)},
        {"v_lshl_add_u64", time_kernel(k_add64, dim3(blocks), dim3(threads), d, 7u)},
        {"add_co+addc (pair)", time_kernel(k_addc, dim3(blocks), dim3(threads), d, 7u)},
        {"v_fma_f64", time_kernel(k_fma64, dim3(blocks), dim3(threads), (double*)d, 1.5)},
        {"v_lshrrev_b64", time_kernel(k_shr64, dim3(blocks), dim3(threads), d, 7u)},
        {"v_lshlrev_b64", time_kernel(k_shl64, dim3(blocks), dim3(threads), d, 7u)},
        {"v_alignbit_b32", time_kernel(k_alignb, dim3(blocks), dim3(threads), d, 7u)},
        {"v_and_b32", time_kernel(k_and32, dim3(blocks), dim3(threads), d, 7u)},
        {"v_lshl_add_u64 (<<3)", time_kernel(k_lshladd3, dim3(blocks), dim3(threads), d, 7u)},
    };
    for (auto& r : rows) {
        double rate = ops / (r.ms * 1e-3);
        // cycles per wave-instruction per SIMD at 2.4 GHz: SIMDs = CUS*4
        double cyc = 2.4e9 * (CUS * 4.0) / (rate / 64.0);
        printf("%-22s %8.3f ms  %8.2f Tops/s  ~%5.2f cyc/wave-instr/SIMD @2.4GHz\n", r.n, r.ms, rate / 1e12, cyc);
    }
    // field mul throughput
    const int fthreads = 128, fblocks = CUS * 16;
    size_t nel = (size_t)fblocks * fthreads;
    fe *fin, *fout; CHECK(hipMalloc(&fin, nel * 32)); CHECK(hipMalloc(&fout, nel * 32));
    std::vector<u32> h(nel * 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffff : 0xffffffff);
    CHECK(hipMemcpy(fin, h.data(), nel * 32, hipMemcpyHostToDevice));
    const int fit = 2000;
    float ms;
    ms = time_kernel(k_fmul<PastaFp, 1>, dim3(fblocks), dim3(fthreads), fout, fin, fit);
    printf("f_mul<PastaFp> chain1  %8.3f ms  %8.2f Gmul/s\n", ms, (double)nel * fit / ms / 1e6);
    ms = time_kernel(k_fmul<PastaFp, 2>, dim3(fblocks), dim3(fthreads), fout, fin, fit);
    printf("f_mul<PastaFp> chain2  %8.3f ms  %8.2f Gmul/s\n", ms, (double)nel * fit * 2 / ms / 1e6);
    ms = time_kernel(k_fmul<Bn254Fr, 1>, dim3(fblocks), dim3(fthreads), fout, fin, fit);
    printf("f_mul<Bn254Fr> chain1  %8.3f ms  %8.2f Gmul/s\n", ms, (double)nel * fit / ms / 1e6);
    ms = time_kernel(k_fmul<Bn254Fq, 2>, dim3(fblocks), dim3(fthreads), fout, fin, fit);
    printf("f_mul<Bn254Fq> chain2  %8.3f ms  %8.2f Gmul/s\n", ms, (double)nel * fit * 2 / ms / 1e6);
    f29* f29out; CHECK(hipMalloc(&f29out, nel * sizeof(f29)));
    ms = time_kernel(k_fmul29<PastaFp29, 1>, dim3(fblocks), dim3(fthreads), f29out, fin, fit);
    printf("f29_mul<PastaFp29> ch1 %8.3f ms  %8.2f Gmul/s\n", ms, (double)nel * fit / ms / 1e6);
    ms = time_kernel(k_fmul29<PastaFp29, 2>, dim3(fblocks), dim3(fthreads), f29out, fin, fit);
    printf("f29_mul<PastaFp29> ch2 %8.3f ms  %8.2f Gmul/s\n", ms, (double)nel * fit * 2 / ms / 1e6);
    ms = time_kernel(k_fmul29<Bn254Fq29, 1>, dim3(fblocks), dim3(fthreads), f29out, fin, fit);
    printf("f29_mul<Bn254Fq29> ch1 %8.3f ms  %8.2f Gmul/s\n", ms, (double)nel * fit / ms / 1e6);
    ms = time_kernel(k_fsqr29<PastaFp29>, dim3(fblocks), dim3(fthreads), f29out, fin, fit);
    printf("f29_sqr<PastaFp29>     %8.3f ms  %8.2f Gsqr/s\n", ms, (double)nel * fit / ms / 1e6);
    // mixed-add throughput with random gathers from a 64 MB table
    u32 npts = 1 << 20;
    affine_t* pts; CHECK(hipMalloc(&pts, (size_t)npts * 64));
    std::vector<u32> hp((size_t)npts * 16);
    for (size_t i = 0; i < hp.size(); i++) hp[i] = (u32)(i * 2246822519u + 12345) & ((i % 8 == 7) ? 0x0fffffff : 0xffffffff);
    CHECK(hipMemcpy(pts, hp.data(), (size_t)npts * 64, hipMemcpyHostToDevice));
    xyzz_t* xo; CHECK(hipMalloc(&xo, nel * 128));
    const int ait = 256;
    {
        // canonical-looking table entries (top word < 2^30) so that unpacked limbs are in range
        xyzz29_rec* xo29; CHECK(hipMalloc(&xo29, nel * sizeof(xyzz29_rec)));
        for (u32 np : {1u << 10, 1u << 20}) {
            ms = time_kernel(k_madd29<PastaFp29>, dim3(fblocks), dim3(fthreads), xo29, pts, ait, np);
            printf("x29_add_mixed<PastaFp29> table %7u pts %8.3f ms  %8.2f Gadd/s\n", np, ms, (double)nel * ait / ms / 1e6);
        }
        ms = time_kernel(k_madd29<Bn254Fq29>, dim3(fblocks), dim3(fthreads), xo29, pts, ait, 1u << 20);
        printf("x29_add_mixed<Bn254Fq29>                    %8.3f ms  %8.2f Gadd/s\n", ms, (double)nel * ait / ms / 1e6);
    }
    ms = time_kernel(k_madd<PastaFp>, dim3(fblocks), dim3(fthreads), xo, pts, ait, npts);
    printf("xyzz_add_mixed<PastaFp> %8.3f ms  %8.2f Gadd/s (garbage points: exercises the generic path)\n", ms, (double)nel * ait / ms / 1e6);
    ms = time_kernel(k_madd<Bn254Fq>, dim3(fblocks), dim3(fthreads), xo, pts, ait, npts);
    printf("xyzz_add_mixed<Bn254Fq> %8.3f ms  %8.2f Gadd/s\n", ms, (double)nel * ait / ms / 1e6);
    return 0;
}
