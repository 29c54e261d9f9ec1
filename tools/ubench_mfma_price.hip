// ubench_mfma_price.hip -- prices VERDICT r2 item 5's question: could the Montgomery-REDUCTION half of the field multiplication (the m * p products:
// 81 of BN254's 162 multiply-adds) move to the matrix cores as a lane-batched constant-matrix product on i8 MFMA?
// An MFMA formulation needs the low half of a * b as 8-bit digits (33 of them), returns m = T_lo * p' and m * p as i32 column sums (33 + 66 columns)
// that must be carry-propagated back into bytes / 29-bit limbs by the vector ALU.  This benchmark gives the matrix cores AND the lane <-> matrix-row
// shuffles for FREE and measures only that vector-ALU plumbing next to the half it would replace:
//   A  f29_mul            the shipped multiplication (product + reduction, 162 multiply-adds)
//   B  product only       a * b into 18 normalised limbs (81 multiply-adds): what stays on the vector ALU either way
//   C  B + plumbing       B, digit split, carry propagation of 33 + 66 i32 columns, limb merge -- the MFMA variant with a free MFMA
// MFMA pays only if C is clearly faster than A.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I delay-encryption-in-halo2_amd/csrc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ec29.cuh"

template <class F>
FP_DEV void product18(const f29& a, const f29& b, u32 (&T)[18]) {      // product scanning, no reduction
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) if (k - i >= 0 && k - i < 9) { acc = mad_wide(a.v[i], b.v[k - i], acc); asm("" : "+v"(acc)); }
        T[k] = (u32)acc & F29_MASK;
        acc >>= F29_BITS;
    }
    T[17] = (u32)acc;
}
template <class F9, int V>
__global__ void k_bench(f29* out, const fe* in, int iters) {
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    f29 m = f29_unpack(f_load(&in[gid]));
    m.v[8] &= 0xffff;
    f29 x[2] = {m, m};
    x[1].v[0] ^= 5;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (V == 0) { x[c] = f29_mul<F9>(x[c], m); continue; }
            u32 T[18];
            product18<F9>(x[c], m, T);
            if (V == 2) {
                // digit split of T[0..8] (261 bits) into 33 bytes
                u32 d[33];
#pragma unroll
                for (int k = 0; k < 33; k++) {
                    const int bit = 8 * k, l = bit / 29, o = bit % 29;
                    u32 v = T[l] >> o;
                    if (o > 21 && l + 1 < 9) v |= T[l + 1] << (29 - o);
                    d[k] = v & 255u;
                }
                // (MFMA #1 free: m's column sums) -> carry propagation into 33 bytes
                u32 mb[33], c1 = 0;
#pragma unroll
                for (int k = 0; k < 33; k++) { u32 col = d[k]; asm("" : "+v"(col)); c1 += col; mb[k] = c1 & 255u; c1 >>= 8; }
                // (MFMA #2 free: m * p column sums, 66 columns) -> carry through the low half, bytes of the high half -> 29-bit limbs added to T[9..17]
                u32 c2 = 0;
#pragma unroll
                for (int k = 0; k < 33; k++) { u32 col = mb[k]; asm("" : "+v"(col)); c2 = (c2 + col) >> 8; }
                u32 hb[33];
#pragma unroll
                for (int k = 0; k < 33; k++) { u32 col = mb[32 - k]; asm("" : "+v"(col)); c2 += col; hb[k] = c2 & 255u; c2 >>= 8; }
#pragma unroll
                for (int l = 0; l < 9; l++) {
                    u32 v = 0;
#pragma unroll
                    for (int k = 0; k < 33; k++) {
                        const int lo = 8 * k - 29 * l;      // bit offset of byte k inside limb l
                        if (lo > -8 && lo < 29) v |= lo >= 0 ? (hb[k] << lo) : (hb[k] >> -lo);
                    }
                    T[9 + l] += v & F29_MASK;
                }
            }
            f29 r;
#pragma unroll
            for (int l = 0; l < 9; l++) r.v[l] = T[9 + l];
            x[c] = f29_norm(r);
            x[c].v[8] &= 0xffff;
        }
    }
    out[gid] = f29_add(x[0], x[1]);
}
template <class K>
float time_it(K k, f29* out, const fe* in, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, out, in, 8);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(a); hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, out, in, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    typedef f29_of<Bn254Fr>::type F9;
    const size_t n = 4096 * 256;
    std::vector<fe> h(n);
    for (size_t i = 0; i < n; i++) for (int w = 0; w < 8; w++) h[i].v[w] = (u32)(i * 2654435761u + w * 40503u) & (w == 7 ? 0x0fffffffu : 0xffffffffu);
    fe* in; f29* out;
    hipMalloc(&in, n * sizeof(fe)); hipMalloc(&out, n * sizeof(f29));
    hipMemcpy(in, h.data(), n * sizeof(fe), hipMemcpyHostToDevice);
    const int iters = 400;
    const double muls = 2.0 * n * iters;
    const float a = time_it(k_bench<F9, 0>, out, in, iters), b = time_it(k_bench<F9, 1>, out, in, iters), c = time_it(k_bench<F9, 2>, out, in, iters);
    printf("bn256::Fr, %zu lanes x %d x 2 chains\n", n, iters);
    printf("A  f29_mul (product + reduction)            %8.3f ms  %7.1f Gmul/s\n", a, muls / a / 1e6);
    printf("B  product only (81 multiply-adds)          %8.3f ms  %7.1f G/s   -> the reduction half costs %.3f ms\n", b, muls / b / 1e6, a - b);
    printf("C  B + MFMA plumbing (matrix cores free)    %8.3f ms  %7.1f G/s   -> the plumbing alone costs %.3f ms = %.2f x the half it would replace\n", c, muls / c / 1e6,
           c - b, (c - b) / (a - b));
    printf("speed-up of the full multiplication with a free MFMA and free shuffles: %.2f x (VERDICT r2 bar: >= 1.3)\n", a / c);
    return 0;
}
