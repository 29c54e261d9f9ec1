#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec29.cuh"
__global__ void k(u32* out) {
    u32 lane = threadIdx.x;
    f29 a; for (int i = 0; i < 9; i++) a.v[i] = lane * 16 + i;
    f29 b0 = f29_quad_bcast<0>(a), b1 = f29_quad_bcast<1>(a), b2 = f29_quad_bcast<2>(a), b3 = f29_quad_bcast<3>(a);
    out[lane * 8 + 0] = b0.v[0]; out[lane * 8 + 1] = b1.v[0]; out[lane * 8 + 2] = b2.v[0]; out[lane * 8 + 3] = b3.v[8];
    f29 A = f29_zero(), B = f29_zero(), C = f29_zero(), D = f29_zero();
    A.v[3] = 100; B.v[3] = 200; C.v[3] = 300; D.v[3] = 400;
    out[lane * 8 + 4] = f29_sel4(A, B, C, D, lane & 3).v[3];
}
int main() {
    u32* d; hipMalloc(&d, 64 * 8 * 4); k<<<1, 64>>>(d);
    u32 h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 12; l++) printf("lane %2d: b0 %4u b1 %4u b2 %4u b3(limb8) %4u sel %u\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4]);
    return 0;
}
