// dev test: quad-cooperative add/double vs the single-lane versions (projective equality)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec29.cuh"
template <class F> __device__ bool same_point(const xyzz29& a, const xyzz29& b) {
    bool ia = f29_is_zero_slow<F>(a.zz), ib = f29_is_zero_slow<F>(b.zz);
    if (ia || ib) return ia == ib;
    f29 l = f29_mul<F>(a.x, b.zz), r = f29_mul<F>(b.x, a.zz);
    f29 l2 = f29_mul<F>(a.y, b.zzz), r2 = f29_mul<F>(b.y, a.zzz);
    return f29_is_zero_slow<F>(f29_norm(f29_sub(l, r, F::KM))) && f29_is_zero_slow<F>(f29_norm(f29_sub(l2, r2, F::KM)));
}
template <class CV> __global__ void k(int* bad, int iters) {
    typedef typename f29_of<typename CV::Base>::type F;
    typedef typename CV::Base FB;
    u32 quad = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    // generator in internal form
    fe gx, gy; for (int i = 0; i < 8; i++) { gx.v[i] = CV::GX_M[i]; gy.v[i] = CV::GY_M[i]; }
    xyzz29 g; g.x = f29_from_std<F>(gx); g.y = f29_from_std<F>(gy); g.zz = f29_one<F>(); g.zzz = f29_one<F>();
    xyzz29 a = g, b = x29_double<F>(g);
    for (u32 i = 0; i < quad % 7; i++) b = x29_add<F>(b, g);
    int nbad = 0;
    for (int it = 0; it < iters; it++) {
        xyzz29 s1 = x29_add<F>(a, b), s2 = x29_add_quad<F>(a, b);
        if (!same_point<F>(s1, s2)) nbad |= 1;
        xyzz29 d1 = x29_double<F>(a), d2 = x29_double_quad<F>(a);
        if (!same_point<F>(d1, d2)) nbad |= 2;
        xyzz29 e1 = x29_add<F>(a, a), e2 = x29_add_quad<F>(a, a);   // P + P
        if (!same_point<F>(e1, e2) || !same_point<F>(e1, d1)) nbad |= 4;
        xyzz29 id = x29_identity();
        if (!same_point<F>(x29_add_quad<F>(a, id), a) || !same_point<F>(x29_add_quad<F>(id, a), a) || !same_point<F>(x29_add_quad<F>(id, id), id)) nbad |= 8;
        if (!same_point<F>(x29_double_quad<F>(id), id)) nbad |= 16;
        a = s2; b = d2;
    }
    if (nbad) atomicOr(bad, nbad);
}
int main() {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    k<CurveBn254><<<2, 128>>>(d, 20); k<CurvePallas><<<2, 128>>>(d, 20);
    int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("quad test flags: %d (0 = ok)\n", h);
    // partial wave: 32 threads
    hipMemset(d, 0, 4); k<CurveBn254><<<1, 32>>>(d, 5); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("quad test (32 threads) flags: %d\n", h);
    return 0;
}
