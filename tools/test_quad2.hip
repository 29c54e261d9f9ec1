#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec29.cuh"
__device__ bool eq(const f29& a, const f29& b) { u32 d = 0; for (int i = 0; i < 9; i++) d |= a.v[i] ^ b.v[i]; return d == 0; }
template <class CV> __global__ void k(int* flags) {
    typedef typename f29_of<typename CV::Base>::type F;
    fe gx, gy; for (int i = 0; i < 8; i++) { gx.v[i] = CV::GX_M[i]; gy.v[i] = CV::GY_M[i]; }
    xyzz29 g; g.x = f29_from_std<F>(gx); g.y = f29_from_std<F>(gy); g.zz = f29_one<F>(); g.zzz = f29_one<F>();
    xyzz29 a = x29_double<F>(g), b = x29_add<F>(x29_double<F>(g), g);
    const u32 role = threadIdx.x & 3;
    int f = 0;
    f29 u1 = f29_mul<F>(a.x, b.zz), u2 = f29_mul<F>(b.x, a.zz), s1 = f29_mul<F>(a.y, b.zzz), s2 = f29_mul<F>(b.y, a.zzz);
    f29 m = f29_mul<F>(f29_sel4(a.x, b.x, a.y, b.y, role), f29_sel4(b.zz, a.zz, b.zzz, a.zzz, role));
    f29 q1 = f29_quad_bcast<0>(m), q2 = f29_quad_bcast<1>(m), q3 = f29_quad_bcast<2>(m), q4 = f29_quad_bcast<3>(m);
    if (!eq(u1, q1)) f |= 1; if (!eq(u2, q2)) f |= 2; if (!eq(s1, q3)) f |= 4; if (!eq(s2, q4)) f |= 8;
    f29 p = f29_norm(f29_sub(u2, u1, F::KM)), rr = f29_norm(f29_sub(s2, s1, F::KM));
    f29 pp = f29_sqr<F>(p), r2 = f29_sqr<F>(rr), zz12 = f29_mul<F>(a.zz, b.zz), zzz12 = f29_mul<F>(a.zzz, b.zzz);
    m = f29_mul<F>(f29_sel4(p, rr, a.zz, a.zzz, role), f29_sel4(p, rr, b.zz, b.zzz, role));
    if (!eq(pp, f29_quad_bcast<0>(m))) f |= 16; if (!eq(r2, f29_quad_bcast<1>(m))) f |= 32;
    if (!eq(zz12, f29_quad_bcast<2>(m))) f |= 64; if (!eq(zzz12, f29_quad_bcast<3>(m))) f |= 128;
    // level 3 / 4 and outputs
    f29 ppp = f29_mul<F>(p, pp), qq = f29_mul<F>(u1, pp);
    f29 keep = m;
    m = f29_mul<F>(f29_sel4(p, u1, keep, p, role), pp);
    if (!eq(ppp, f29_quad_bcast<0>(m))) f |= 256; if (!eq(qq, f29_quad_bcast<1>(m))) f |= 512;
    f29 zz3 = f29_mul<F>(zz12, pp);
    if (!eq(zz3, f29_quad_bcast<2>(m))) f |= 1024;
    f29 x3 = f29_norm(f29_sub(r2, f29_add(ppp, f29_dbl(qq)), F::KB));
    f29 t = f29_sub(qq, x3, F::KA);
    m = f29_mul<F>(f29_sel4(rr, s1, rr, keep, role), f29_sel4(t, ppp, t, ppp, role));
    if (!eq(f29_mul<F>(rr, t), f29_quad_bcast<0>(m))) f |= 2048;
    if (!eq(f29_mul<F>(s1, ppp), f29_quad_bcast<1>(m))) f |= 4096;
    if (!eq(f29_mul<F>(zzz12, ppp), f29_quad_bcast<3>(m))) f |= 8192;
    xyzz29 r1 = x29_add<F>(a, b), r2q = x29_add_quad<F>(a, b);
    if (!eq(r1.x, r2q.x)) f |= 1 << 14; if (!eq(r1.y, r2q.y)) f |= 1 << 15; if (!eq(r1.zz, r2q.zz)) f |= 1 << 16; if (!eq(r1.zzz, r2q.zzz)) f |= 1 << 17;
    xyzz29 d1 = x29_double<F>(a), d2 = x29_double_quad<F>(a);
    if (!eq(d1.x, d2.x)) f |= 1 << 18; if (!eq(d1.y, d2.y)) f |= 1 << 19; if (!eq(d1.zz, d2.zz)) f |= 1 << 20; if (!eq(d1.zzz, d2.zzz)) f |= 1 << 21;
    {   // double, level by level
        f29 u = f29_dbl(a.y);
        f29 v = f29_sqr<F>(u), w = f29_mul<F>(u, v), s = f29_mul<F>(a.x, v), xx = f29_sqr<F>(a.x);
        f29 M = f29_norm(f29_add(f29_dbl(xx), xx));
        f29 mm = f29_sqr<F>(M);
        f29 X3 = f29_norm(f29_sub(mm, f29_dbl(s), F::KB));
        f29 T = f29_sub(s, X3, F::KA);
        f29 mt = f29_mul<F>(M, T), wy = f29_mul<F>(w, a.y);
        f29 mq = f29_mul<F>(f29_sel4(M, w, w, w, role), f29_sel4(T, a.y, a.zzz, a.y, role));
        if (!eq(mt, f29_quad_bcast<0>(mq))) f |= 1 << 22;
        if (!eq(wy, f29_quad_bcast<1>(mq))) f |= 1 << 23;
        f29 y1 = f29_norm(f29_sub(mt, wy, F::KM));
        if (!eq(y1, d1.y)) f |= 1 << 24;
        if (!eq(y1, d2.y)) f |= 1 << 25;
    }
    if (f) atomicOr(flags, f);
}
int main() {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    k<CurveBn254><<<1, 64>>>(d);
    int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("level flags: 0x%x\n", h);
    return 0;
}
