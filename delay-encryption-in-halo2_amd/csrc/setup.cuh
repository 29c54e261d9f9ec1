// setup.cuh -- ParamsKZG::setup on the device (the reference's benches build their SRS with `ParamsKZG::<Bn256>::setup(K, OsRng)`,
// benches/delay_enc.rs:43, mod_pow.rs:131, pose_enc.rs:48; upstream halo2_proofs/src/poly/kzg/commitment.rs @ v2023_04_20):
//     g[i] = [s^i] G,      g_lagrange[i] = [L_i(s)] G,   L_i(s) = omega^i (s^n - 1) / (n (s - omega^i)),      g2 = G2,  s_g2 = [s] G2.
// 2 n fixed-base multiplications of ONE point: a table T[w][d] = d * 2^(8 w) * G (32 byte-windows x 256 entries, 512 KB, L2-resident, in the
// kernels' packed internal form like the MSM tables) turns each into <= 32 mixed additions on the carry-free multiplier; the scalars
// (powers of s; omega^i, one batch inversion and two multiplications for L_i(s)) are made on the device too.  The two G2 points are O(1) host work.
#pragma once
#include "ec29.cuh"
#include "internal.hpp"

#define FB_WINDOWS 32

// T[w][d] = d * 2^(8 w) * G  (d = 0: the all-zero identity entry)
template <class CV>
__global__ __launch_bounds__(64) void k_fb_table(affine_t* table) {
    typedef typename CV::Base F;
    typedef typename f29_of<F>::type F9;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_WINDOWS * 256) return;
    const u32 w = t >> 8, d = t & 255;
    affine_t o;
    o.x = f_zero(); o.y = f_zero();
    if (d) {
        affine_t g;
#pragma unroll
        for (int i = 0; i < 8; i++) { g.x.v[i] = CV::GX_M[i]; g.y.v[i] = CV::GY_M[i]; }
        xyzz_t q = xyzz_from_affine<F>(g);
        for (u32 i = 0; i < 8 * w; i++) q = xyzz_double<F>(q);
        xyzz_t r = xyzz_identity();
        for (int bit = 7; bit >= 0; bit--) {
            r = xyzz_double<F>(r);
            if ((d >> bit) & 1) r = xyzz_add<F>(r, q);
        }
        const affine_t a = xyzz_to_affine<F>(r);
        o.x = f29_to_packed_canon<F9>(f29_from_std<F9>(a.x));
        o.y = f29_to_packed_canon<F9>(f29_from_std<F9>(a.y));
    }
    aff_store(&table[t], o);
}

// out[i] = [scalars[i]] G as an affine point in standard form ((0, 0) for a zero scalar)
template <class CV>
__global__ __launch_bounds__(128) void k_fb_mul(const fe* scalars, const affine_t* table, affine_t* out, u64 n) {
    typedef typename f29_of<typename CV::Base>::type F;
    typedef typename CV::Scalar FS;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe s = f_from_mont<FS>(f_load(&scalars[i]));
    xyzz29 acc = x29_identity();
    bool have = false;
    for (u32 w = 0; w < FB_WINDOWS; w++) {
        const u32 d = (s.v[w >> 2] >> (8 * (w & 3))) & 255u;
        if (!d) continue;
        const aff29 q = a29_from_packed(aff_load(&table[w * 256 + d]));
        if (!have) { acc = x29_from_affine<F>(q, false); have = true; }
        else acc = x29_add_mixed<F>(acc, q);
    }
    msm_emit<F>(acc, nullptr, &out[i]);
}

// out[i] = base^i, standard Montgomery form; thread t fills a run of 64 from base^(64 t)
template <class FS>
__global__ void k_powers_std(fe* out, fe base, u64 n) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 start = t * 64;
    if (start >= n) return;
    fe acc = f_one<FS>(), b = base;
    for (u64 e = start; e; e >>= 1) {
        if (e & 1) acc = f_mul<FS>(acc, b);
        b = f_sqr<FS>(b);
    }
    const u64 end = start + 64 < n ? start + 64 : n;
    for (u64 j = start; j < end; j++) {
        f_store(&out[j], acc);
        acc = f_mul<FS>(acc, base);
    }
}
template <class FS>
__global__ void k_lag_den(const fe* w, fe s, fe* den, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f_store(&den[i], f_sub<FS>(s, f_load(&w[i])));
}
template <class FS>
__global__ void k_lag_fin(const fe* inv, const fe* w, fe c, fe* out, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f_store(&out[i], f_mul<FS>(f_mul<FS>(f_load(&inv[i]), f_load(&w[i])), c));
}

// d_g, d_gl: n affine points each (device).  s, omega, cfac = (s^n - 1) / n: Montgomery scalars from the host.
template <class CV>
int kzg_setup_t(dehalo_ctx* ctx, uint32_t k, const uint64_t s[4], const uint64_t omega[4], const uint64_t cfac[4], affine_t* d_g, affine_t* d_gl, hipStream_t st) {
    typedef typename CV::Scalar FS;
    const u64 n = 1ull << k;
    affine_t* table = nullptr;
    fe *pw = nullptr, *w = nullptr, *lag = nullptr;
    auto cleanup = [&]() { (void)hipFree(table); (void)hipFree(pw); (void)hipFree(w); (void)hipFree(lag); };
    hipError_t e = hipMalloc((void**)&table, FB_WINDOWS * 256 * sizeof(affine_t));
    if (e == hipSuccess) e = hipMalloc((void**)&pw, n * sizeof(fe));
    if (e == hipSuccess) e = hipMalloc((void**)&w, n * sizeof(fe));
    if (e == hipSuccess) e = hipMalloc((void**)&lag, n * sizeof(fe));
    if (e != hipSuccess) { cleanup(); return dh_fail(ctx, DEHALO_ERR_OOM, std::string("params_setup: ") + hipGetErrorString(e)); }
    const unsigned runs = (unsigned)((n + 63) / 64), pb = (runs + 127) / 128, eb = (unsigned)((n + 255) / 256);
    k_fb_table<CV><<<FB_WINDOWS * 256 / 64, 64, 0, st>>>(table);
    k_powers_std<FS><<<pb, 128, 0, st>>>(pw, fe_from_u64(s), n);
    k_powers_std<FS><<<pb, 128, 0, st>>>(w, fe_from_u64(omega), n);
    k_lag_den<FS><<<eb, 256, 0, st>>>(w, fe_from_u64(s), lag, n);
    int rc = hipGetLastError() == hipSuccess ? 0 : dh_fail(ctx, DEHALO_ERR_HIP, "params_setup: launch failed");
    if (rc == 0) rc = dehalo_batch_invert_device(ctx, FS::ID, (uint64_t*)lag, n, st);
    if (rc == 0) {
        k_lag_fin<FS><<<eb, 256, 0, st>>>(lag, w, fe_from_u64(cfac), lag, n);
        k_fb_mul<CV><<<(unsigned)((n + 127) / 128), 128, 0, st>>>(pw, table, d_g, n);
        k_fb_mul<CV><<<(unsigned)((n + 127) / 128), 128, 0, st>>>(lag, table, d_gl, n);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = dh_fail(ctx, DEHALO_ERR_HIP, "params_setup: kernels failed");
    }
    cleanup();
    return rc;
}
