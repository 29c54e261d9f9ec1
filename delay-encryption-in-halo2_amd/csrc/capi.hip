// capi.hip -- C ABI (include/dehalo.h) over the gfx950 kernels in ntt.cuh / msm.cuh.
// Host logic only: argument checks, HBM workspace, pass / window planning, launches,
// HIP-event timing.  No CPU arithmetic path exists here: every field or group operation
// runs on the device, and context creation fails without one.
#include "../../include/dehalo.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "msm.cuh"
#include "ntt.cuh"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct TwiddleEntry {
    int field;
    uint32_t log_n;
    uint64_t omega[4];
    fe* tw;
};

struct TimedRegion {
    int kernel_id;
    hipEvent_t a, b;
};

}  // namespace

struct dehalo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::mutex mu;
    // workspace (grow-only)
    DevBuf ws_scalars, ws_out, ws_count, ws_cursor, ws_off, ws_toff0, ws_cnt1, ws_off1, ws_toff1, ws_bsum, ws_idx, ws_partial0,
        ws_partial1, ws_buckets, ws_contrib, ws_tree, ws_gsums, ws_ntt_scratch, ws_ntt_io, ws_ntt_io2, ws_fop[3], ws_tmp_bases;
    std::vector<TwiddleEntry> twiddles;
    bool timing = false;
    std::vector<TimedRegion> regions;
    double timing_ms[DEHALO_K_COUNT] = {0, 0, 0, 0};
    uint64_t timing_cnt[DEHALO_K_COUNT] = {0, 0, 0, 0};
};

struct dehalo_bases {
    int curve;
    size_t n;
    uint32_t c, W;
    int precomp;
    affine_t* table;  // n * (precomp ? W : 1) affine points in HBM
};

namespace {

int fail(dehalo_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            int code_ = (e_ == hipErrorOutOfMemory) ? DEHALO_ERR_OOM : DEHALO_ERR_HIP;                        \
            return fail(ctx, code_, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
        }                                                                                                    \
    } while (0)

int ensure(dehalo_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

#define TRY(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != 0) return rc_; \
    } while (0)

struct ScopedTimer {
    dehalo_ctx* ctx;
    hipStream_t s;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(dehalo_ctx* c, hipStream_t st, int kid) : ctx(c), s(st), id(kid) {
        if (ctx->timing) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
            (void)hipEventRecord(a, s);
        }
    }
    ~ScopedTimer() {
        if (a && b) {
            (void)hipEventRecord(b, s);
            ctx->regions.push_back({id, a, b});
        }
    }
};

uint32_t log2_ceil(size_t n) {
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

fe fe_from_u64(const uint64_t v[4]) {
    fe r;
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (u32)v[i];
        r.v[2 * i + 1] = (u32)(v[i] >> 32);
    }
    return r;
}

template <class Fn>
int dispatch_field(dehalo_ctx* ctx, int field, Fn f) {
    switch (field) {
        case DEHALO_FIELD_BN254_FR: return f(Bn254Fr{});
        case DEHALO_FIELD_BN254_FQ: return f(Bn254Fq{});
        case DEHALO_FIELD_PASTA_FP: return f(PastaFp{});
        case DEHALO_FIELD_PASTA_FQ: return f(PastaFq{});
        default: return fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    }
}
template <class Fn>
int dispatch_curve(dehalo_ctx* ctx, int curve, Fn f) {
    switch (curve) {
        case DEHALO_CURVE_BN254_G1: return f(CurveBn254{});
        case DEHALO_CURVE_PALLAS: return f(CurvePallas{});
        case DEHALO_CURVE_VESTA: return f(CurveVesta{});
        default: return fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    }
}

// ------------------------------------------------------------------------------------------
// NTT
// ------------------------------------------------------------------------------------------
template <class F>
int get_twiddles(dehalo_ctx* ctx, uint32_t log_n, const uint64_t omega[4], hipStream_t s, const fe** out) {
    for (auto& t : ctx->twiddles)
        if (t.field == F::ID && t.log_n == log_n && !memcmp(t.omega, omega, 32)) {
            *out = t.tw;
            return 0;
        }
    uint64_t half = log_n ? (1ull << (log_n - 1)) : 1;
    fe* tw = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&tw, half * sizeof(fe)));
    uint64_t threads = (half + 63) / 64;
    uint32_t blocks = (uint32_t)((threads + 127) / 128);
    k_twiddle_gen<F><<<blocks, 128, 0, s>>>(tw, fe_from_u64(omega), half);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->twiddles.size() >= 16) {  // bounded cache: drop the oldest
        HIP_TRY(ctx, hipStreamSynchronize(s));
        HIP_TRY(ctx, hipFree(ctx->twiddles.front().tw));
        ctx->twiddles.erase(ctx->twiddles.begin());
    }
    TwiddleEntry e;
    e.field = F::ID; e.log_n = log_n; memcpy(e.omega, omega, 32); e.tw = tw;
    ctx->twiddles.push_back(e);
    *out = tw;
    return 0;
}

struct NttScale {
    uint32_t pre_mode = 0, post_mode = 0;
    fe pre_z{}, post0{}, post_z{};
};

// Transforms `batch` polynomials: src (src_len valid elements each, zero-extended to 2^log_n,
// src_stride apart) -> dst (dst_stride apart).  src == dst allowed.
template <class F>
int run_ntt(dehalo_ctx* ctx, const fe* src, uint64_t src_len, uint64_t src_stride, fe* dst, uint64_t dst_stride, uint32_t log_n,
            const uint64_t omega[4], size_t batch, const NttScale& sc, hipStream_t s) {
    if (log_n > (uint32_t)F::TWO_ADICITY) return fail(ctx, DEHALO_ERR_UNSUPPORTED, "log_n exceeds the field's two-adicity");
    if (log_n > 30) return fail(ctx, DEHALO_ERR_INVALID, "log_n > 30");
    if (batch == 0) return 0;
    const fe* tw = nullptr;
    TRY(get_twiddles<F>(ctx, log_n, omega, s, &tw));
    ScopedTimer timer(ctx, s, DEHALO_K_NTT_PASS);

    // plan: radices
    uint32_t L, rad[NTT_MAX_PASSES];
    if (log_n <= NTT_TILE_LOG) { L = 1; rad[0] = log_n; }
    else {
        L = (log_n + 7) / 8;
        if (L > NTT_MAX_PASSES) return fail(ctx, DEHALO_ERR_INVALID, "log_n too large");
        uint32_t base = log_n / L, extra = log_n % L;
        for (uint32_t i = 0; i < L; i++) rad[i] = base + (i < extra ? 1 : 0);
    }
    const uint64_t N = 1ull << log_n;
    fe* scratch = nullptr;
    if (L > 1) {
        TRY(ensure(ctx, ctx->ws_ntt_scratch, batch * N * sizeof(fe)));
        scratch = (fe*)ctx->ws_ntt_scratch.p;
    }
    static bool attr_set = false;  // one process drives one GPU
    const size_t lds_max = 2 * NTT_TILE * 16 + (NTT_TILE / 2) * sizeof(fe);
    if (!attr_set) {
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr_set = true;
    }
    uint32_t log_m = log_n;
    for (uint32_t p = 0; p < L; p++) {
        NttPassParams P;
        memset(&P, 0, sizeof(P));
        bool first = p == 0, last = p == L - 1;
        P.tw = tw;
        P.log_n = log_n; P.log_m = log_m; P.r = rad[p];
        P.is_final = last ? 1 : 0;
        P.r1 = L > 1 ? rad[0] : 0;
        if (first) { P.src = src; P.src_len = src_len; P.src_stride = src_stride; }
        else { P.src = scratch; P.src_len = N; P.src_stride = N; }
        if (last) { P.dst = dst; P.dst_stride = dst_stride; }
        else { P.dst = scratch; P.dst_stride = N; }
        if (first && sc.pre_mode) { P.pre_mode = sc.pre_mode; P.pre_z = sc.pre_z; }
        if (last && sc.post_mode) { P.post_mode = sc.post_mode; P.post0 = sc.post0; P.post_z = sc.post_z; }
        uint32_t log_c;
        if (!last) {
            uint32_t log_cols = log_m - rad[p];
            log_c = std::min<uint32_t>(NTT_TILE_LOG - rad[p], log_cols);
        } else {
            log_c = std::min<uint32_t>(NTT_TILE_LOG - rad[p], P.r1);
            P.nrev = L > 2 ? L - 2 : 0;
            for (uint32_t i = 0; i < P.nrev; i++) P.rev_r[i] = rad[1 + i];
        }
        P.log_c = log_c;
        uint64_t tiles = N >> (rad[p] + log_c);
        size_t lds = 2 * NTT_TILE * 16 + ((size_t)1 << rad[p]) / 2 * sizeof(fe);
        dim3 grid((uint32_t)tiles, (uint32_t)batch);
        k_ntt_pass<F><<<grid, NTT_THREADS, lds, s>>>(P);
        HIP_TRY(ctx, hipGetLastError());
        log_m -= rad[p];
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// MSM
// ------------------------------------------------------------------------------------------
uint32_t choose_window(size_t n) {
    uint32_t l = log2_ceil(n ? n : 1);
    if (l >= 16) return 16;
    return std::max<uint32_t>(6, l > 1 ? l - 1 : 1);
}

int run_scan(dehalo_ctx* ctx, const u32* cnt, u32 total, u32 L, u32* off, u32* toff, hipStream_t s) {
    u32 nblocks = (total + SCAN_BLOCK - 1) / SCAN_BLOCK;
    TRY(ensure(ctx, ctx->ws_bsum, (size_t)nblocks * 2 * sizeof(u32)));
    u32* bs_i = (u32*)ctx->ws_bsum.p;
    u32* bs_t = bs_i + nblocks;
    k_scan_block_sums<<<nblocks, SCAN_THREADS, 0, s>>>(cnt, total, L, bs_i, bs_t);
    k_scan_top<<<1, SCAN_THREADS, 0, s>>>(bs_i, bs_t, nblocks);
    k_scan_apply<<<nblocks, SCAN_THREADS, 0, s>>>(cnt, total, L, bs_i, bs_t, off, toff);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class CV>
int run_msm(dehalo_ctx* ctx, const dehalo_bases* bases, const fe* d_scalars, size_t len, size_t batch, jacobian_t* d_out, hipStream_t s) {
    typedef typename CV::Scalar FS;
    if (batch == 0) return 0;
    if (len == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, batch * sizeof(jacobian_t), s));
        return 0;
    }
    MsmGeom g;
    g.n = (u32)len; g.table_n = (u32)bases->n; g.c = bases->c; g.W = bases->W; g.nb = 1u << (g.c - 1);
    g.G = bases->precomp ? 1 : g.W;
    g.batch = (u32)batch;
    g.slices = (u32)std::min<size_t>(256, std::max<size_t>(1, len / 2048));
    const uint64_t total_groups = (uint64_t)batch * g.G;
    const uint64_t total_buckets = total_groups * g.nb;
    const uint64_t Mmax = (uint64_t)batch * len * g.W;
    if (Mmax >= (1ull << 32) || total_buckets >= (1ull << 31)) return fail(ctx, DEHALO_ERR_INVALID, "batch * len * windows too large for one launch");
    {   // task length: enough tasks to fill 256 CUs, short enough to balance
        uint64_t l0 = Mmax / (256 * 1024);
        u32 L0 = 4;
        while (L0 < 64 && L0 < l0) L0 <<= 1;
        g.L0 = L0;
    }
    const uint64_t nt0_max = Mmax / g.L0 + total_buckets;
    const uint64_t nt1_max = nt0_max / MSM_L1 + total_buckets;
    const u32 per_group = (g.nb + MSM_RED_M - 1) / MSM_RED_M;

    TRY(ensure(ctx, ctx->ws_count, total_buckets * 4));
    TRY(ensure(ctx, ctx->ws_cursor, total_buckets * 4));
    TRY(ensure(ctx, ctx->ws_off, (total_buckets + 1) * 4));
    TRY(ensure(ctx, ctx->ws_toff0, (total_buckets + 1) * 4));
    TRY(ensure(ctx, ctx->ws_cnt1, total_buckets * 4));
    TRY(ensure(ctx, ctx->ws_off1, (total_buckets + 1) * 4));
    TRY(ensure(ctx, ctx->ws_toff1, (total_buckets + 1) * 4));
    TRY(ensure(ctx, ctx->ws_idx, Mmax * 4));
    TRY(ensure(ctx, ctx->ws_partial0, nt0_max * sizeof(xyzz_t)));
    TRY(ensure(ctx, ctx->ws_partial1, nt1_max * sizeof(xyzz_t)));
    TRY(ensure(ctx, ctx->ws_buckets, total_buckets * sizeof(xyzz_t)));
    TRY(ensure(ctx, ctx->ws_contrib, total_groups * per_group * sizeof(xyzz_t)));
    TRY(ensure(ctx, ctx->ws_tree, total_groups * ((per_group + 2 * MSM_TREE_THREADS - 1) / (2 * MSM_TREE_THREADS)) * sizeof(xyzz_t)));
    TRY(ensure(ctx, ctx->ws_gsums, total_groups * sizeof(xyzz_t)));
    u32* count = (u32*)ctx->ws_count.p;
    u32* cursor = (u32*)ctx->ws_cursor.p;
    u32* off = (u32*)ctx->ws_off.p;
    u32* toff0 = (u32*)ctx->ws_toff0.p;
    u32* cnt1 = (u32*)ctx->ws_cnt1.p;
    u32* off1 = (u32*)ctx->ws_off1.p;
    u32* toff1 = (u32*)ctx->ws_toff1.p;
    u32* idx = (u32*)ctx->ws_idx.p;
    xyzz_t* partial0 = (xyzz_t*)ctx->ws_partial0.p;
    xyzz_t* partial1 = (xyzz_t*)ctx->ws_partial1.p;
    xyzz_t* buckets = (xyzz_t*)ctx->ws_buckets.p;
    xyzz_t* contrib = (xyzz_t*)ctx->ws_contrib.p;
    xyzz_t* tree = (xyzz_t*)ctx->ws_tree.p;
    xyzz_t* gsums = (xyzz_t*)ctx->ws_gsums.p;

    const size_t lds_hist = (size_t)g.nb * 4;
    if (lds_hist > 48 * 1024) {
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_msm_hist<FS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hist));
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_msm_scatter<FS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hist));
    }
    const u32 tb = (u32)total_buckets;
    {
        ScopedTimer t(ctx, s, DEHALO_K_MSM_SORT);
        HIP_TRY(ctx, hipMemsetAsync(count, 0, total_buckets * 4, s));
        HIP_TRY(ctx, hipMemsetAsync(cursor, 0, total_buckets * 4, s));
        dim3 grid(g.slices, g.G, (u32)batch);
        k_msm_hist<FS><<<grid, MSM_SORT_THREADS, lds_hist, s>>>(g, d_scalars, count);
        TRY(run_scan(ctx, count, tb, g.L0, off, toff0, s));
        k_msm_scatter<FS><<<grid, MSM_SORT_THREADS, lds_hist, s>>>(g, d_scalars, off, cursor, idx);
        HIP_TRY(ctx, hipGetLastError());
    }
    {
        ScopedTimer t(ctx, s, DEHALO_K_MSM_ACCUMULATE);
        u32 blocks = (u32)((nt0_max + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS);
        k_msm_accum0<CV><<<blocks, MSM_ACC_THREADS, 0, s>>>(g, tb, idx, off, toff0, bases->table, partial0);
        HIP_TRY(ctx, hipGetLastError());
    }
    {
        ScopedTimer t(ctx, s, DEHALO_K_MSM_REDUCE);
        // level 1: merge partials in chunks of MSM_L1
        k_diff<<<(tb + 255) / 256, 256, 0, s>>>(toff0, tb, cnt1);
        TRY(run_scan(ctx, cnt1, tb, MSM_L1, off1, toff1, s));
        u32 blocks1 = (u32)((nt1_max + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS);
        k_msm_merge<CV><<<blocks1, MSM_ACC_THREADS, 0, s>>>(tb, MSM_L1, toff0, toff1, partial0, partial1, 0);
        // level 2: whatever is left per bucket
        k_msm_merge<CV><<<(tb + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS, MSM_ACC_THREADS, 0, s>>>(tb, 0xffffffffu, toff1, nullptr, partial1, buckets, 1);
        // bucket reduction
        u32 nthreads = per_group * (u32)total_groups;
        k_msm_reduce_local<CV><<<(nthreads + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS, MSM_ACC_THREADS, 0, s>>>(g.nb, (u32)total_groups, buckets, contrib);
        const xyzz_t* cur = contrib;
        u32 cnt = per_group;
        xyzz_t* bufs[2] = {tree, contrib};  // ping-pong: contrib is free once consumed
        int which = 0;
        while (cnt > 1) {
            u32 out_cnt = (cnt + 2 * MSM_TREE_THREADS - 1) / (2 * MSM_TREE_THREADS);
            xyzz_t* o = out_cnt == 1 ? gsums : bufs[which];
            dim3 grid(out_cnt, (u32)total_groups);
            k_msm_tree_sum<CV><<<grid, MSM_TREE_THREADS, 0, s>>>(cur, cnt, o, out_cnt);
            cur = o; cnt = out_cnt; which ^= 1;
        }
        if (cur != gsums) HIP_TRY(ctx, hipMemcpyAsync(gsums, cur, total_groups * sizeof(xyzz_t), hipMemcpyDeviceToDevice, s));
        k_msm_final<CV><<<(u32)batch, 64, 0, s>>>(g, gsums, d_out);
        HIP_TRY(ctx, hipGetLastError());
    }
    return 0;
}

hipStream_t pick_stream(dehalo_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

int register_impl(dehalo_ctx* ctx, int curve, const uint64_t* affine_xy, size_t n, size_t stride_bytes, int window_bits, int precompute,
                  dehalo_bases** out) {
    if (!affine_xy || !out || n == 0 || stride_bytes < 64 || n >= (1ull << 30)) return fail(ctx, DEHALO_ERR_INVALID, "bases_register: bad argument");
    if (window_bits != 0 && (window_bits < 4 || window_bits > 16)) return fail(ctx, DEHALO_ERR_INVALID, "window_bits must be 0 or in [4, 16]");
    uint32_t c = window_bits ? (uint32_t)window_bits : choose_window(n);
    if (c < 4) c = 4;
    uint32_t W = (256 + c - 1) / c;
    if (precompute && (uint64_t)n * W >= (1ull << 31)) return fail(ctx, DEHALO_ERR_INVALID, "precomputed table too large");
    dehalo_bases* b = new dehalo_bases();
    b->curve = curve; b->n = n; b->c = c; b->W = W; b->precomp = precompute ? 1 : 0; b->table = nullptr;
    size_t rows = precompute ? W : 1;
    hipError_t e = hipMalloc((void**)&b->table, rows * n * sizeof(affine_t));
    if (e != hipSuccess) { delete b; return fail(ctx, DEHALO_ERR_OOM, std::string("bases table: ") + hipGetErrorString(e)); }
    e = hipMemcpy2DAsync(b->table, 64, affine_xy, stride_bytes, 64, n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && precompute) {
        int rc = dispatch_curve(ctx, curve, [&](auto cv) {
            typedef decltype(cv) CV;
            k_msm_precompute<CV><<<(u32)((n + 127) / 128), 128, 0, ctx->stream>>>(b->table, (u32)n, c, W);
            return 0;
        });
        if (rc) { (void)hipFree(b->table); delete b; return rc; }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(b->table); delete b; return fail(ctx, DEHALO_ERR_HIP, std::string("bases upload: ") + hipGetErrorString(e)); }
    *out = b;
    return 0;
}

}  // namespace

// ==========================================================================================
extern "C" {

const char* dehalo_version(void) { return "dehalo 0.1 gfx950"; }

int dehalo_ctx_create(int device, dehalo_ctx** out) {
    if (!out) return DEHALO_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return DEHALO_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DEHALO_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DEHALO_ERR_NO_DEVICE;  // code objects are gfx950-only
    if (hipSetDevice(device) != hipSuccess) return DEHALO_ERR_NO_DEVICE;
    dehalo_ctx* ctx = new dehalo_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return DEHALO_ERR_HIP; }
    *out = ctx;
    return 0;
}

void dehalo_ctx_destroy(dehalo_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->ws_scalars, &ctx->ws_out, &ctx->ws_count, &ctx->ws_cursor, &ctx->ws_off, &ctx->ws_toff0, &ctx->ws_cnt1, &ctx->ws_off1,
                      &ctx->ws_toff1, &ctx->ws_bsum, &ctx->ws_idx, &ctx->ws_partial0, &ctx->ws_partial1, &ctx->ws_buckets, &ctx->ws_contrib,
                      &ctx->ws_tree, &ctx->ws_gsums, &ctx->ws_ntt_scratch, &ctx->ws_ntt_io, &ctx->ws_ntt_io2, &ctx->ws_fop[0], &ctx->ws_fop[1],
                      &ctx->ws_fop[2], &ctx->ws_tmp_bases};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& t : ctx->twiddles) (void)hipFree(t.tw);
    for (auto& r : ctx->regions) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* dehalo_last_error(const dehalo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dehalo_ctx_synchronize(dehalo_ctx* ctx) {
    if (!ctx) return DEHALO_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_bases_register(dehalo_ctx* ctx, int curve, const uint64_t* affine_xy, size_t n, size_t stride_bytes, int window_bits, int precompute,
                          dehalo_bases** out) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (curve < 0 || curve > 2) return fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return register_impl(ctx, curve, affine_xy, n, stride_bytes, window_bits, precompute, out);
}

int dehalo_bases_release(dehalo_ctx* ctx, dehalo_bases* bases) {
    if (!ctx || !bases) return DEHALO_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(bases->table);
    delete bases;
    return 0;
}

size_t dehalo_bases_len(const dehalo_bases* bases) { return bases ? bases->n : 0; }

int dehalo_msm_device(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* d_scalars, size_t len, size_t batch, uint64_t* d_out_jacobian,
                      void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!bases || (!d_scalars && len) || !d_out_jacobian) return fail(ctx, DEHALO_ERR_INVALID, "msm: null argument");
    if (len > bases->n) return fail(ctx, DEHALO_ERR_INVALID, "msm: more scalars than registered bases");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    hipStream_t s = pick_stream(ctx, stream);
    return dispatch_curve(ctx, bases->curve, [&](auto cv) {
        typedef decltype(cv) CV;
        return run_msm<CV>(ctx, bases, (const fe*)d_scalars, len, batch, (jacobian_t*)d_out_jacobian, s);
    });
}

int dehalo_msm_batch(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* const* scalars, size_t len, size_t batch, uint64_t* out_jacobian) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!bases || !scalars || !out_jacobian) return fail(ctx, DEHALO_ERR_INVALID, "msm: null argument");
    if (len > bases->n) return fail(ctx, DEHALO_ERR_INVALID, "msm: more scalars than registered bases");
    for (size_t b = 0; b < batch; b++)
        if (!scalars[b] && len) return fail(ctx, DEHALO_ERR_INVALID, "msm: null scalar column");
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        TRY(ensure(ctx, ctx->ws_scalars, std::max<size_t>(32, batch * len * 32)));
        TRY(ensure(ctx, ctx->ws_out, std::max<size_t>(96, batch * 96)));
        for (size_t b = 0; b < batch && len; b++)
            HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->ws_scalars.p + b * len * 32, scalars[b], len * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    TRY(dehalo_msm_device(ctx, bases, (const uint64_t*)ctx->ws_scalars.p, len, batch, (uint64_t*)ctx->ws_out.p, nullptr));
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipMemcpyAsync(out_jacobian, ctx->ws_out.p, batch * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_msm(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* scalars, size_t len, uint64_t out_jacobian[12]) {
    const uint64_t* cols[1] = {scalars};
    return dehalo_msm_batch(ctx, bases, cols, len, 1, out_jacobian);
}

int dehalo_best_multiexp(dehalo_ctx* ctx, int curve, const uint64_t* scalars, const uint64_t* affine_xy, size_t len, uint64_t out_jacobian[12]) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!out_jacobian || ((!scalars || !affine_xy) && len)) return fail(ctx, DEHALO_ERR_INVALID, "best_multiexp: null argument");
    if (curve < 0 || curve > 2) return fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    if (len == 0) { memset(out_jacobian, 0, 96); return 0; }
    dehalo_bases* b = nullptr;
    TRY(dehalo_bases_register(ctx, curve, affine_xy, len, 64, 0, 0, &b));
    int rc = dehalo_msm(ctx, b, scalars, len, out_jacobian);
    dehalo_bases_release(ctx, b);
    return rc;
}

int dehalo_to_affine(dehalo_ctx* ctx, int curve, const uint64_t* jacobian, size_t count, uint64_t* affine_xy) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!jacobian || !affine_xy) && count) return fail(ctx, DEHALO_ERR_INVALID, "to_affine: null argument");
    if (count == 0) return 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    TRY(ensure(ctx, ctx->ws_fop[0], count * 96));
    TRY(ensure(ctx, ctx->ws_fop[1], count * 64));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_fop[0].p, jacobian, count * 96, hipMemcpyHostToDevice, ctx->stream));
    TRY(dispatch_curve(ctx, curve, [&](auto cv) {
        typedef decltype(cv) CV;
        k_jac_to_affine<CV><<<(u32)((count + 63) / 64), 64, 0, ctx->stream>>>((const jacobian_t*)ctx->ws_fop[0].p, (affine_t*)ctx->ws_fop[1].p, (u32)count);
        return 0;
    }));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(affine_xy, ctx->ws_fop[1].p, count * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- NTT family -----------------------------------------------------------------------------
static int ntt_device_impl(dehalo_ctx* ctx, int field, const uint64_t* d_src, uint64_t src_len, uint64_t src_stride, uint64_t* d_dst,
                           uint64_t dst_stride, uint32_t log_n, const uint64_t omega[4], size_t batch, const NttScale& sc, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!d_src || !d_dst || !omega) return fail(ctx, DEHALO_ERR_INVALID, "ntt: null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    hipStream_t s = pick_stream(ctx, stream);
    return dispatch_field(ctx, field, [&](auto f) {
        typedef decltype(f) F;
        return run_ntt<F>(ctx, (const fe*)d_src, src_len, src_stride, (fe*)d_dst, dst_stride, log_n, omega, batch, sc, s);
    });
}

int dehalo_ntt_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_n, const uint64_t omega[4], size_t batch, void* stream) {
    NttScale sc;
    uint64_t N = 1ull << (log_n & 63);
    return ntt_device_impl(ctx, field, d_a, N, N, d_a, N, log_n, omega, batch, sc, stream);
}

int dehalo_intt_scaled_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_n, const uint64_t omega_inv[4], const uint64_t n_inv[4],
                              size_t batch, void* stream) {
    if (!n_inv) return fail(ctx, DEHALO_ERR_INVALID, "intt: null n_inv");
    NttScale sc;
    sc.post_mode = 1; sc.post0 = fe_from_u64(n_inv);
    uint64_t N = 1ull << (log_n & 63);
    return ntt_device_impl(ctx, field, d_a, N, N, d_a, N, log_n, omega_inv, batch, sc, stream);
}

int dehalo_coset_ntt_device(dehalo_ctx* ctx, int field, const uint64_t* d_coeffs, uint32_t log_n, uint64_t* d_ext_out, uint32_t log_ext,
                            const uint64_t omega_ext[4], const uint64_t zeta[4], size_t batch, void* stream) {
    if (!zeta || log_ext < log_n) return fail(ctx, DEHALO_ERR_INVALID, "coset_ntt: bad argument");
    if (d_coeffs == d_ext_out) return fail(ctx, DEHALO_ERR_INVALID, "coset_ntt: coeffs and ext_out may not alias");
    NttScale sc;
    sc.pre_mode = 1; sc.pre_z = fe_from_u64(zeta);  // zeta^2 is formed inside the kernel
    uint64_t n = 1ull << (log_n & 63), N = 1ull << (log_ext & 63);
    return ntt_device_impl(ctx, field, d_coeffs, n, n, d_ext_out, N, log_ext, omega_ext, batch, sc, stream);
}

int dehalo_coset_intt_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_ext, const uint64_t omega_ext_inv[4],
                             const uint64_t ext_n_inv[4], const uint64_t zeta[4], size_t batch, void* stream) {
    if (!zeta || !ext_n_inv) return fail(ctx, DEHALO_ERR_INVALID, "coset_intt: null argument");
    NttScale sc;
    sc.post_mode = 2; sc.post0 = fe_from_u64(ext_n_inv); sc.post_z = fe_from_u64(zeta);
    uint64_t N = 1ull << (log_ext & 63);
    return ntt_device_impl(ctx, field, d_a, N, N, d_a, N, log_ext, omega_ext_inv, batch, sc, stream);
}

// host-buffer forms: upload, transform, download
static int with_host_io(dehalo_ctx* ctx, const uint64_t* in, size_t in_elems, uint64_t* out, size_t out_elems, bool inplace,
                        int (*fn)(dehalo_ctx*, uint64_t*, uint64_t*, void*), void* arg) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!in || !out) return fail(ctx, DEHALO_ERR_INVALID, "null buffer");
    uint64_t *d_in, *d_out;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        TRY(ensure(ctx, ctx->ws_ntt_io, std::max<size_t>(32, in_elems * 32)));
        d_in = (uint64_t*)ctx->ws_ntt_io.p;
        if (inplace) d_out = d_in;
        else {
            TRY(ensure(ctx, ctx->ws_ntt_io2, std::max<size_t>(32, out_elems * 32)));
            d_out = (uint64_t*)ctx->ws_ntt_io2.p;
        }
        HIP_TRY(ctx, hipMemcpyAsync(d_in, in, in_elems * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    TRY(fn(ctx, d_in, d_out, arg));
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipMemcpyAsync(out, d_out, out_elems * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

struct NttArgs { int field; uint32_t log_n, log_ext; const uint64_t *w, *s, *z; };

int dehalo_ntt(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_n, const uint64_t omega[4]) {
    if (log_n > 30) return fail(ctx, DEHALO_ERR_INVALID, "log_n > 30");
    NttArgs A{field, log_n, 0, omega, nullptr, nullptr};
    size_t N = (size_t)1 << log_n;
    return with_host_io(ctx, a, N, a, N, true, [](dehalo_ctx* c, uint64_t* di, uint64_t*, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_ntt_device(c, A->field, di, A->log_n, A->w, 1, nullptr);
    }, &A);
}

int dehalo_intt_scaled(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_n, const uint64_t omega_inv[4], const uint64_t n_inv[4]) {
    if (log_n > 30) return fail(ctx, DEHALO_ERR_INVALID, "log_n > 30");
    NttArgs A{field, log_n, 0, omega_inv, n_inv, nullptr};
    size_t N = (size_t)1 << log_n;
    return with_host_io(ctx, a, N, a, N, true, [](dehalo_ctx* c, uint64_t* di, uint64_t*, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_intt_scaled_device(c, A->field, di, A->log_n, A->w, A->s, 1, nullptr);
    }, &A);
}

int dehalo_coset_ntt(dehalo_ctx* ctx, int field, const uint64_t* coeffs, uint32_t log_n, uint64_t* ext_out, uint32_t log_ext,
                     const uint64_t omega_ext[4], const uint64_t zeta[4]) {
    if (log_ext > 30 || log_ext < log_n) return fail(ctx, DEHALO_ERR_INVALID, "coset_ntt: bad sizes");
    NttArgs A{field, log_n, log_ext, omega_ext, nullptr, zeta};
    return with_host_io(ctx, coeffs, (size_t)1 << log_n, ext_out, (size_t)1 << log_ext, false, [](dehalo_ctx* c, uint64_t* di, uint64_t* dout, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_coset_ntt_device(c, A->field, di, A->log_n, dout, A->log_ext, A->w, A->z, 1, nullptr);
    }, &A);
}

int dehalo_coset_intt(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_ext, const uint64_t omega_ext_inv[4], const uint64_t ext_n_inv[4],
                      const uint64_t zeta[4]) {
    if (log_ext > 30) return fail(ctx, DEHALO_ERR_INVALID, "log_ext > 30");
    NttArgs A{field, 0, log_ext, omega_ext_inv, ext_n_inv, zeta};
    size_t N = (size_t)1 << log_ext;
    return with_host_io(ctx, a, N, a, N, true, [](dehalo_ctx* c, uint64_t* di, uint64_t*, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_coset_intt_device(c, A->field, di, A->log_ext, A->w, A->s, A->z, 1, nullptr);
    }, &A);
}

int dehalo_field_op(dehalo_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!a || !out || op < 0 || op > 5) return fail(ctx, DEHALO_ERR_INVALID, "field_op: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    TRY(ensure(ctx, ctx->ws_fop[0], n * 32));
    TRY(ensure(ctx, ctx->ws_fop[1], n * 32));
    TRY(ensure(ctx, ctx->ws_fop[2], n * 32));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_fop[0].p, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (b) HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_fop[1].p, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
    TRY(dispatch_field(ctx, field, [&](auto f) {
        typedef decltype(f) F;
        k_field_op<F><<<(u32)((n + 127) / 128), 128, 0, ctx->stream>>>(op, (const fe*)ctx->ws_fop[0].p, b ? (const fe*)ctx->ws_fop[1].p : nullptr,
                                                                     (fe*)ctx->ws_fop[2].p, n);
        return 0;
    }));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->ws_fop[2].p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- measurement ------------------------------------------------------------------------------
int dehalo_timing_enable(dehalo_ctx* ctx, int on) {
    if (!ctx) return DEHALO_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->timing = on != 0;
    return 0;
}

static int timing_collect(dehalo_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    for (auto& r : ctx->regions) {
        HIP_TRY(ctx, hipEventSynchronize(r.b));
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, r.a, r.b));
        ctx->timing_ms[r.kernel_id] += ms;
        ctx->timing_cnt[r.kernel_id] += 1;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    ctx->regions.clear();
    return 0;
}

int dehalo_timing_reset(dehalo_ctx* ctx) {
    if (!ctx) return DEHALO_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(timing_collect(ctx));
    for (int i = 0; i < DEHALO_K_COUNT; i++) { ctx->timing_ms[i] = 0; ctx->timing_cnt[i] = 0; }
    return 0;
}

int dehalo_timing_get(dehalo_ctx* ctx, int kernel_id, double* total_ms, uint64_t* count) {
    if (!ctx || kernel_id < 0 || kernel_id >= DEHALO_K_COUNT || !total_ms || !count) return DEHALO_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(timing_collect(ctx));
    *total_ms = ctx->timing_ms[kernel_id];
    *count = ctx->timing_cnt[kernel_id];
    return 0;
}

}  // extern "C"
