// internal.hpp -- host-side state shared by the translation units of libdehalo.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dehalo.h"
#include "ec.cuh"

// Caller buffers of the host entry points are ordinary pageable memory (a Rust Vec<F>): large ones are pinned for the duration of the
// call so that the copy engine reads / writes them directly instead of going through the runtime's bounce buffers (measured on
// MI355X, profiles/r02_host_path_measurements.txt: dehalo_ntt at 2^20, 32 MiB each way, 5.73 -> 1.35 ms), and so that the lifetime of the device's mapping of
// caller memory is this object's and nothing else's (the pin ends only after the stream that copies has been synchronised).  Registration failing (already
// pinned, exotic mapping) just leaves the pageable path.
// Several threads may hand over the SAME host buffer at once (batch proving: one witness array, four provers): the first registers it, the others count
// themselves in, and the pages are released by whoever leaves last -- a copy that found the buffer page-locked by another thread's registration must not
// lose that registration in flight.  A range that only partly overlaps a registered one registers (or falls back to the pageable path) independently.
struct HostPinRegistry {
    struct Entry { uintptr_t a, b; int refs; };
    std::mutex mu;
    std::vector<Entry> entries;
};
inline HostPinRegistry& host_pin_registry() { static HostPinRegistry r; return r; }

struct HostPin {
    void* p = nullptr;          // the caller's pointer when its pages are page-locked through this object (by its own registration or one it shares)
    uintptr_t key = 0;          // start of the registration it holds a reference to
    HostPin(const void* ptr, size_t bytes) {
        if (!ptr || bytes < HOST_PIN_MIN_BYTES) return;
        HostPinRegistry& reg = host_pin_registry();
        std::lock_guard<std::mutex> lk(reg.mu);
        const uintptr_t a = (uintptr_t)ptr, b = a + bytes;
        for (auto& e : reg.entries)
            if (e.a <= a && b <= e.b) { e.refs++; key = e.a; p = const_cast<void*>(ptr); return; }
        if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) == hipSuccess) {
            reg.entries.push_back({a, b, 1});
            key = a; p = const_cast<void*>(ptr);
        } else (void)hipGetLastError();
    }
    HostPin(const HostPin&) = delete;
    HostPin& operator=(const HostPin&) = delete;
    ~HostPin() {
        if (!p) return;
        HostPinRegistry& reg = host_pin_registry();
        std::lock_guard<std::mutex> lk(reg.mu);
        for (size_t i = 0; i < reg.entries.size(); i++)
            if (reg.entries[i].a == key) {
                if (--reg.entries[i].refs == 0) {
                    (void)hipHostUnregister((void*)key);
                    reg.entries.erase(reg.entries.begin() + i);
                }
                return;
            }
    }
    static constexpr size_t HOST_PIN_MIN_BYTES = 4u << 20;
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// Two library-owned page-locked chunks per context: every byte that travels between CALLER memory and the device either goes through them (a host memcpy
// on one side, a DMA on the other) or moves by DMA from / to pages that are page-locked for the duration of the call (HostPin, or locked by the caller).
// The HIP runtime's own handling of large pageable transfers -- it pins the caller's pages in place and keeps the registration in a small cache keyed by
// (address, size) after the copy -- is never reached from this library: the device then never holds a mapping of caller memory whose lifetime the
// library does not control ("the memory fault").
struct HostStage {
    static constexpr size_t CHUNK = 4u << 20;
    static constexpr size_t DIRECT_MAX = 64u << 10;      // below this the runtime copies through its own staging buffer (a host memcpy): nothing is pinned
    std::mutex mu;
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
};

struct TwiddleEntry {
    int field;
    uint32_t log_n;
    int form;  // 0: standard Montgomery (R = 2^256)
    uint64_t omega[4];
    uint64_t len;   // powers held: N (full table) or N / 2
    fe* tw;
};

struct TimedRegion {
    int kernel_id;
    hipEvent_t a, b;
};

struct dehalo_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    std::string err;           // guarded by err_mu (written on error paths of any thread, with or without mu held)
    std::mutex err_mu;
    std::recursive_mutex mu;   // recursive: host-buffer entry points hold it across their device-form calls
    // workspace (grow-only)
    DevBuf ws_scalars, ws_out, ws_count, ws_counters, ws_off, ws_records, ws_merge_lists, ws_merge_parts, ws_bhist, ws_pcount, ws_pairs, ws_bsum, ws_idx, ws_partial0, ws_buckets,
        ws_contrib, ws_tree, ws_bred_cnt, ws_gsums, ws_ntt_scratch, ws_ntt_io, ws_ntt_io2, ws_fop[3], ws_tmp_bases, ws_poly[5], ws_poly_io[3], ws_evh[4], ws_lookup;
    std::vector<TwiddleEntry> twiddles;
    affine_t* msm_affine_out = nullptr;   // set for the duration of dehalo_msm_device_affine (under mu): k_msm_final also writes affine points
    int msm_acc_points = 48; // > 0: the accumulation's grid is 4, 6, 8, ... layers of one wave per SIMD, the fewest that leave a lane <= this many
                             // points (2^20 x 16: 6 layers of 43; four dense 2^17 columns: 4 of 34, where whole rounds of 3 waves gave one round of 46
                             // on three quarters of the wave slots -- k = 17 proof 12.6 -> 12.05 ms); 0: rounds of msm_acc_waves waves per SIMD
    int msm_acc_waves = 3;   // sizes the accumulation's points per lane (dehalo_ctx_set_tuning): 3 -> 43 points per lane at 2^20 x 16, ~1.5
                             // rounds of the 4 waves per SIMD that are resident; measured best (one round of 86 points at 3 resident waves: 1.30 ms)
    int host_wait_spin_us = 400; // > 0: a host wait for this context's stream polls hipStreamQuery for up to this many microseconds before it blocks (dh_stream_wait;
                               // dehalo_ctx_set_tuning "host_wait_spin_us" / DEHALO_HOST_SPIN_US): the runtime's blocking wait wakes the thread ~15 us after the stream
                               // drained -- five waits of a K = 11 proof: 1.73 -> 1.65 ms; waits longer than this (a k = 17 commitment) block as before, where it was measured to make no difference
    int msm_acc_min_layers = 4; // the fewest layers of one wave per SIMD the accumulation's grid has (msm_acc_points > 0): 2 .. 4 (DEHALO_MSM_ACC_MIN_LAYERS)
    int msm_sort_block = 1024; // threads per workgroup of the sort's two scalar-decoding kernels: 1024 (128 / 115 KiB of LDS, a CU to itself) or 512 (64 / 58 KiB: shares a CU with an NTT
                               // tile / the tails of another context -- measured 2.3 % SLOWER in the step and equal in the proofs, profiles/r06_sort_block_ab.txt) (dehalo_ctx_set_tuning)
    int msm_acc_block = 128; // threads per block of k_msm_accum0: 128, or 768 = one 12-wave block per CU (3 waves per SIMD; dehalo_ctx_set_tuning / DEHALO_MSM_ACC_BLOCK)
    int ntt_full_table_log = 0;    // transforms up to this size keep all N twiddles (32 B x N; one load per inter-pass twiddle), larger ones N / 2 and a
                                   // negation.  Measured equal at 23 x 2^19 with warm clocks (1.19 ms either way: the negation hides behind the load), so
                                   // the default keeps the smaller table
    HostStage stage;
    bool timing = false;
    std::vector<TimedRegion> regions;
    double timing_ms[DEHALO_K_COUNT] = {};
    uint64_t timing_cnt[DEHALO_K_COUNT] = {};
};

struct dehalo_bases {
    int curve;
    size_t n;
    uint32_t c, W;
    int precomp;
    affine_t* table;  // n * (precomp ? W : 1) affine points in HBM, internal (R' = 2^261) canonical form
};

inline int dh_fail(dehalo_ctx* ctx, int code, const std::string& msg) {
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->err_mu);
        ctx->err = msg;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            int code_ = (e_ == hipErrorOutOfMemory) ? DEHALO_ERR_OOM : DEHALO_ERR_HIP;        \
            return dh_fail(ctx, code_, std::string(#expr) + ": " + hipGetErrorString(e_));   \
        }                                                                                    \
    } while (0)

#define TRY(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != 0) return rc_; \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel on the context's device: set once per (device, kernel) and only ever raised --
// the call costs microseconds, and every MSM / NTT / graph launch used to make it
inline hipError_t dh_func_lds(dehalo_ctx* ctx, const void* fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> have;
    std::lock_guard<std::mutex> lk(mu);
    int& cur = have[std::make_pair(ctx->device, fn)];
    if (cur >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}

inline int dh_ensure(dehalo_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

inline bool dh_host_locked(const void* ptr, size_t bytes) {
    {
        HostPinRegistry& reg = host_pin_registry();
        std::lock_guard<std::mutex> lk(reg.mu);
        const uintptr_t a = (uintptr_t)ptr, b = a + bytes;
        for (auto& e : reg.entries)
            if (e.a <= a && b <= e.b) return true;
    }
    hipPointerAttribute_t at;      // page-locked by the caller (hipHostMalloc / hipHostRegister / torch's pin_memory)
    if (hipPointerGetAttributes(&at, ptr) == hipSuccess) return at.type == hipMemoryTypeHost;
    (void)hipGetLastError();       // an ordinary pageable pointer is reported as an error by some runtime versions
    return false;
}

inline int dh_stage_init(dehalo_ctx* ctx) {
    HostStage& st = ctx->stage;
    for (int i = 0; i < 2; i++) {
        if (!st.buf[i]) HIP_TRY(ctx, hipHostMalloc(&st.buf[i], HostStage::CHUNK, hipHostMallocDefault));
        if (!st.ev[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&st.ev[i], hipEventDisableTiming));
    }
    return 0;
}

// host -> device from CALLER memory, queued on `s`.  On return the source has been read in full unless it is page-locked (then the copy is an ordinary
// asynchronous DMA from it and the caller's pin / the caller itself keeps the pages until `s` has been synchronised, as before).
inline int dh_h2d(dehalo_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    if (!bytes) return 0;
    if (bytes <= HostStage::DIRECT_MAX || dh_host_locked(h_src, bytes)) {
        HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
        return 0;
    }
    HostStage& st = ctx->stage;
    std::lock_guard<std::mutex> lk(st.mu);
    TRY(dh_stage_init(ctx));
    int i = 0;
    for (size_t off = 0; off < bytes; off += HostStage::CHUNK, i ^= 1) {
        const size_t len = std::min(HostStage::CHUNK, bytes - off);
        if (st.busy[i]) { HIP_TRY(ctx, hipEventSynchronize(st.ev[i])); st.busy[i] = false; }
        memcpy(st.buf[i], (const char*)h_src + off, len);
        HIP_TRY(ctx, hipMemcpyAsync((char*)d_dst + off, st.buf[i], len, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipEventRecord(st.ev[i], s));
        st.busy[i] = true;
    }
    return 0;
}

// device -> host into CALLER memory, behind everything queued on `s`; synchronous: the data is in h_dst when this returns
// Host wait for a stream.  With host_wait_spin_us set the thread first polls (a transcript round trip of the prover: the few commitments of a phase come back and
// the next phase's launches wait for the challenge hashed from them -- five to seven such waits a proof), then blocks as hipStreamSynchronize always does.
inline hipError_t dh_stream_wait(dehalo_ctx* ctx, hipStream_t s) {
    if (ctx->host_wait_spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipErrorNotReady) return q;
#if defined(__x86_64__) || defined(__i386__)
            for (int i = 0; i < 64; i++) __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= ctx->host_wait_spin_us) break;
        }
    }
    return hipStreamSynchronize(s);
}

inline int dh_d2h(dehalo_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, hipStream_t s) {
    if (!bytes) { HIP_TRY(ctx, dh_stream_wait(ctx, s)); return 0; }
    if (bytes <= HostStage::DIRECT_MAX || dh_host_locked(h_dst, bytes)) {
        HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, dh_stream_wait(ctx, s));
        return 0;
    }
    HostStage& st = ctx->stage;
    std::lock_guard<std::mutex> lk(st.mu);
    TRY(dh_stage_init(ctx));
    for (int i = 0; i < 2; i++)
        if (st.busy[i]) { HIP_TRY(ctx, hipEventSynchronize(st.ev[i])); st.busy[i] = false; }
    int i = 0;
    size_t prev_off = 0, prev_len = 0;
    for (size_t off = 0; off < bytes; off += HostStage::CHUNK, i ^= 1) {
        const size_t len = std::min(HostStage::CHUNK, bytes - off);
        HIP_TRY(ctx, hipMemcpyAsync(st.buf[i], (const char*)d_src + off, len, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipEventRecord(st.ev[i], s));
        if (prev_len) {      // the previous chunk lands in the other buffer: hand it to the caller while this one is in flight
            HIP_TRY(ctx, hipEventSynchronize(st.ev[i ^ 1]));
            memcpy((char*)h_dst + prev_off, st.buf[i ^ 1], prev_len);
        }
        prev_off = off; prev_len = len;
    }
    HIP_TRY(ctx, hipEventSynchronize(st.ev[i ^ 1]));
    memcpy((char*)h_dst + prev_off, st.buf[i ^ 1], prev_len);
    return 0;
}

// the limit of a precomputed table registered with window_bits = 0: n x windows < 2^30 (capi.hip)
bool dh_precomputed_table_fits(int curve, size_t n);

// Experiment switches.  The default build reads NO tuning from the environment: DH_EXPERIMENT_ENV("DEHALO_...") is a null pointer there and the name is not even in
// the binary, so an environment variable cannot change which kernels a drop-in library runs (per-context tuning goes through dehalo_ctx_set_tuning, validated).
// A measurement build (`make EXPERIMENTS=1`: -DDEHALO_EXPERIMENTS, what tools/ab_*.sh build) turns them back into getenv and compiles the wall-clock phase stamps in.
// The default build reads three diagnostics, host side only: DEHALO_PROVER_TRACE, DEHALO_SYNTH_TRACE (timelines on stderr), DEHALO_SYNTH_THREADS (witness threads).
#ifdef DEHALO_EXPERIMENTS
#define DH_EXPERIMENT_ENV(name) getenv(name)
#else
#define DH_EXPERIMENT_ENV(name) ((const char*)nullptr)
#endif

// DEHALO_CO_LDS (bytes; experiments with co-resident contexts): every latency-bound kernel that is meant to run in the wave slot a
// 768-thread accumulation block leaves free asks for at least this much LDS per block in all (its own + unused padding), so that no two such blocks
// fit one compute unit (> 80 KB of the 160) and the accumulation's next block always finds its three waves per SIMD.  0 = off.
inline size_t dh_co_lds_pad(size_t own_static, size_t own_dynamic) {
    static const size_t want = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_CO_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
    if (want <= own_static + own_dynamic) return own_dynamic;
    return want - own_static;
}
inline int dh_co_lds_attr(dehalo_ctx* ctx, const void* fn, size_t dyn) {
    if (dyn > 48 * 1024) HIP_TRY(ctx, dh_func_lds(ctx, fn, (int)dyn));
    return 0;
}

struct ScopedTimer {
    dehalo_ctx* ctx;
    hipStream_t s;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(dehalo_ctx* c, hipStream_t st, int kid) : ctx(c), s(st), id(kid) {
        if (ctx->timing) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
                a = b = nullptr;
                return;
            }
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

inline fe fe_from_u64(const uint64_t v[4]) {
    fe r;
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (u32)v[i];
        r.v[2 * i + 1] = (u32)(v[i] >> 32);
    }
    return r;
}

inline uint32_t log2_ceil(size_t n) {
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

struct NttScale {
    uint32_t pre_mode = 0, post_mode = 0;
    fe pre_z{}, post0{}, post_z{};
    int form_shift = 0;   // +1: emit the internal form (x * 2^261, canonical, packed); -1: the input is in it
};

// per-curve / per-field entry points (one translation unit each: msm_*.hip, ntt_*.hip)
#define DECL_MSM(NAME)                                                                                                                   \
    int run_msm_##NAME(dehalo_ctx* ctx, const dehalo_bases* bases, const fe* d_scalars, size_t len, size_t batch, jacobian_t* d_out,     \
                       hipStream_t s);                                                                                                   \
    int build_table_##NAME(dehalo_ctx* ctx, dehalo_bases* b, const affine_t* d_std_points, hipStream_t s);                               \
    int to_affine_##NAME(dehalo_ctx* ctx, const jacobian_t* d_in, affine_t* d_out, uint32_t count, hipStream_t s);                      \
    int point_sum_##NAME(dehalo_ctx* ctx, const jacobian_t* d_in, uint32_t count, jacobian_t* d_out, hipStream_t s);
DECL_MSM(bn254)
DECL_MSM(pallas)
DECL_MSM(vesta)
#undef DECL_MSM
// ParamsKZG::setup's device half (setup.cuh, instantiated in msm_bn254.hip): g[i] = [s^i] G, g_lagrange[i] = [L_i(s)] G into device memory
int kzg_setup_bn254(dehalo_ctx* ctx, uint32_t k, const uint64_t s[4], const uint64_t omega[4], const uint64_t cfac[4], affine_t* d_g, affine_t* d_gl, hipStream_t st);

#define DECL_NTT(NAME)                                                                                                                       \
    int run_ntt_##NAME(dehalo_ctx* ctx, const fe* src, uint64_t src_len, uint64_t src_stride, fe* dst, uint64_t dst_stride, uint32_t log_n, \
                       const uint64_t omega[4], size_t batch, const NttScale& sc, hipStream_t s);                                            \
    int field_op_##NAME(dehalo_ctx* ctx, int op, const fe* a, const fe* b, fe* out, uint64_t n, hipStream_t s);
DECL_NTT(bn254_fr)
DECL_NTT(bn254_fq)
DECL_NTT(pasta_fp)
DECL_NTT(pasta_fq)
#undef DECL_NTT

// field-vector primitives (poly.cuh), instantiated in the same per-field translation units
#define DECL_POLY(NAME)                                                                                                                     \
    int eval_poly_##NAME(dehalo_ctx* ctx, const fe* c, uint64_t len, uint64_t stride, size_t batch, const uint64_t pt[4], fe* out, hipStream_t s); \
    int eval_poly_multi_##NAME(dehalo_ctx* ctx, const fe* const* polys, size_t count, uint64_t len, const uint64_t* pts, uint32_t npts, fe* out, hipStream_t s, const uint8_t* masks); \
    int batch_invert_##NAME(dehalo_ctx* ctx, fe* v, uint64_t len, hipStream_t s);                                                           \
    int prefix_product_##NAME(dehalo_ctx* ctx, const fe* in, uint64_t len, fe* out, hipStream_t s);                                         \
    int grand_product_##NAME(dehalo_ctx* ctx, const fe* num, const fe* den, uint64_t len, size_t batch, uint64_t stride, fe* z, hipStream_t s); \
    int lincomb_##NAME(dehalo_ctx* ctx, const fe* const* cols, const uint64_t* coefs, size_t count, uint64_t len, fe* out, const uint64_t* sub0, hipStream_t s); \
    int scale_##NAME(dehalo_ctx* ctx, fe* a, uint64_t len, const uint64_t* pattern, uint32_t period, const fe* d_factor, hipStream_t s);    \
    int kate_division_##NAME(dehalo_ctx* ctx, const fe* a, uint64_t len, const uint64_t pt[4], fe* q, hipStream_t s);                       \
    int kate_division_batch_##NAME(dehalo_ctx* ctx, const fe* const* a, uint64_t len, const uint64_t* pts, fe* const* q, size_t count, hipStream_t s);
DECL_POLY(bn254_fr)
DECL_POLY(bn254_fq)
DECL_POLY(pasta_fp)
DECL_POLY(pasta_fq)
#undef DECL_POLY

// lookup_permute.hip
int lookup_permute_impl(dehalo_ctx* ctx, int field, const fe* d_inputs, const fe* d_tables, uint64_t n, size_t batch, uint64_t stride, fe* d_out_inputs,
                        fe* d_out_tables, hipStream_t s);

// per lookup of a call (null array: none): its table as distinct rows -- representative row indices and multiplicities on the device, their number (0 or more
// than 2048: the general path); lookups that share a table carry the same arrays
struct LookupDistinct { const uint32_t* d_rep_rows; const uint32_t* d_mult; uint32_t count; };
int lookup_permute_ptrs(dehalo_ctx* ctx, int field, const fe* const* d_inputs, const fe* const* d_tables, uint64_t n, size_t batch, fe* const* d_out_inputs,
                        fe* const* d_out_tables, hipStream_t s, int* d_status = nullptr, const LookupDistinct* distinct = nullptr);

// quotient-numerator kernels (evalh.cuh)
struct dehalo_graph;
#define DECL_EVALH(NAME)                                                                                                                    \
    int convert_form_##NAME(dehalo_ctx* ctx, const fe* in, fe* out, uint64_t n, int to_internal, hipStream_t s);                           \
    int graph_upload_##NAME(dehalo_ctx* ctx, dehalo_graph* g, const uint64_t* constants, hipStream_t s);                                   \
    int graph_evaluate_##NAME(dehalo_ctx* ctx, const dehalo_graph* g, const dehalo_eval_inputs* in, uint32_t log_rows, uint32_t rot_scale, \
                              const fe* prev, fe* out, hipStream_t s);                                                                     \
    int graph_evaluate_batch_##NAME(dehalo_ctx* ctx, const dehalo_graph* const* graphs, uint32_t count, const dehalo_eval_inputs* in, uint32_t log_rows, \
                                    uint32_t rot_scale, fe* const* outs, hipStream_t s);                                                   \
    int perm_h_##NAME(dehalo_ctx* ctx, const dehalo_perm_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s);         \
    int lookup_h_##NAME(dehalo_ctx* ctx, const dehalo_lookup_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s);    \
    int lookup_h_batch_##NAME(dehalo_ctx* ctx, const dehalo_lookup_inputs* in, uint32_t count, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s); \
    int product_terms_##NAME(dehalo_ctx* ctx, const dehalo_product_inputs* in, uint64_t n, fe* num, fe* den, uint64_t stride, hipStream_t s);
DECL_EVALH(bn254_fr)
DECL_EVALH(bn254_fq)
DECL_EVALH(pasta_fp)
DECL_EVALH(pasta_fq)
#undef DECL_EVALH
