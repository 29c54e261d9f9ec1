// prover.hip -- the whole call behind the C ABI: ParamsKZG, keygen_vk / keygen_pk, ProvingKey / VerifyingKey RawBytes, Blake2bWrite and
// create_proof (KZG / GWC) [UPSTREAM halo2_proofs @ v2023_04_20: poly/kzg/commitment.rs, plonk/keygen.rs, plonk.rs, transcript.rs,
// plonk/prover.rs, plonk/{lookup,permutation,vanishing}/prover.rs, poly/kzg/multiopen/gwc/prover.rs] -- the calls the reference makes at
// benches/delay_enc.rs:41-54 (params), :84-115 (keys), :120-134 (create_proof into a Blake2bWrite transcript).
//
// Every C entry point is a function-try-block: an allocation failure on the host (std::bad_alloc) leaves as DEHALO_ERR_OOM, never as an exception.
// Host logic only: it orders the phases, hashes the transcript and does O(1) field arithmetic per challenge (hostfield.hpp); every column
// operation is one of this library's device entry points (include/dehalo.h), called directly.  No CPU path for column work exists.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <thread>
#include <unordered_map>

#include "blake2b.hpp"
#include "hostrng.hpp"
#include "internal.hpp"
#include "plonk_host.hpp"

namespace {

using clk = std::chrono::steady_clock;
inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

// ---- device memory owned by an object of this file ----
struct DevMem {
    fe* p = nullptr;
    size_t elems = 0;
    DevMem() = default;
    DevMem(const DevMem&) = delete;
    DevMem& operator=(const DevMem&) = delete;
    ~DevMem() { reset(); }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
        elems = 0;
    }
    int alloc(dehalo_ctx* ctx, size_t n_elems, bool zero = true) {
        reset();
        if (!n_elems) return 0;
        HIP_TRY(ctx, hipMalloc((void**)&p, n_elems * sizeof(fe)));
        elems = n_elems;
        if (zero) HIP_TRY(ctx, hipMemsetAsync(p, 0, n_elems * sizeof(fe), ctx->stream));
        return 0;
    }
    fe* at(size_t elem) const { return p + elem; }
    uint64_t* u64(size_t elem = 0) const { return (uint64_t*)(p + elem); }
};

// the blinding rows of a phase's columns: column c gets `rows` elements at dst + c * dst_pitch from src + c * rows (a 2-D device copy through the
// runtime took 10-45 us for these few hundred bytes on the proving thread's stream, right before the phase's commitment)
__global__ void k_place_rows(fe* __restrict__ dst, uint64_t dst_pitch, const fe* __restrict__ src, uint32_t rows, uint32_t ncols) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * ncols) dst[(uint64_t)(i / rows) * dst_pitch + i % rows] = src[i];
}

__global__ void k_gather_elems(const fe* __restrict__ src, const uint64_t* __restrict__ idx, fe* __restrict__ dst, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[idx[i]];
}

// DEHALO_RNG_OS, the vanishing argument's random polynomial (n scalars: 4 MiB at k = 17): generated ON THE DEVICE from the proof's ChaCha20 key (32 bytes of
// operating-system entropy) instead of being drawn on a host thread and uploaded.  Scalar i = the first of ChaCha20 blocks (counter low = i, counter high =
// attempt 0, 1, ...), nonce = the helper's stream, whose first 256 bits masked to the modulus' length are < p: uniform over the field.
struct ChaKey { u32 k[8]; };
__device__ __forceinline__ u32 cha_rotl(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
__global__ void k_chacha_scalars(ChaKey key, u64 stream, fe p, u32 top_mask, fe* out, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (u32 attempt = 0;; attempt++) {
        u32 st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
                      (u32)i, ((u32)(i >> 32) & 0xffffu) | (attempt << 16), (u32)stream, (u32)(stream >> 32)};
        u32 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = st[j];
#define CHA_QR(a, b, c, d)                                                                                         \
    x[a] += x[b]; x[d] = cha_rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = cha_rotl(x[b] ^ x[c], 12);                \
    x[a] += x[b]; x[d] = cha_rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = cha_rotl(x[b] ^ x[c], 7);
        for (int r = 0; r < 10; r++) {
            CHA_QR(0, 4, 8, 12) CHA_QR(1, 5, 9, 13) CHA_QR(2, 6, 10, 14) CHA_QR(3, 7, 11, 15)
            CHA_QR(0, 5, 10, 15) CHA_QR(1, 6, 11, 12) CHA_QR(2, 7, 8, 13) CHA_QR(3, 4, 9, 14)
        }
#undef CHA_QR
        fe v;
#pragma unroll
        for (int j = 0; j < 8; j++) v.v[j] = x[j] + st[j];
        v.v[7] &= top_mask;
        bool below = false, decided = false;
#pragma unroll
        for (int j = 7; j >= 0; j--)
            if (!decided && v.v[j] != p.v[j]) { below = v.v[j] < p.v[j]; decided = true; }
        if (below) {
            out[i] = v;
            return;
        }
    }
}

void put_u32_be(std::vector<uint8_t>& o, uint32_t v) { for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
void put_u32_be(uint8_t* o, uint32_t v) { for (int i = 0; i < 4; i++) o[i] = (uint8_t)(v >> (8 * (3 - i))); }
uint32_t get_u32_be(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

}   // namespace

// ================================================================================================ ParamsKZG
struct dehalo_params {
    dehalo_ctx* ctx = nullptr;
    int curve = 0;
    uint32_t k = 0;
    size_t n = 0;
    std::vector<uint64_t> g, g_lagrange;      // host copies (write())
    uint8_t g2[128] = {}, s_g2[128] = {};
    dehalo_bases *bases_g = nullptr, *bases_gl = nullptr;
};

extern "C" int dehalo_params_create(dehalo_ctx* ctx, int curve, uint32_t k, const uint64_t* g, const uint64_t* g_lagrange, const uint8_t* g2, const uint8_t* s_g2,
                                    dehalo_params** out) try {
    if (!ctx || !out || !g || !g_lagrange) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_create: null argument");
    if (k > 28 || curve_scalar_field(curve) < 0) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_create: k or curve out of range");
    std::unique_ptr<dehalo_params> p(new dehalo_params);
    p->ctx = ctx; p->curve = curve; p->k = k; p->n = (size_t)1 << k;
    p->g.assign(g, g + 8 * p->n);
    p->g_lagrange.assign(g_lagrange, g_lagrange + 8 * p->n);
    if (g2) memcpy(p->g2, g2, 128);
    if (s_g2) memcpy(p->s_g2, s_g2, 128);
    TRY(dehalo_bases_register(ctx, curve, g, p->n, 64, 0, 1, &p->bases_g));
    const int rc = dehalo_bases_register(ctx, curve, g_lagrange, p->n, 64, 0, 1, &p->bases_gl);
    if (rc) {
        (void)dehalo_bases_release(ctx, p->bases_g);
        return rc;
    }
    *out = p.release();
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

// ---- ParamsKZG::setup: the two G2 points on the host (O(1): one 254-bit scalar multiplication over Fq2 = Fq[i] / (i^2 + 1)) -------------------
namespace {
struct Fq2 { Fe a, b; };
struct G2Host {
    const HostField* q;
    explicit G2Host(const HostField* f) : q(f) {}
    Fq2 add(const Fq2& x, const Fq2& y) const { return {q->add(x.a, y.a), q->add(x.b, y.b)}; }
    Fq2 sub(const Fq2& x, const Fq2& y) const { return {q->sub(x.a, y.a), q->sub(x.b, y.b)}; }
    Fq2 mul(const Fq2& x, const Fq2& y) const { return {q->sub(q->mul(x.a, y.a), q->mul(x.b, y.b)), q->add(q->mul(x.a, y.b), q->mul(x.b, y.a))}; }
    Fq2 inv(const Fq2& x) const {
        const Fe d = q->invert(q->add(q->sqr(x.a), q->sqr(x.b)));
        return {q->mul(x.a, d), q->neg(q->mul(x.b, d))};
    }
    bool is_zero(const Fq2& x) const { return x.a.is_zero() && x.b.is_zero(); }
    struct Pt { Fq2 x, y; bool inf; };
    Pt padd(const Pt& P, const Pt& Q) const {      // affine chord-and-tangent (a = 0): 381 inversions for one scalar multiplication are nothing here
        if (P.inf) return Q;
        if (Q.inf) return P;
        Fq2 lam;
        if (is_zero(sub(P.x, Q.x))) {
            if (is_zero(add(P.y, Q.y))) return Pt{{}, {}, true};
            const Fq2 xx = mul(P.x, P.x);
            lam = mul(add(add(xx, xx), xx), inv(add(P.y, P.y)));
        } else lam = mul(sub(Q.y, P.y), inv(sub(Q.x, P.x)));
        Pt R;
        R.inf = false;
        R.x = sub(sub(mul(lam, lam), P.x), Q.x);
        R.y = sub(mul(lam, sub(P.x, R.x)), P.y);
        return R;
    }
    Pt scalar_mul(const uint64_t k_canonical[4], Pt P) const {
        Pt acc{{}, {}, true};
        for (int i = 0; i < 256; i++) {
            if ((k_canonical[i >> 6] >> (i & 63)) & 1) acc = padd(acc, P);
            P = padd(P, P);
        }
        return acc;
    }
    void to_raw(const Pt& P, uint8_t out[128]) const {      // G2Affine RawBytes: x.c0 | x.c1 | y.c0 | y.c1, Montgomery limbs; all zero = identity
        memset(out, 0, 128);
        if (P.inf) return;
        memcpy(out, P.x.a.v, 32); memcpy(out + 32, P.x.b.v, 32); memcpy(out + 64, P.y.a.v, 32); memcpy(out + 96, P.y.b.v, 32);
    }
};
// the generator of BN254's G2 (halo2curves bn256::G2Affine::generator(); canonical limbs)
const uint64_t BN254_G2_GEN[4][4] = {{0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull},
                                     {0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull},
                                     {0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull},
                                     {0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull}};
}   // namespace

extern "C" int dehalo_params_setup(dehalo_ctx* ctx, int curve, uint32_t k, const uint64_t s[4], dehalo_params** out) try {
    if (!ctx || !out || !s) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_setup: null argument");
    if (curve != DEHALO_CURVE_BN254_G1) return dh_fail(ctx, DEHALO_ERR_UNSUPPORTED, "params_setup: ParamsKZG needs a pairing: BN254 only");
    const HostField* f = host_field(curve_scalar_field(curve));
    if (k > f->two_adicity) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_setup: k out of range");
    if (k > 25) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_setup: k > 25: the SRS tables would be too large (include/dehalo.h)");      // (2 x 15 x 2^26 x 64 B = 128 GB at k = 26: fits the index and the card, never exercised)
    // checked BEFORE the 2^(k+1) fixed-base multiplications: the tables this call registers must fit (BN254: k <= 25, include/dehalo.h)
    if (!dh_precomputed_table_fits(curve, (size_t)1 << k)) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_setup: 2^k x windows >= 2^30: precomputed table too large");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    Fe sm;
    memcpy(sm.v, s, 32);
    const size_t n = (size_t)1 << k;
    // omega = ROOT_OF_UNITY^(2^(S - k));  (s^n - 1) / n
    Fe omega = f->root_of_unity;
    for (uint32_t i = k; i < f->two_adicity; i++) omega = f->sqr(omega);
    Fe sn = sm;
    for (uint32_t i = 0; i < k; i++) sn = f->sqr(sn);
    const Fe cfac = f->mul(f->sub(sn, f->one), f->invert(f->from_u64((uint64_t)n)));
    std::unique_ptr<dehalo_params> p(new dehalo_params);
    p->ctx = ctx; p->curve = curve; p->k = k; p->n = n;
    DevMem dg, dgl;
    TRY(dg.alloc(ctx, 2 * n, false));
    TRY(dgl.alloc(ctx, 2 * n, false));
    TRY(kzg_setup_bn254(ctx, k, sm.v, omega.v, cfac.v, (affine_t*)dg.p, (affine_t*)dgl.p, ctx->stream));
    TRY(dehalo_bases_register_device(ctx, curve, dg.u64(), n, 0, 1, &p->bases_g));
    int rc = dehalo_bases_register_device(ctx, curve, dgl.u64(), n, 0, 1, &p->bases_gl);
    if (rc == 0) {
        p->g.resize(8 * n); p->g_lagrange.resize(8 * n);
        rc = dehalo_download(ctx, dg.p, 64 * n, p->g.data());
        if (rc == 0) rc = dehalo_download(ctx, dgl.p, 64 * n, p->g_lagrange.data());
    }
    if (rc) {
        (void)dehalo_bases_release(ctx, p->bases_g);
        if (p->bases_gl) (void)dehalo_bases_release(ctx, p->bases_gl);
        return rc;
    }
    {   // g2 = the generator, s_g2 = [s] g2
        const HostField* q = host_field(curve_base_field(curve));
        G2Host g2(q);
        G2Host::Pt G;
        G.inf = false;
        Fe c[4];
        for (int i = 0; i < 4; i++) { memcpy(c[i].v, BN254_G2_GEN[i], 32); c[i] = q->from_canonical(c[i]); }
        G.x = {c[0], c[1]}; G.y = {c[2], c[3]};
        const Fe sc = f->to_canonical(sm);
        g2.to_raw(G, p->g2);
        g2.to_raw(g2.scalar_mul(sc.v, G), p->s_g2);
    }
    *out = p.release();
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_params_read(dehalo_ctx* ctx, int curve, const uint8_t* bytes, size_t len, dehalo_params** out) try {
    if (!ctx || !bytes || !out) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_read: null argument");
    if (len < 4) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_read: unexpected end of input");
    const uint32_t k = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 24);      // u32 LE
    if (k > 28) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_read: k out of range");
    const size_t n = (size_t)1 << k;
    if (len != 4 + 2 * 64 * n + 256) return dh_fail(ctx, DEHALO_ERR_INVALID, "params_read: length does not match k");
    // (the points are 8-byte aligned only if the caller's buffer is: copy through aligned vectors)
    std::vector<uint64_t> g(8 * n), gl(8 * n);
    memcpy(g.data(), bytes + 4, 64 * n);
    memcpy(gl.data(), bytes + 4 + 64 * n, 64 * n);
    return dehalo_params_create(ctx, curve, k, g.data(), gl.data(), bytes + 4 + 128 * n, bytes + 4 + 128 * n + 128, out);
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" size_t dehalo_params_size(const dehalo_params* p) { return p ? 4 + 2 * 64 * p->n + 256 : 0; }

extern "C" int dehalo_params_write(const dehalo_params* p, uint8_t* out, size_t cap) try {
    if (!p || !out) return DEHALO_ERR_INVALID;
    if (cap < dehalo_params_size(p)) return dh_fail(p->ctx, DEHALO_ERR_INVALID, "params_write: buffer too small");
    for (int i = 0; i < 4; i++) out[i] = (uint8_t)(p->k >> (8 * i));
    memcpy(out + 4, p->g.data(), 64 * p->n);
    memcpy(out + 4 + 64 * p->n, p->g_lagrange.data(), 64 * p->n);
    memcpy(out + 4 + 128 * p->n, p->g2, 128);
    memcpy(out + 4 + 128 * p->n + 128, p->s_g2, 128);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_params_release(dehalo_ctx* ctx, dehalo_params* p) try {
    if (!p) return 0;
    if (p->bases_g) (void)dehalo_bases_release(ctx ? ctx : p->ctx, p->bases_g);
    if (p->bases_gl) (void)dehalo_bases_release(ctx ? ctx : p->ctx, p->bases_gl);
    delete p;
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_params_commit_device(dehalo_ctx* ctx, const dehalo_params* p, const uint64_t* d_polys, size_t batch, int lagrange, uint64_t* d_out_affine,
                                           void* stream) try {
    if (!ctx || !p) return DEHALO_ERR_INVALID;
    return dehalo_msm_device_affine(ctx, lagrange ? p->bases_gl : p->bases_g, d_polys, p->n, batch, nullptr, d_out_affine, stream);
} catch (...) { return DEHALO_ERR_OOM; }

// ================================================================================================ transcript
struct dehalo_transcript {
    int curve = 0;
    const HostField *fq = nullptr, *fr = nullptr;      // base field (coordinates), scalar field (challenges)
    Blake2b state;
    std::vector<uint8_t> proof;

    void init(int c) {
        curve = c;
        fq = host_field(curve_base_field(c));
        fr = host_field(curve_scalar_field(c));
        state.init(64, "Halo2-Transcript");
        proof.clear();
    }
    Fe squeeze() {      // Challenge255: Blake2b-512 over everything absorbed + the prefix byte 0 (which stays absorbed), reduced mod r
        const uint8_t z = 0;
        state.update(&z, 1);
        uint8_t d[64];
        state.digest(d);
        return fr->from_u512(d);
    }
    void common_scalar(const Fe& s) {
        uint8_t b[33];
        b[0] = 2;
        fr->to_bytes(s, b + 1);
        state.update(b, 33);
    }
    void write_scalar(const Fe& s) {
        uint8_t b[33];
        b[0] = 2;
        fr->to_bytes(s, b + 1);
        state.update(b, 33);
        proof.insert(proof.end(), b + 1, b + 33);
    }
    // affine {x, y} Montgomery; false for the identity (upstream: "cannot write points at infinity to the transcript")
    bool write_point(const uint64_t xy[8], bool also_to_proof = true) {
        Fe x, y;
        memcpy(x.v, xy, 32);
        memcpy(y.v, xy + 4, 32);
        if (x.is_zero() && y.is_zero()) return false;
        uint8_t b[65];
        b[0] = 1;
        fq->to_bytes(x, b + 1);
        fq->to_bytes(y, b + 33);
        state.update(b, 65);
        if (also_to_proof) {      // GroupEncoding: x little-endian, bit 7 of the last byte = y is odd
            uint8_t c[32];
            memcpy(c, b + 1, 32);
            c[31] |= (uint8_t)((b[33] & 1) << 7);
            proof.insert(proof.end(), c, c + 32);
        }
        return true;
    }
};

extern "C" int dehalo_transcript_create(int curve, dehalo_transcript** out) try {
    if (!out || curve_scalar_field(curve) < 0) return DEHALO_ERR_INVALID;
    dehalo_transcript* t = new dehalo_transcript;
    t->init(curve);
    *out = t;
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" int dehalo_transcript_common_scalar(dehalo_transcript* t, const uint64_t s[4]) try {
    if (!t || !s) return DEHALO_ERR_INVALID;
    Fe v;
    memcpy(v.v, s, 32);
    t->common_scalar(v);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" int dehalo_transcript_write_scalar(dehalo_transcript* t, const uint64_t s[4]) try {
    if (!t || !s) return DEHALO_ERR_INVALID;
    Fe v;
    memcpy(v.v, s, 32);
    t->write_scalar(v);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" int dehalo_transcript_write_point(dehalo_transcript* t, const uint64_t xy[8]) try {
    if (!t || !xy) return DEHALO_ERR_INVALID;
    return t->write_point(xy) ? 0 : DEHALO_ERR_INVALID;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" int dehalo_transcript_squeeze_challenge(dehalo_transcript* t, uint64_t out[4]) try {
    if (!t || !out) return DEHALO_ERR_INVALID;
    const Fe c = t->squeeze();
    memcpy(out, c.v, 32);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" size_t dehalo_transcript_len(const dehalo_transcript* t) { return t ? t->proof.size() : 0; }
extern "C" int dehalo_transcript_finalize(const dehalo_transcript* t, uint8_t* out, size_t cap) try {
    if (!t || (!out && !t->proof.empty())) return DEHALO_ERR_INVALID;
    if (cap < t->proof.size()) return DEHALO_ERR_INVALID;
    if (!t->proof.empty()) memcpy(out, t->proof.data(), t->proof.size());
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" void dehalo_transcript_release(dehalo_transcript* t) { delete t; }

extern "C" int dehalo_field_info(int field, uint64_t out[24]) try {
    const HostField* f = host_field(field);
    if (!f || !out) return DEHALO_ERR_INVALID;
    memcpy(out, f->p, 32);
    memcpy(out + 4, f->one.v, 32);
    memcpy(out + 8, f->root_of_unity.v, 32);
    memcpy(out + 12, f->zeta.v, 32);
    memcpy(out + 16, f->delta.v, 32);
    memcpy(out + 20, f->gen.v, 32);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_rng_scalars(dehalo_rng* rng, int field, uint64_t skip, uint64_t* out, size_t count) try {
    const HostField* f = host_field(field);
    if (!f || (!out && count)) return DEHALO_ERR_INVALID;
    HostRng r;
    TRY(r.init(rng, f));
    r.skip(skip);
    return r.scalars(out, count);
} catch (...) { return DEHALO_ERR_OOM; }

// ================================================================================================ keys
struct dehalo_pk {
    dehalo_ctx* ctx = nullptr;
    int curve = 0;
    const HostField* f = nullptr;
    HostCS cs;
    HostDomain dom;
    uint32_t k = 0, num_selectors = 0;
    std::vector<uint64_t> fixed_commitments, perm_commitments;      // Montgomery affine, 8 u64 each
    std::vector<std::vector<uint8_t>> selectors;                    // packed 8 bools per byte, LSB first
    Fe transcript_repr{};
    // device: values / polys in upstream's standard form, extended-domain columns in the kernels' internal form
    DevMem l_ext, fixed_values, fixed_polys, fixed_cosets, perm_values, perm_polys, perm_cosets;
    dehalo_graph* custom_gates = nullptr;
    std::vector<dehalo_graph*> lookup_graphs;
    std::vector<std::pair<dehalo_graph*, dehalo_graph*>> compress_graphs;

    ~dehalo_pk() {
        if (custom_gates) (void)dehalo_graph_release(ctx, custom_gates);
        for (auto* g : lookup_graphs) (void)dehalo_graph_release(ctx, g);
        for (auto& g : compress_graphs) {
            (void)dehalo_graph_release(ctx, g.first);
            (void)dehalo_graph_release(ctx, g.second);
        }
    }
    size_t vk_size() const { return 8 + 64 * (size_t)cs.num_fixed + 64 * cs.perm_cols.size() + (size_t)num_selectors * ((dom.n + 7) / 8); }
    void vk_write(uint8_t* o) const {
        put_u32_be(o, k);
        put_u32_be(o + 4, cs.num_fixed);
        o += 8;
        memcpy(o, fixed_commitments.data(), 64 * (size_t)cs.num_fixed);
        o += 64 * (size_t)cs.num_fixed;
        memcpy(o, perm_commitments.data(), 64 * cs.perm_cols.size());
        o += 64 * cs.perm_cols.size();
        for (auto& s : selectors) {
            memcpy(o, s.data(), s.size());
            o += s.size();
        }
    }
    size_t size() const {
        const size_t n = dom.n, m = dom.m, nf = cs.num_fixed, npc = cs.perm_cols.size();
        auto poly = [](size_t ln) { return 4 + 32 * ln; };
        auto sl = [&](size_t cnt, size_t ln) { return 4 + cnt * poly(ln); };
        return vk_size() + 3 * poly(m) + 2 * sl(nf, n) + sl(nf, m) + 2 * sl(npc, n) + sl(npc, m);
    }
    void default_transcript_repr() {
        std::vector<uint8_t> body(vk_size());
        vk_write(body.data());
        cs.encode(body);
        Blake2b h;
        h.init(64, "Halo2-Verify-Key");
        const uint64_t len = body.size();
        h.update(&len, 8);
        h.update(body.data(), body.size());
        uint8_t d[64];
        h.digest(d);
        transcript_repr = f->from_u512(d);
    }
    int compile_graphs() {
        TRY(custom_gates_graph(cs, f).compile(ctx, &custom_gates));
        for (auto& lk : cs.lookups) {
            dehalo_graph *g = nullptr, *gi = nullptr, *gt = nullptr;
            TRY(lookup_table_value_graph(cs, lk, f).compile(ctx, &g));
            lookup_graphs.push_back(g);
            TRY(compress_graph(cs, lk.inputs, f).compile(ctx, &gi));
            const int rc = compress_graph(cs, lk.tables, f).compile(ctx, &gt);
            compress_graphs.push_back({gi, gt});
            if (rc) return rc;
        }
        return 0;
    }
};

namespace {

int pk_common_init(dehalo_ctx* ctx, int curve, const dehalo_constraint_system* csd, uint32_t k, dehalo_pk* pk) {
    pk->ctx = ctx;
    pk->curve = curve;
    pk->k = k;
    pk->f = host_field(curve_scalar_field(curve));
    if (!pk->f) return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve");
    const std::string err = pk->cs.load(csd);
    if (!err.empty()) return dh_fail(ctx, DEHALO_ERR_INVALID, err);
    if (k > 28 || !pk->dom.init(pk->f, pk->cs.degree(), k)) return dh_fail(ctx, DEHALO_ERR_UNSUPPORTED, "extended_k exceeds the field's two-adicity");
    if (pk->dom.n < (size_t)pk->cs.blinding_factors() + 3) return dh_fail(ctx, DEHALO_ERR_INVALID, "not enough rows available");      // Error::NotEnoughRowsAvailable
    return 0;
}

// values (cnt x n, standard form, device) -> polys (lagrange_to_coeff), cosets (coeff_to_extended, internal form), commitments to the host
int lagrange_to_all(dehalo_pk* pk, const dehalo_params* params, const fe* values, size_t cnt, fe* polys, fe* cosets, uint64_t* commitments_host) {
    dehalo_ctx* ctx = pk->ctx;
    if (!cnt) return 0;
    const HostDomain& d = pk->dom;
    DevMem aff;
    if (commitments_host) {
        TRY(aff.alloc(ctx, 2 * cnt, false));
        TRY(dehalo_msm_device_affine(ctx, params->bases_gl, (const uint64_t*)values, d.n, cnt, nullptr, aff.u64(), nullptr));
    }
    TRY(dehalo_lagrange_to_coeff_device(ctx, pk->f->id, (const uint64_t*)values, (uint64_t*)polys, d.k, d.omega_inv.v, d.ifft_divisor.v, cnt, nullptr));
    TRY(dehalo_coset_ntt_form_device(ctx, pk->f->id, (const uint64_t*)polys, d.k, (uint64_t*)cosets, d.extended_k, d.ext_omega.v, d.g_coset.v, cnt, DEHALO_FORM_OUT_INTERNAL,
                                     nullptr));
    if (commitments_host) TRY(dehalo_download(ctx, aff.p, cnt * 64, commitments_host));
    else TRY(dehalo_ctx_synchronize(ctx));
    return 0;
}

// (n, 4) device column of omega^i: the forward NTT of the unit vector e_1
int omega_powers(dehalo_ctx* ctx, const HostDomain& d, fe* col) {
    HIP_TRY(ctx, hipMemsetAsync(col, 0, d.n * sizeof(fe), ctx->stream));
    TRY(dh_h2d(ctx, col + (d.n > 1 ? 1 : 0), d.f->one.v, 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // (the source is a host object: copied before returning)
    if (d.n > 1) TRY(dehalo_ntt_device(ctx, d.f->id, (uint64_t*)col, d.k, d.omega.v, 1, nullptr));
    return 0;
}

}   // namespace

extern "C" int dehalo_keygen(dehalo_ctx* ctx, const dehalo_params* params, const dehalo_constraint_system* csd, const uint64_t* fixed, const uint64_t* mapping,
                             const uint8_t* const* selectors, uint32_t num_selectors, uint32_t flags, dehalo_pk** out) try {
    if (!ctx || !params || !out) return dh_fail(ctx, DEHALO_ERR_INVALID, "keygen: null argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    std::unique_ptr<dehalo_pk> pk(new dehalo_pk);
    TRY(pk_common_init(ctx, params->curve, csd, params->k, pk.get()));
    const HostCS& cs = pk->cs;
    const HostDomain& d = pk->dom;
    const size_t n = d.n, m = d.m, nf = cs.num_fixed, npc = cs.perm_cols.size();
    if ((nf && !fixed) || (npc && !mapping) || (num_selectors && !selectors)) return dh_fail(ctx, DEHALO_ERR_INVALID, "keygen: null column data");
    const int fid = pk->f->id;
    // fixed columns
    TRY(pk->fixed_values.alloc(ctx, nf * n, false));
    TRY(pk->fixed_polys.alloc(ctx, nf * n, false));
    TRY(pk->fixed_cosets.alloc(ctx, nf * m, false));
    pk->fixed_commitments.assign(8 * nf, 0);
    if (nf) {
        HostPin pin_fixed(fixed, nf * n * 32);
        TRY(dh_h2d(ctx, pk->fixed_values.p, fixed, nf * n * 32, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (flags & DEHALO_KEYGEN_FIXED_CANONICAL) TRY(dehalo_field_op_device(ctx, fid, 4, pk->fixed_values.u64(), nullptr, pk->fixed_values.u64(), nf * n, nullptr));
        TRY(lagrange_to_all(pk.get(), params, pk->fixed_values.p, nf, pk->fixed_polys.p, pk->fixed_cosets.p, pk->fixed_commitments.data()));
    }
    // permutation: sigma_j(omega^i) = delta^(column of the mapped cell) * omega^(its row)  [permutation::keygen::Assembly::build_pk]
    TRY(pk->perm_values.alloc(ctx, npc * n, false));
    TRY(pk->perm_polys.alloc(ctx, npc * n, false));
    TRY(pk->perm_cosets.alloc(ctx, npc * m, false));
    pk->perm_commitments.assign(8 * npc, 0);
    if (npc) {
        for (size_t i = 0; i < npc * n; i++)
            if (mapping[i] >= npc * n) return dh_fail(ctx, DEHALO_ERR_INVALID, "keygen: permutation mapping points outside the permutation's columns");
        DevMem ident, w;
        uint64_t* d_map = nullptr;
        TRY(ident.alloc(ctx, npc * n, false));
        TRY(w.alloc(ctx, n, false));
        TRY(omega_powers(ctx, d, w.p));
        Fe dj = pk->f->one;
        for (size_t j = 0; j < npc; j++) {
            HIP_TRY(ctx, hipMemcpyAsync(ident.at(j * n), w.p, n * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
            if (j) TRY(dehalo_scale_device(ctx, fid, ident.u64(j * n), n, dj.v, 1, nullptr, nullptr));
            dj = pk->f->mul(dj, pk->f->delta);
        }
        HIP_TRY(ctx, hipMalloc((void**)&d_map, npc * n * 8));
        HostPin pin_map(mapping, npc * n * 8);
        hipError_t e = dh_h2d(ctx, d_map, mapping, npc * n * 8, ctx->stream) == 0 ? hipSuccess : hipErrorUnknown;
        if (e == hipSuccess) {
            k_gather_elems<<<(unsigned)((npc * n + 255) / 256), 256, 0, ctx->stream>>>(ident.p, d_map, pk->perm_values.p, npc * n);
            e = hipStreamSynchronize(ctx->stream);
        }
        (void)hipFree(d_map);
        HIP_TRY(ctx, e);
        TRY(lagrange_to_all(pk.get(), params, pk->perm_values.p, npc, pk->perm_polys.p, pk->perm_cosets.p, pk->perm_commitments.data()));
    }
    // l0, l_last, l_active_row = 1 - (l_last + l_blind) over the extended domain
    {
        const size_t u = n - (cs.blinding_factors() + 1);
        std::vector<Fe> lag(3 * n, Fe{{0, 0, 0, 0}});
        lag[0] = pk->f->one;
        lag[n + u] = pk->f->one;
        for (size_t i = 0; i < u; i++) lag[2 * n + i] = pk->f->one;
        DevMem vals, polys;
        TRY(vals.alloc(ctx, 3 * n, false));
        TRY(polys.alloc(ctx, 3 * n, false));
        TRY(pk->l_ext.alloc(ctx, 3 * m, false));
        HostPin pin_lag(lag.data(), 3 * n * 32);
        TRY(dh_h2d(ctx, vals.p, lag.data(), 3 * n * 32, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        TRY(lagrange_to_all(pk.get(), params, vals.p, 3, polys.p, pk->l_ext.p, nullptr));
    }
    pk->num_selectors = num_selectors;
    for (uint32_t s = 0; s < num_selectors; s++) {
        if (!selectors[s]) return dh_fail(ctx, DEHALO_ERR_INVALID, "keygen: null selector");
        std::vector<uint8_t> packed((n + 7) / 8, 0);
        for (size_t i = 0; i < n; i++)
            if (selectors[s][i]) packed[i >> 3] |= (uint8_t)(1u << (i & 7));
        pk->selectors.push_back(std::move(packed));
    }
    pk->default_transcript_repr();
    TRY(pk->compile_graphs());
    TRY(dehalo_ctx_synchronize(ctx));
    *out = pk.release();
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" size_t dehalo_pk_size(const dehalo_pk* pk) { return pk ? pk->size() : 0; }
extern "C" size_t dehalo_vk_size(const dehalo_pk* pk) { return pk ? pk->vk_size() : 0; }
extern "C" int dehalo_vk_write(const dehalo_pk* pk, uint8_t* out, size_t cap) try {
    if (!pk || !out) return DEHALO_ERR_INVALID;
    if (cap < pk->vk_size()) return dh_fail(pk->ctx, DEHALO_ERR_INVALID, "vk_write: buffer too small");
    pk->vk_write(out);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_pk_write(dehalo_ctx* ctx, const dehalo_pk* pk, uint8_t* out, size_t cap) try {
    if (!pk || !out) return DEHALO_ERR_INVALID;
    if (!ctx) ctx = pk->ctx;
    if (cap < pk->size()) return dh_fail(ctx, DEHALO_ERR_INVALID, "pk_write: buffer too small");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    const size_t n = pk->dom.n, m = pk->dom.m, nf = pk->cs.num_fixed, npc = pk->cs.perm_cols.size();
    pk->vk_write(out);
    uint8_t* o = out + pk->vk_size();
    DevMem tmp;      // extended-domain columns leave in upstream's standard form
    TRY(tmp.alloc(ctx, m, false));
    auto poly = [&](const fe* src, size_t len, bool internal) -> int {
        put_u32_be(o, (uint32_t)len);
        o += 4;
        if (internal) {
            TRY(dehalo_convert_form_device(ctx, pk->f->id, (const uint64_t*)src, tmp.u64(), len, 0, nullptr));
            src = tmp.p;
        }
        TRY(dehalo_download(ctx, src, len * 32, o));
        o += len * 32;
        return 0;
    };
    auto slice = [&](const DevMem& mem, size_t cnt, size_t len, bool internal) -> int {
        put_u32_be(o, (uint32_t)cnt);
        o += 4;
        for (size_t i = 0; i < cnt; i++) TRY(poly(mem.at(i * len), len, internal));
        return 0;
    };
    for (int i = 0; i < 3; i++) TRY(poly(pk->l_ext.at((size_t)i * m), m, true));
    TRY(slice(pk->fixed_values, nf, n, false));
    TRY(slice(pk->fixed_polys, nf, n, false));
    TRY(slice(pk->fixed_cosets, nf, m, true));
    TRY(slice(pk->perm_values, npc, n, false));
    TRY(slice(pk->perm_polys, npc, n, false));
    TRY(slice(pk->perm_cosets, npc, m, true));
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_pk_read(dehalo_ctx* ctx, int curve, const dehalo_constraint_system* csd, const uint8_t* bytes, size_t len, uint32_t num_selectors, dehalo_pk** out) try {
    if (!ctx || !bytes || !out) return dh_fail(ctx, DEHALO_ERR_INVALID, "pk_read: null argument");
    if (len < 8) return dh_fail(ctx, DEHALO_ERR_INVALID, "pk_read: unexpected end of input");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    const uint32_t k = get_u32_be(bytes), nf_file = get_u32_be(bytes + 4);
    std::unique_ptr<dehalo_pk> pk(new dehalo_pk);
    TRY(pk_common_init(ctx, curve, csd, k, pk.get()));
    const size_t n = pk->dom.n, m = pk->dom.m, nf = pk->cs.num_fixed, npc = pk->cs.perm_cols.size();
    if (nf_file != nf) return dh_fail(ctx, DEHALO_ERR_INVALID, "pk_read: the key's number of fixed commitments differs from the circuit's fixed columns");
    pk->num_selectors = num_selectors;
    if (len != pk->size()) return dh_fail(ctx, DEHALO_ERR_INVALID, "pk_read: length does not match the circuit (unexpected end of input or trailing bytes)");
    HostPin pin_blob(bytes, len);      // every polynomial below is copied straight out of the caller's blob
    const uint8_t* p = bytes + 8;
    pk->fixed_commitments.resize(8 * nf);
    memcpy(pk->fixed_commitments.data(), p, 64 * nf);
    p += 64 * nf;
    pk->perm_commitments.resize(8 * npc);
    memcpy(pk->perm_commitments.data(), p, 64 * npc);
    p += 64 * npc;
    for (uint32_t s = 0; s < num_selectors; s++) {
        pk->selectors.emplace_back(p, p + (n + 7) / 8);
        p += (n + 7) / 8;
    }
    const int fid = pk->f->id;
    auto poly = [&](fe* dst, size_t want, bool to_internal) -> int {
        if (get_u32_be(p) != want) return dh_fail(ctx, DEHALO_ERR_INVALID, "pk_read: polynomial length differs from the domain's");
        p += 4;
        TRY(dh_h2d(ctx, dst, p, want * 32, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        p += want * 32;
        if (to_internal) TRY(dehalo_convert_form_device(ctx, fid, (const uint64_t*)dst, (uint64_t*)dst, want, 1, nullptr));
        return 0;
    };
    auto slice = [&](DevMem& mem, size_t cnt, size_t ln, bool to_internal) -> int {
        if (get_u32_be(p) != cnt) return dh_fail(ctx, DEHALO_ERR_INVALID, "pk_read: polynomial count differs from the circuit's");
        p += 4;
        TRY(mem.alloc(ctx, cnt * ln, false));
        for (size_t i = 0; i < cnt; i++) TRY(poly(mem.at(i * ln), ln, to_internal));
        return 0;
    };
    TRY(pk->l_ext.alloc(ctx, 3 * m, false));
    for (int i = 0; i < 3; i++) TRY(poly(pk->l_ext.at((size_t)i * m), m, true));
    TRY(slice(pk->fixed_values, nf, n, false));
    TRY(slice(pk->fixed_polys, nf, n, false));
    TRY(slice(pk->fixed_cosets, nf, m, true));
    TRY(slice(pk->perm_values, npc, n, false));
    TRY(slice(pk->perm_polys, npc, n, false));
    TRY(slice(pk->perm_cosets, npc, m, true));
    pk->default_transcript_repr();
    TRY(pk->compile_graphs());
    TRY(dehalo_ctx_synchronize(ctx));
    *out = pk.release();
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_pk_set_transcript_repr(dehalo_pk* pk, const uint64_t repr[4]) try {
    if (!pk || !repr) return DEHALO_ERR_INVALID;
    memcpy(pk->transcript_repr.v, repr, 32);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" int dehalo_pk_get_transcript_repr(const dehalo_pk* pk, uint64_t repr[4]) try {
    if (!pk || !repr) return DEHALO_ERR_INVALID;
    memcpy(repr, pk->transcript_repr.v, 32);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }
extern "C" int dehalo_pk_release(dehalo_ctx* ctx, dehalo_pk* pk) try {
    if (!pk) return 0;
    dehalo_ctx* c = ctx ? ctx : pk->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    delete pk;
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

// ================================================================================================ create_proof
struct dehalo_prover {
    dehalo_ctx *ctx = nullptr, *side = nullptr;
    const dehalo_params* params = nullptr;
    const dehalo_pk* pk = nullptr;
    const HostField* f = nullptr;
    size_t n = 0, m = 0, u = 0;
    uint32_t k = 0, ek = 0, bf = 0, A = 0, L = 0, S = 0, I = 0, NC = 0, pieces = 0;
    uint32_t o_adv = 0, o_perm = 0, o_pz = 0, o_lz = 0, o_rand = 0;
    DevMem cols, polys_own, instance, instance_values, compressed, num, den, ext, h, table_value, hfold, qbuf, wbuf, jac, jac_side, evals, blind_dev, omega_col;
    fe* polys = nullptr;      // coefficient forms: polys_own with a side context, cols (in place) without
    std::vector<std::pair<dehalo_graph*, dehalo_graph*>> perm_graphs;      // per set: (denominator, numerator)
    std::vector<uint32_t> table_rep;      // per lookup: the first lookup with the same table expressions (shares its compressed table)
    // tables of fixed columns as distinct rows (lookup_permute.hip): per representative lookup one row index per distinct tuple of its table expressions' values over
    // the usable rows and the tuple's multiplicity, on the device; count 0: not such a table, or more distinct rows than the permutation's one-tile path takes
    struct TableRows { uint32_t* d_rep = nullptr; uint32_t* d_mult = nullptr; uint32_t count = 0; };
    std::vector<TableRows> table_rows;
    dehalo_graph *lookup_den = nullptr, *lookup_num = nullptr;
    hipEvent_t ev_ready[3] = {nullptr, nullptr, nullptr}, ev_inst = nullptr, ev_side = nullptr;      // ev_ready: one per commitment phase
    hipEvent_t ev_helper = nullptr;      // the helper thread waits for ITS work on the side stream through this event: a hipStreamSynchronize there holds the
    uint64_t* pin_helper = nullptr;      // stream against the proving thread's launches (0.4 ms of the lookups' phase); its point lands in this page-locked slot
    hipStream_t hs = nullptr;            // ... and its work (upload, the random polynomial's commitment) runs on a stream of its own beside the side context's
    uint64_t* adv_pin = nullptr;         // dehalo_create_proof_circuit: the advice columns the witness generator writes (page-locked, kept across proofs)
    uint64_t* rand_pin = nullptr;        // host-drawn random polynomial (page-locked: its upload is one DMA that holds no stream)
    // opening plan (depends on the circuit only)
    std::vector<int32_t> rots;
    std::vector<const uint64_t*> plist;
    std::vector<int64_t> write_idx;
    std::vector<uint8_t> eval_wanted;      // per polynomial of plist: the rotations (bits, in `rots` order) anyone reads its value at
    struct Group { int32_t rot; std::vector<const uint64_t*> ptrs; std::vector<int64_t> idx; };
    std::vector<Group> groups;
    std::vector<const uint64_t*> hp_ptrs;
    size_t hpiece0 = 0, eval_count = 0;
    // host staging
    std::vector<uint64_t> blind_host, host_aff, host_jac, host_evals;
    double timings[8] = {};
    // single-proof sharding (dehalo_prover_set_shard): this process computes columns [count * rank / world, count * (rank + 1) / world) of every multi-column
    // commitment and exchanges the points through the caller's all-gather
    uint32_t shard_rank = 0, shard_world = 1;
    dehalo_gather_fn shard_gather = nullptr;
    void* shard_user = nullptr;
    bool trace = false;      // DEHALO_PROVER_TRACE=1: host timestamps inside the phases go to stderr after each proof
    std::vector<std::pair<const char*, double>> ticks;
    clk::time_point t0;
    void tk(const char* label) {
        if (trace) ticks.push_back({label, ms_since(t0)});
    }
    std::mutex mu;      // one create_proof at a time per prover

    ~dehalo_prover() {
        for (auto& g : perm_graphs) {
            if (g.first) (void)dehalo_graph_release(ctx, g.first);
            if (g.second) (void)dehalo_graph_release(ctx, g.second);
        }
        if (lookup_den) (void)dehalo_graph_release(ctx, lookup_den);
        if (lookup_num) (void)dehalo_graph_release(ctx, lookup_num);
        for (hipEvent_t e : {ev_ready[0], ev_ready[1], ev_ready[2], ev_inst, ev_side, ev_helper})
            if (e) (void)hipEventDestroy(e);
        if (pin_helper) (void)hipHostFree(pin_helper);
        if (rand_pin) (void)hipHostFree(rand_pin);
        if (adv_pin) (void)hipHostFree(adv_pin);
        if (hs) (void)hipStreamDestroy(hs);
        for (auto& t : table_rows) {
            if (t.d_rep) (void)hipFree(t.d_rep);
            if (t.d_mult) (void)hipFree(t.d_mult);
        }
    }

    const uint64_t* col_ptr(const DevMem& mem, size_t col, size_t len) const { return (const uint64_t*)mem.at(col * len); }

    int build_product_graphs() {
        const HostCS& cs = pk->cs;
        const uint32_t chunk = cs.chunk_len(), nf = cs.num_fixed, npc = (uint32_t)cs.perm_cols.size();
        auto kind = [](uint32_t k) { return k == DEHALO_COLUMN_ADVICE ? DEHALO_SRC_ADVICE : k == DEHALO_COLUMN_FIXED ? DEHALO_SRC_FIXED : DEHALO_SRC_INSTANCE; };
        // fixed slots: [circuit fixed..., sigma_0.., omega column]; challenges: delta^j * beta per permutation column
        for (uint32_t s = 0; s < S; s++) {
            GraphBuilder gd(f), gn(f);
            GSrc dacc{}, nacc{};
            bool first = true;
            for (uint32_t j = s * chunk; j < std::min((s + 1) * chunk, npc); j++) {
                const auto& pc = cs.perm_cols[j];
                GSrc col = gd.column(kind(pc.kind), pc.index);
                GSrc t = gd.add_calc(DEHALO_CALC_MUL, GSrc{DEHALO_SRC_BETA, 0, 0}, gd.column(DEHALO_SRC_FIXED, nf + j));
                t = gd.add_calc(DEHALO_CALC_ADD, gd.add_calc(DEHALO_CALC_ADD, col, t), GSrc{DEHALO_SRC_GAMMA, 0, 0});
                dacc = first ? t : gd.add_calc(DEHALO_CALC_MUL, dacc, t);
                col = gn.column(kind(pc.kind), pc.index);
                t = gn.add_calc(DEHALO_CALC_MUL, GSrc{DEHALO_SRC_CHALLENGE, j, 0}, gn.column(DEHALO_SRC_FIXED, nf + npc));
                t = gn.add_calc(DEHALO_CALC_ADD, gn.add_calc(DEHALO_CALC_ADD, col, t), GSrc{DEHALO_SRC_GAMMA, 0, 0});
                nacc = first ? t : gn.add_calc(DEHALO_CALC_MUL, nacc, t);
                first = false;
            }
            gd.add_calc(DEHALO_CALC_STORE, dacc);
            gn.add_calc(DEHALO_CALC_STORE, nacc);
            dehalo_graph *d = nullptr, *nn = nullptr;
            TRY(gd.compile(ctx, &d));
            const int rc = gn.compile(ctx, &nn);
            perm_graphs.push_back({d, nn});
            if (rc) return rc;
        }
        // advice slots: [compressed_input, compressed_table, permuted_input, permuted_table]
        GraphBuilder gd(f), gn(f);
        gd.add_calc(DEHALO_CALC_MUL, gd.add_calc(DEHALO_CALC_ADD, gd.column(DEHALO_SRC_ADVICE, 2), GSrc{DEHALO_SRC_BETA, 0, 0}),
                    gd.add_calc(DEHALO_CALC_ADD, gd.column(DEHALO_SRC_ADVICE, 3), GSrc{DEHALO_SRC_GAMMA, 0, 0}));
        gn.add_calc(DEHALO_CALC_MUL, gn.add_calc(DEHALO_CALC_ADD, gn.column(DEHALO_SRC_ADVICE, 0), GSrc{DEHALO_SRC_BETA, 0, 0}),
                    gn.add_calc(DEHALO_CALC_ADD, gn.column(DEHALO_SRC_ADVICE, 1), GSrc{DEHALO_SRC_GAMMA, 0, 0}));
        TRY(gd.compile(ctx, &lookup_den));
        TRY(gn.compile(ctx, &lookup_num));
        return 0;
    }

    // Which value goes where: the transcript's order of the evaluations [UPSTREAM plonk/prover.rs: advice, fixed, vanishing random_eval,
    // permutation (sigma; products), lookups] and the opening queries grouped by point in order of first appearance [UPSTREAM
    // permutation::Constructed::open, lookup::Evaluated::open, pk.permutation.open, vanishing::Evaluated::open; gwc/prover.rs].
    int opening_plan() {
        const HostCS& cs = pk->cs;
        std::vector<int32_t> rs = {0, 1, -1, -(int32_t)(bf + 1)};
        for (auto& q : cs.advice_q) rs.push_back(q.rotation);
        for (auto& q : cs.fixed_q) rs.push_back(q.rotation);
        std::sort(rs.begin(), rs.end());
        rs.erase(std::unique(rs.begin(), rs.end()), rs.end());
        if (rs.size() > 4) return dh_fail(ctx, DEHALO_ERR_UNSUPPORTED, "more than four distinct opening rotations");
        rots = rs;
        const size_t nfix = cs.num_fixed, npc = cs.perm_cols.size();
        hp_ptrs.clear();
        for (uint32_t i = 0; i < pieces; i++) hp_ptrs.push_back((const uint64_t*)h.at(i * n));
        plist.clear();
        for (uint32_t c = 0; c < NC; c++) plist.push_back((const uint64_t*)(polys + c * n));
        for (size_t c = 0; c < nfix; c++) plist.push_back(col_ptr(pk->fixed_polys, c, n));
        for (size_t c = 0; c < npc; c++) plist.push_back(col_ptr(pk->perm_polys, c, n));
        for (auto* p : hp_ptrs) plist.push_back(p);
        const size_t ntot = plist.size();
        const size_t b_cols = 0, b_fixed = NC, b_sigma = NC + nfix, b_hp = NC + nfix + npc;
        auto ridx = [&](int32_t r) { return (size_t)(std::find(rots.begin(), rots.end(), r) - rots.begin()); };
        auto idx = [&](size_t base, size_t col, int32_t r) { return (int64_t)(ridx(r) * ntot + base + col); };
        const int32_t last = -(int32_t)(bf + 1);
        write_idx.clear();
        for (auto& q : cs.advice_q) write_idx.push_back(idx(b_cols, o_adv + q.index, q.rotation));
        for (auto& q : cs.fixed_q) write_idx.push_back(idx(b_fixed, q.index, q.rotation));
        write_idx.push_back(idx(b_cols, o_rand, 0));                                   // vanishing: random_eval
        for (size_t j = 0; j < npc; j++) write_idx.push_back(idx(b_sigma, j, 0));      // pk.permutation.evaluate
        for (uint32_t s = 0; s < S; s++) {                                              // permutation products
            write_idx.push_back(idx(b_cols, o_pz + s, 0));
            write_idx.push_back(idx(b_cols, o_pz + s, 1));
            if (s != S - 1) write_idx.push_back(idx(b_cols, o_pz + s, last));
        }
        for (uint32_t l = 0; l < L; l++) {                                              // lookups
            const size_t zc = o_lz + l, ai = o_perm + 2 * l, ti = o_perm + 2 * l + 1;
            write_idx.push_back(idx(b_cols, zc, 0));
            write_idx.push_back(idx(b_cols, zc, 1));
            write_idx.push_back(idx(b_cols, ai, 0));
            write_idx.push_back(idx(b_cols, ai, -1));
            write_idx.push_back(idx(b_cols, ti, 0));
        }
        struct Q { int32_t r; const uint64_t* ptr; int64_t i; };
        std::vector<Q> qs;
        auto cptr = [&](size_t c) { return (const uint64_t*)(polys + c * n); };
        for (auto& q : cs.advice_q) qs.push_back({q.rotation, cptr(o_adv + q.index), idx(b_cols, o_adv + q.index, q.rotation)});
        for (uint32_t s = 0; s < S; s++) {                                              // permutation::Constructed::open
            qs.push_back({0, cptr(o_pz + s), idx(b_cols, o_pz + s, 0)});
            qs.push_back({1, cptr(o_pz + s), idx(b_cols, o_pz + s, 1)});
        }
        for (int s = (int)S - 2; s >= 0; s--) qs.push_back({last, cptr(o_pz + s), idx(b_cols, o_pz + s, last)});      // sets.iter().rev().skip(1)
        for (uint32_t l = 0; l < L; l++) {                                              // lookup::Evaluated::open
            const size_t zc = o_lz + l, ai = o_perm + 2 * l, ti = o_perm + 2 * l + 1;
            qs.push_back({0, cptr(zc), idx(b_cols, zc, 0)});
            qs.push_back({0, cptr(ai), idx(b_cols, ai, 0)});
            qs.push_back({0, cptr(ti), idx(b_cols, ti, 0)});
            qs.push_back({-1, cptr(ai), idx(b_cols, ai, -1)});
            qs.push_back({1, cptr(zc), idx(b_cols, zc, 1)});
        }
        for (auto& q : cs.fixed_q) qs.push_back({q.rotation, col_ptr(pk->fixed_polys, q.index, n), idx(b_fixed, q.index, q.rotation)});
        for (size_t j = 0; j < npc; j++) qs.push_back({0, col_ptr(pk->perm_polys, j, n), idx(b_sigma, j, 0)});      // pk.permutation.open
        qs.push_back({0, hfold.u64(), -1});                                            // vanishing::Evaluated::open: h, then the random polynomial
        qs.push_back({0, cptr(o_rand), idx(b_cols, o_rand, 0)});
        groups.clear();
        for (auto& q : qs) {
            Group* g = nullptr;
            for (auto& gg : groups)
                if (gg.rot == q.r) { g = &gg; break; }
            if (!g) {
                groups.push_back(Group{q.r, {}, {}});
                g = &groups.back();
            }
            g->ptrs.push_back(q.ptr);
            g->idx.push_back(q.i);
        }
        if (groups.size() > 4) return dh_fail(ctx, DEHALO_ERR_UNSUPPORTED, "more opening points than the prover's buffers hold");
        hpiece0 = (size_t)idx(b_hp, 0, 0);
        eval_count = rots.size() * ntot;
        eval_wanted.assign(ntot, 0);
        auto want = [&](int64_t i) { if (i >= 0) eval_wanted[(size_t)i % ntot] |= (uint8_t)(1u << ((size_t)i / ntot)); };
        for (int64_t i : write_idx) want(i);
        for (auto& g : groups) for (int64_t i : g.idx) want(i);
        for (uint32_t i = 0; i < pieces; i++) want((int64_t)hpiece0 + i);                // the pieces of h at x: the folded quotient's value
        return 0;
    }

    int init(dehalo_ctx* c, dehalo_ctx* s, const dehalo_params* pa, const dehalo_pk* key) {
        ctx = c; side = s; params = pa; pk = key; f = key->f;
        const HostCS& cs = pk->cs;
        const HostDomain& d = pk->dom;
        n = d.n; m = d.m; k = d.k; ek = d.extended_k;
        bf = cs.blinding_factors();
        u = n - (bf + 1);
        A = cs.num_advice; L = (uint32_t)cs.lookups.size(); S = cs.num_sets(); I = cs.num_instance;
        pieces = d.quotient_poly_degree;
        NC = A + 2 * L + S + L + 1;
        o_adv = 0; o_perm = A; o_pz = A + 2 * L; o_lz = A + 2 * L + S; o_rand = A + 2 * L + S + L;
        if ((size_t)pieces * n > m) return dh_fail(ctx, DEHALO_ERR_UNSUPPORTED, "quotient does not fit the extended domain");
        TRY(cols.alloc(ctx, (size_t)NC * n));
        if (side) TRY(polys_own.alloc(ctx, (size_t)NC * n));
        polys = side ? polys_own.p : cols.p;
        TRY(instance.alloc(ctx, (size_t)std::max<uint32_t>(I, 1) * n));
        TRY(instance_values.alloc(ctx, (size_t)std::max<uint32_t>(I, 1) * n));
        TRY(compressed.alloc(ctx, (size_t)std::max<uint32_t>(2 * L, 1) * n));
        TRY(num.alloc(ctx, (size_t)std::max<uint32_t>(S + L, 1) * n));
        TRY(den.alloc(ctx, (size_t)std::max<uint32_t>(S + L, 1) * n));
        TRY(ext.alloc(ctx, (size_t)(NC - 1 + I) * m));
        TRY(h.alloc(ctx, m));
        TRY(table_value.alloc(ctx, (size_t)std::max<uint32_t>(L, 1) * m));
        TRY(hfold.alloc(ctx, n));
        TRY(qbuf.alloc(ctx, 4 * n));
        TRY(wbuf.alloc(ctx, 4 * n));
        TRY(jac.alloc(ctx, 3 * (size_t)std::max<uint32_t>(NC, 8) + 2 + (L + 7) / 8));      // + the lookups' status flags behind a phase's points (one int32 each)
        TRY(jac_side.alloc(ctx, 3));
        // blinding values of a proof but the random polynomial, compacted: [advice rows | permuted rows | product rows]
        const size_t rows = n - u;
        TRY(blind_dev.alloc(ctx, std::max<size_t>(1, (size_t)A * rows + (size_t)2 * L * rows + (size_t)(S + L) * bf)));
        TRY(omega_col.alloc(ctx, n, false));
        TRY(omega_powers(ctx, d, omega_col.p));
        table_rep.clear();
        for (uint32_t l = 0; l < L; l++) table_rep.push_back(cs.table_representative(l));
        TRY(find_table_rows());
        TRY(build_product_graphs());
        TRY(opening_plan());
        TRY(evals.alloc(ctx, eval_count + 8));
        for (hipEvent_t* e : {&ev_ready[0], &ev_ready[1], &ev_ready[2], &ev_inst, &ev_side, &ev_helper}) HIP_TRY(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
        HIP_TRY(ctx, hipHostMalloc((void**)&pin_helper, 128, hipHostMallocDefault));
        HIP_TRY(ctx, hipHostMalloc((void**)&rand_pin, (size_t)n * 32, hipHostMallocDefault));
        if (side) HIP_TRY(ctx, hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
        host_aff.resize(8 * (size_t)std::max<uint32_t>(NC, 8));
        host_jac.resize(12 * (size_t)std::max<uint32_t>(NC, 8) + 8 + L);
        host_evals.resize(4 * eval_count);
        TRY(dehalo_ctx_synchronize(ctx));
        return 0;
    }

    // Which rows of a lookup table are equal does not depend on theta when its expressions read fixed columns only: evaluate them once, on the host, over the usable
    // rows and keep one representative row per distinct tuple + its multiplicity (the delay-encryption circuit's (tag, value) range table: 339 tuples in 131,066 rows).
    Fe eval_fixed_expr(uint32_t node, size_t row, const std::vector<std::vector<Fe>>& colv) const {
        const HostCS& cs = pk->cs;
        const dehalo_expr_node& e = cs.nodes[node];
        switch (e.kind) {
            case DEHALO_EXPR_CONSTANT: return cs.constants[e.a];
            case DEHALO_EXPR_FIXED: return colv[e.a][(size_t)(((int64_t)row + e.rotation) % (int64_t)n + (int64_t)n) % n];
            case DEHALO_EXPR_NEGATED: return f->neg(eval_fixed_expr(e.a, row, colv));
            case DEHALO_EXPR_SCALED: return f->mul(eval_fixed_expr(e.a, row, colv), cs.constants[e.b]);
            case DEHALO_EXPR_SUM: return f->add(eval_fixed_expr(e.a, row, colv), eval_fixed_expr(e.b, row, colv));
            default: return f->mul(eval_fixed_expr(e.a, row, colv), eval_fixed_expr(e.b, row, colv));
        }
    }
    int find_table_rows() {
        const HostCS& cs = pk->cs;
        table_rows.assign(L, TableRows{});
        static const bool enabled = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_PROVER_TABLE_ROWS"); return !(e && e[0] == '0'); }();      // (0: every table sorted in full, for the A/B)
        if (!enabled) return 0;
        std::vector<std::vector<Fe>> colv(cs.num_fixed);
        for (uint32_t l = 0; l < L; l++) {
            if (table_rep[l] != l) continue;
            bool fixed_only = true;
            std::vector<uint32_t> need;
            for (uint32_t e : cs.lookups[l].tables) { fixed_only = fixed_only && cs.expr_fixed_only(e); cs.expr_fixed_columns(e, need); }
            if (!fixed_only) continue;
            for (uint32_t c : need)
                if (colv[c].empty()) {
                    colv[c].resize(n);
                    TRY(dehalo_download(ctx, pk->fixed_values.at((size_t)c * n), n * 32, colv[c].data()));
                }
            const size_t T = cs.lookups[l].tables.size();
            struct Key { std::vector<uint64_t> w; bool operator==(const Key& o) const { return w == o.w; } };
            struct Hash { size_t operator()(const Key& k) const { uint64_t h = 0x9e3779b97f4a7c15ull; for (uint64_t x : k.w) { h ^= x + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); } return (size_t)h; } };
            std::unordered_map<Key, uint32_t, Hash> seen;
            std::vector<uint32_t> rep, mult;
            bool too_many = false;
            Key key;
            key.w.resize(4 * T);
            for (size_t i = 0; i < u && !too_many; i++) {
                for (size_t t = 0; t < T; t++) {
                    const Fe v = eval_fixed_expr(cs.lookups[l].tables[t], i, colv);
                    memcpy(&key.w[4 * t], v.v, 32);
                }
                auto it = seen.find(key);
                if (it != seen.end()) mult[it->second]++;
                else if (rep.size() == 2048) too_many = true;
                else { seen.emplace(key, (uint32_t)rep.size()); rep.push_back((uint32_t)i); mult.push_back(1); }
            }
            if (too_many || rep.empty()) continue;
            TableRows& tr = table_rows[l];
            HIP_TRY(ctx, hipMalloc((void**)&tr.d_rep, rep.size() * 4));
            HIP_TRY(ctx, hipMalloc((void**)&tr.d_mult, rep.size() * 4));
            TRY(dehalo_upload(ctx, rep.data(), rep.size() * 4, tr.d_rep));
            TRY(dehalo_upload(ctx, mult.data(), mult.size() * 4, tr.d_mult));
            tr.count = (uint32_t)rep.size();
        }
        return 0;
    }

    // Jacobian {x, y, z} (Montgomery, base field) -> affine on the HOST: one inversion per phase by Montgomery's trick (a dozen field
    // multiplications per point, one exponentiation per call: ~15 us) instead of a 12 k-instruction safegcd chain on ONE lane of the MSM's last
    // kernel in front of every read-back (~50 us of device latency per commitment phase, six phases per proof).  false: a point at infinity.
    bool normalize_host(const uint64_t* jac, size_t count, uint64_t* affine_out) const {
        const HostField* fq = host_field(curve_base_field(pk->curve));
        std::vector<Fe> z(count), pre(count);
        Fe acc = fq->one;
        for (size_t i = 0; i < count; i++) {
            memcpy(z[i].v, jac + 12 * i + 8, 32);
            if (z[i].is_zero()) return false;
            pre[i] = acc;
            acc = fq->mul(acc, z[i]);
        }
        Fe inv = fq->invert(acc);
        for (size_t i = count; i-- > 0;) {
            const Fe zi = fq->mul(inv, pre[i]);
            inv = fq->mul(inv, z[i]);
            const Fe zi2 = fq->sqr(zi), zi3 = fq->mul(zi2, zi);
            Fe x, y;
            memcpy(x.v, jac + 12 * i, 32);
            memcpy(y.v, jac + 12 * i + 4, 32);
            x = fq->mul(x, zi2);
            y = fq->mul(y, zi3);
            memcpy(affine_out + 8 * i, x.v, 32);
            memcpy(affine_out + 8 * i + 4, y.v, 32);
        }
        return true;
    }

    // commit `count` columns starting at `src`, read back, normalise, absorb (and append to the proof)
    // `flags` > 0: that many int32 status words sit behind the points in `jac` (deferred lookup permutation) and come back with them; any non-zero one fails the call
    int commit(dehalo_transcript* tr, const fe* src, size_t count, bool lagrange, const std::function<int()>& before_sync = nullptr, size_t flags = 0) {
        // sharded (dehalo_prover_set_shard): this process's share of the columns only; everything else of the proof is computed by every process
        const bool sharded = shard_world > 1 && shard_gather && count > 1;
        const size_t lo = sharded ? count * shard_rank / shard_world : 0, hi = sharded ? count * (shard_rank + 1) / shard_world : count;
        if (hi > lo)
            TRY(dehalo_msm_device(ctx, lagrange ? params->bases_gl : params->bases_g, (const uint64_t*)(src + lo * n), n, hi - lo, jac.u64() + 12 * lo, nullptr));
        tk("commit queued");
        if (before_sync) TRY(before_sync());
        tk("side work queued");
        TRY(dehalo_download(ctx, jac.p, count * 96 + flags * 4, host_jac.data()));
        tk("points on host");
        for (size_t i = 0; i < flags; i++)
            if (reinterpret_cast<const int32_t*>(host_jac.data() + 12 * count)[i])
                return dh_fail(ctx, DEHALO_ERR_NOT_IN_TABLE, "permute_expression_pair: an input value of lookup " + std::to_string(i) + " is not in the table (ConstraintSystemFailure)");
        if (hi > lo && !normalize_host(host_jac.data() + 12 * lo, hi - lo, host_aff.data() + 8 * lo)) return dh_fail(ctx, DEHALO_ERR_INVALID, "cannot write points at infinity to the transcript");
        if (sharded) {      // every process ends with all `count` affine points, in column order
            std::vector<uint32_t> first(shard_world), num(shard_world);
            for (uint32_t r = 0; r < shard_world; r++) {
                first[r] = (uint32_t)(count * r / shard_world);
                num[r] = (uint32_t)(count * (r + 1) / shard_world) - first[r];
            }
            if (shard_gather(shard_user, host_aff.data(), (uint32_t)count, first.data(), num.data(), shard_world) != 0)
                return dh_fail(ctx, DEHALO_ERR_INVALID, "the shard gather callback failed");
            tk("points gathered");
        }
        for (size_t i = 0; i < count; i++)
            if (!tr->write_point(host_aff.data() + 8 * i)) return dh_fail(ctx, DEHALO_ERR_INVALID, "cannot write points at infinity to the transcript");
        return 0;
    }

    int run(const uint64_t* advice, const uint64_t* const* instances, const size_t* instance_lens, uint32_t num_instance_columns, dehalo_rng* rng_in,
            dehalo_transcript* tr, uint32_t flags, const dehalo_circuit_inputs* synth_in = nullptr, dehalo_synthesis_info* synth_info = nullptr);
};

namespace {

struct EvalIn {      // dehalo_eval_inputs with owned scalar storage
    dehalo_eval_inputs in{};
    void cols(const std::vector<const uint64_t*>& fixed, const std::vector<const uint64_t*>& advice, const std::vector<const uint64_t*>& instance) {
        in.fixed = fixed.data(); in.num_fixed = (uint32_t)fixed.size();
        in.advice = advice.data(); in.num_advice = (uint32_t)advice.size();
        in.instance = instance.data(); in.num_instance = (uint32_t)instance.size();
    }
};

}   // namespace

int dehalo_prover::run(const uint64_t* advice, const uint64_t* const* instances, const size_t* instance_lens, uint32_t num_instance_columns, dehalo_rng* rng_in,
                       dehalo_transcript* tr, uint32_t flags, const dehalo_circuit_inputs* synth_in, dehalo_synthesis_info* synth_info) {
    const HostCS& cs = pk->cs;
    const HostDomain& d = pk->dom;
    const int fid = f->id;
    const size_t rows = n - u;
    const uint32_t rot_scale = (uint32_t)(m / n);
    const auto t_start = clk::now();
    auto t_phase = t_start;
    ticks.clear();
    t0 = t_start;
    trace = getenv("DEHALO_PROVER_TRACE") != nullptr;
    auto mark = [&](int slot) {
        const auto now = clk::now();
        timings[slot] = std::chrono::duration<double, std::milli>(now - t_phase).count();
        t_phase = now;
    };
    (void)hipSetDevice(ctx->device);
    hipStream_t ms = ctx->stream, ss = side ? side->stream : nullptr;

    // ---- random scalars, in upstream's order: advice blinding rows (column after column), one blind per advice column (unused by KZG), per
    // lookup (bf + 1 rows permuted input, bf + 1 permuted table, two unused blinds), per grand product (bf rows + one unused blind), the random
    // polynomial (n), its blind, the h pieces' blinds
    HostRng rng;
    TRY(rng.init(rng_in, f));
    const size_t c_adv = (size_t)A * rows, c_advb = A, c_lk = (size_t)L * (2 * rows + 2), c_pr = (size_t)(S + L) * (bf + 1);
    const size_t draws_before = c_adv + c_advb + c_lk + c_pr;
    blind_host.resize(4 * std::max<size_t>(1, draws_before));
    // the one large draw (n scalars, drawn AFTER every blinding value) is produced, uploaded and -- with a side context -- committed by a helper
    // thread while the earlier phases run
    HostRng rng_poly = rng.fork(draws_before, 1);
    TRY(rng.scalars(blind_host.data(), draws_before));
    std::atomic<int> helper_rc{0};
    uint64_t rand_point[8] = {};
    std::thread helper;
    double helper_ms[3] = {};
    const bool device_rng = rng.kind == DEHALO_RNG_OS;      // the large draw comes from a ChaCha20 kernel keyed with this proof's entropy
    // The random polynomial's commitment.  Committing COEFFICIENTS over g equals committing their forward transform (the values on the domain) over g_lagrange,
    // and upstream writes the point right behind the grand products' commitments with no challenge in between: so (round 4, with a side context) the helper only
    // draws and uploads, and the polynomial rides as ONE MORE COLUMN of the products' MSM launch -- a whole sort / accumulate / merge / reduce pipeline per proof
    // less, and none running beside the lookups' phase.  DEHALO_PROVER_RANDOM_SEPARATE=1: the helper commits it with an MSM of its own, as in round 3 (A/B measurements).
    static const bool random_separate_env = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_PROVER_RANDOM_SEPARATE"); return e && e[0] == '1'; }();
    const bool random_separate = random_separate_env;
    auto device_draw = [&](fe* dst, hipStream_t st) -> int {
        ChaKey ck;
        memcpy(ck.k, rng.key, 32);
        fe pw;
        for (int i = 0; i < 4; i++) { pw.v[2 * i] = (u32)f->p[i]; pw.v[2 * i + 1] = (u32)(f->p[i] >> 32); }
        const u32 top_mask = f->bits >= 256 ? 0xffffffffu : ((1u << (f->bits - 224)) - 1);
        k_chacha_scalars<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ck, /* stream of the fork */ 1, pw, top_mask, dst, n);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };
    auto helper_body = [&]() {
        (void)hipSetDevice(ctx->device);
        const auto th0 = clk::now();
        int rc = 0;
        if (!device_rng) {
            rc = rng_poly.scalars(rand_pin, n);
        }
        helper_ms[0] = ms_since(th0);
        if (!rc) {
            if (side) {
                fe* dst = polys + (size_t)o_rand * n;
                hipError_t e = hipSuccess;
                if (device_rng) rc = device_draw(dst, hs);
                else e = hipMemcpyAsync(dst, rand_pin, n * 32, hipMemcpyHostToDevice, hs);
                if (e != hipSuccess) rc = dh_fail(side, DEHALO_ERR_HIP, std::string("random polynomial upload: ") + hipGetErrorString(e));
                if (!rc && random_separate) rc = dehalo_msm_device(side, params->bases_g, (const uint64_t*)dst, n, 1, jac_side.u64(), hs);      // (the side context's MSM workspace is this thread's alone)
                helper_ms[1] = ms_since(th0);
                if (!rc) {      // wait for this stream through an event of this thread's own: the coefficients (and the point) are there before the join
                    if (random_separate) e = hipMemcpyAsync(pin_helper, jac_side.p, 96, hipMemcpyDeviceToHost, hs);
                    if (e == hipSuccess) e = hipEventRecord(ev_helper, hs);
                    if (e == hipSuccess) e = hipEventSynchronize(ev_helper);
                    if (e != hipSuccess) rc = dh_fail(side, DEHALO_ERR_HIP, std::string("random polynomial: ") + hipGetErrorString(e));
                }
                if (!rc && random_separate && !normalize_host(pin_helper, 1, rand_point)) memset(rand_point, 0, sizeof rand_point);      // (the identity: refused by write_point below)
                helper_ms[2] = ms_since(th0);
            } else {
                // without a side context only the draw is taken off the critical path; the upload is queued by the proving thread
            }
        }
        helper_rc.store(rc);
    };
    struct Joiner {      // the helper must have finished before this call returns, whatever path it takes
        std::thread& t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{helper};

    // create_proof of a CIRCUIT (upstream's call synthesizes inside): the random polynomial's draw, upload and commitment start now, on the helper's
    // thread and stream, and run on an otherwise idle device while this thread (and the synthesis pool) writes the advice columns
    if (synth_in) {
        if (synth_in->k != k || A != 5) return dh_fail(ctx, DEHALO_ERR_INVALID, "create_proof_circuit: the circuit's k / advice columns differ from the key's");
        if (!adv_pin) HIP_TRY(ctx, hipHostMalloc((void**)&adv_pin, (size_t)A * n * 32, hipHostMallocDefault));
        helper = std::thread(helper_body);
        const int src = dehalo_synthesize(synth_in, adv_pin, nullptr, nullptr, nullptr, synth_info);
        if (src) return dh_fail(ctx, src, "create_proof_circuit: the circuit's inputs are invalid or it does not fit 2^k rows");
        advice = adv_pin;
        flags = (flags & ~(uint32_t)DEHALO_PROOF_ADVICE_ON_DEVICE) | DEHALO_PROOF_ADVICE_CANONICAL;
        tk("witness synthesized");
    }

    // compacted blinding rows -> one upload
    {
        const size_t total = (size_t)A * rows + (size_t)2 * L * rows + (size_t)(S + L) * bf;
        std::vector<uint64_t>& b = blind_host;
        std::vector<uint64_t> packed(4 * std::max<size_t>(1, total));
        size_t o = 0;
        memcpy(packed.data(), b.data(), 32 * c_adv);
        o += c_adv;
        const uint64_t* lk = b.data() + 4 * (c_adv + c_advb);
        for (uint32_t l = 0; l < L; l++) {      // (input rows, table rows, two unused blinds) per lookup
            memcpy(packed.data() + 4 * o, lk + 4 * (size_t)l * (2 * rows + 2), 32 * 2 * rows);
            o += 2 * rows;
        }
        const uint64_t* pr = b.data() + 4 * (c_adv + c_advb + c_lk);
        for (uint32_t s = 0; s < S + L; s++) {      // (bf rows, one unused blind) per product
            memcpy(packed.data() + 4 * o, pr + 4 * (size_t)s * (bf + 1), 32 * bf);
            o += bf;
        }
        if (total) {
            TRY(dh_h2d(ctx, blind_dev.p, packed.data(), total * 32, ms));
            HIP_TRY(ctx, hipStreamSynchronize(ms));      // `packed` is a local
        }
    }
    tk("blinds drawn and uploaded");
    const fe* bl_adv = blind_dev.p;
    const fe* bl_perm = blind_dev.p + (size_t)A * rows;
    const fe* bl_prod = bl_perm + (size_t)2 * L * rows;

    tr->common_scalar(pk->transcript_repr);      // vk.hash_into
    // ---- instance columns: values into the transcript (KZG: QUERY_INSTANCE = false), polynomials on the device
    if (num_instance_columns != I) return dh_fail(ctx, DEHALO_ERR_INVALID, "instances.len() != num_instance_columns");      // Error::InvalidInstances
    if (I) HIP_TRY(ctx, hipMemsetAsync(instance.p, 0, (size_t)I * n * 32, ms));
    for (uint32_t i = 0; i < I; i++) {
        const size_t len = instance_lens ? instance_lens[i] : 0;
        if (len > u) return dh_fail(ctx, DEHALO_ERR_INVALID, "instance column too long");      // Error::InstanceTooLarge
        if (len && (!instances || !instances[i])) return dh_fail(ctx, DEHALO_ERR_INVALID, "null instance column");
        for (size_t j = 0; j < len; j++) {
            Fe v;
            memcpy(v.v, instances[i] + 4 * j, 32);
            tr->common_scalar(v);
        }
        if (len) {
            TRY(dh_h2d(ctx, instance.at((size_t)i * n), instances[i], len * 32, ms));
            HIP_TRY(ctx, hipStreamSynchronize(ms));
        }
    }
    if (I) HIP_TRY(ctx, hipMemcpyAsync(instance_values.p, instance.p, (size_t)I * n * 32, hipMemcpyDeviceToDevice, ms));
    const uint32_t nco = NC - 1;
    if (I && side) HIP_TRY(ctx, hipEventRecord(ev_inst, ms));
    if (I && !side) TRY(dehalo_intt_scaled_device(ctx, fid, instance.u64(), k, d.omega_inv.v, d.ifft_divisor.v, I, nullptr));

    auto side_ntt = [&](uint32_t first, uint32_t count, hipEvent_t e) -> int {
        // polys[first..] = lagrange_to_coeff(cols[..]), ext[..] = coeff_to_extended(..) on the side context once `e` (recorded BEFORE the phase's
        // commitment was queued) has passed; the launches themselves are made after the commitment's, while this thread would only wait
        HIP_TRY(side, hipStreamWaitEvent(ss, e, 0));
        // (out of place: no copy of the columns first -- a 42 MB device-to-device hipMemcpyAsync held this thread for 0.4 ms in the lookups' phase)
        TRY(dehalo_lagrange_to_coeff_device(side, fid, (const uint64_t*)cols.at((size_t)first * n), (uint64_t*)(polys + (size_t)first * n), k, d.omega_inv.v, d.ifft_divisor.v, count,
                                            nullptr));
        TRY(dehalo_coset_ntt_form_device(side, fid, (const uint64_t*)(polys + (size_t)first * n), k, ext.u64((size_t)first * m), ek, d.ext_omega.v, d.g_coset.v, count,
                                         DEHALO_FORM_OUT_INTERNAL, nullptr));
        return 0;
    };

    // ---- advice: witness, blinding rows, commitments
    if (!advice) return dh_fail(ctx, DEHALO_ERR_INVALID, "null advice");
    // a host witness is pinned until this call returns (every stream has been synchronised by then); one that is page-locked already stays as it is
    struct PinnedUpload {      // (the copy below is asynchronous: on every way out, wait for its stream before the pin goes)
        HostPin pin;
        hipStream_t s;
        PinnedUpload(const void* ptr, size_t bytes, hipStream_t s_) : pin(ptr, bytes), s(s_) {}
        ~PinnedUpload() { if (pin.p) (void)hipStreamSynchronize(s); }
    } pin_advice((flags & DEHALO_PROOF_ADVICE_ON_DEVICE) || synth_in ? nullptr : advice, (size_t)A * n * 32, ms);
    if (flags & DEHALO_PROOF_ADVICE_ON_DEVICE) HIP_TRY(ctx, hipMemcpyAsync(cols.at((size_t)o_adv * n), advice, (size_t)A * n * 32, hipMemcpyDeviceToDevice, ms));
    else TRY(dh_h2d(ctx, cols.at((size_t)o_adv * n), advice, (size_t)A * n * 32, ms));      // a DMA from the pinned pages, or staged (witness below 4 MiB)
    if (flags & DEHALO_PROOF_ADVICE_CANONICAL) TRY(dehalo_field_op_device(ctx, fid, 4, cols.u64((size_t)o_adv * n), nullptr, cols.u64((size_t)o_adv * n), (size_t)A * n, nullptr));
    if (A) k_place_rows<<<(unsigned)((rows * A + 255) / 256), 256, 0, ms>>>(cols.at((size_t)o_adv * n + u), n, bl_adv, (uint32_t)rows, A);
    if (side) HIP_TRY(ctx, hipEventRecord(ev_ready[0], ms));

    // pointer tables of the phases
    std::vector<const uint64_t*> fixed_v, adv_v, inst_v, fixed_c, adv_c, inst_c, none;
    for (uint32_t i = 0; i < cs.num_fixed; i++) fixed_v.push_back(col_ptr(pk->fixed_values, i, n)), fixed_c.push_back(col_ptr(pk->fixed_cosets, i, m));
    for (uint32_t i = 0; i < A; i++) adv_v.push_back(col_ptr(cols, o_adv + i, n)), adv_c.push_back(col_ptr(ext, o_adv + i, m));
    for (uint32_t i = 0; i < I; i++) inst_v.push_back(col_ptr(instance_values, i, n)), inst_c.push_back(col_ptr(ext, nco + i, m));
    const uint32_t FF = DEHALO_EVAL_COLUMNS_INTERNAL | DEHALO_EVAL_VALUES_INTERNAL;
    // one gate polynomial: Horner(0, [g], y) = g does not depend on y, so the custom-gate pass of evaluate_h needs the advice (and instance)
    // cosets only -- queued on the side context right behind them, long before y exists
    const bool gates_early = side && cs.gates.size() == 1;
    const Fe zero{{0, 0, 0, 0}};

    auto after_advice_queued = [&]() -> int {
        if (I && side) {
            HIP_TRY(side, hipStreamWaitEvent(ss, ev_inst, 0));
            TRY(dehalo_intt_scaled_device(side, fid, instance.u64(), k, d.omega_inv.v, d.ifft_divisor.v, I, nullptr));
            TRY(dehalo_coset_ntt_form_device(side, fid, instance.u64(), k, ext.u64((size_t)nco * m), ek, d.ext_omega.v, d.g_coset.v, I, DEHALO_FORM_OUT_INTERNAL, nullptr));
        }
        if (side) TRY(side_ntt(o_adv, A, ev_ready[0]));
        if (gates_early) {
            EvalIn e;
            e.cols(fixed_c, adv_c, inst_c);
            e.in.form_flags = FF;
            e.in.y = zero.v;
            TRY(dehalo_graph_evaluate_device(side, pk->custom_gates, &e.in, ek, rot_scale, nullptr, h.u64(), nullptr));
        }
        if (!synth_in) helper = std::thread(helper_body);      // the host is idle from here to the read-back
        return 0;
    };
    TRY(commit(tr, cols.at((size_t)o_adv * n), A, true, after_advice_queued));
    mark(0);
    const Fe theta = tr->squeeze();
    tk("theta");

    // ---- lookups: compress, permute, blind, commit
    if (L) {
        std::vector<const dehalo_graph*> graphs;
        std::vector<uint64_t*> outs;
        for (uint32_t l = 0; l < L; l++) {      // lookups with the same table expressions share ONE compressed table column (their representative's)
            graphs.push_back(pk->compress_graphs[l].first);
            outs.push_back(compressed.u64((size_t)2 * l * n));
            if (table_rep[l] == l) {
                graphs.push_back(pk->compress_graphs[l].second);
                outs.push_back(compressed.u64((size_t)(2 * l + 1) * n));
            }
        }
        EvalIn e;
        e.cols(fixed_v, adv_v, inst_v);
        e.in.theta = theta.v;
        TRY(dehalo_graph_evaluate_batch_device(ctx, graphs.data(), (uint32_t)graphs.size(), &e.in, k, 1, outs.data(), nullptr));
        tk("compress queued");
        // the blinding rows [u, n) first: the permutation writes rows [0, u) only and ends with a read-back
        k_place_rows<<<(unsigned)((rows * 2 * L + 255) / 256), 256, 0, ms>>>(cols.at((size_t)o_perm * n + u), n, bl_perm, (uint32_t)rows, 2 * L);
        std::vector<const uint64_t*> pin, ptab;
        std::vector<uint64_t*> pout_in, pout_tab;
        for (uint32_t l = 0; l < L; l++) {
            pin.push_back(compressed.u64((size_t)2 * l * n));
            ptab.push_back(compressed.u64((size_t)(2 * table_rep[l] + 1) * n));
            pout_in.push_back(cols.u64((size_t)(o_perm + 2 * l) * n));
            pout_tab.push_back(cols.u64((size_t)(o_perm + 2 * l + 1) * n));
        }
        // status flags behind the 2 L points of this phase: the stream runs from the permutation straight into the commitment, the flags come back with the points
        std::vector<const uint32_t*> trep, tmult;
        std::vector<uint32_t> tcount;
        for (uint32_t l = 0; l < L; l++) {
            const TableRows& t = table_rows[table_rep[l]];
            trep.push_back(t.d_rep); tmult.push_back(t.d_mult); tcount.push_back(t.count);
        }
        TRY(dehalo_permute_expression_pair_distinct_device(ctx, fid, pin.data(), ptab.data(), u, L, pout_in.data(), pout_tab.data(), trep.data(), tmult.data(), tcount.data(),
                                                           reinterpret_cast<int32_t*>(jac.u64() + 12 * 2 * (size_t)L), nullptr));
        tk("permute queued");
        if (side) HIP_TRY(ctx, hipEventRecord(ev_ready[1], ms));
        TRY(commit(tr, cols.at((size_t)o_perm * n), 2 * L, true, side ? std::function<int()>([&]() { return side_ntt(o_perm, 2 * L, ev_ready[1]); }) : nullptr, L));
    }
    mark(1);
    const Fe beta = tr->squeeze();
    const Fe gamma = tr->squeeze();
    tk("beta gamma");

    // ---- grand products: permutation sets, then lookups; one batched inversion
    const uint32_t npc = (uint32_t)cs.perm_cols.size();
    const uint32_t extra = random_separate ? 0u : 1u;      // the random polynomial's values as the launch's last column (cols[o_rand] sits right behind the products)
    if (extra) {
        if (helper.joinable()) helper.join();              // (long finished: the draw takes 0.5 ms at k = 17 and started before the advice commitment)
        tk("helper joined");
        if (helper_rc.load()) return helper_rc.load();
        fe* rl = cols.at((size_t)o_rand * n);
        if (side) HIP_TRY(ctx, hipMemcpyAsync(rl, polys + (size_t)o_rand * n, n * sizeof(fe), hipMemcpyDeviceToDevice, ms));      // (the helper put the coefficients there)
        else {      // without a side context the coefficient forms live in `cols` itself: keep a copy for after the commitment
            if (device_rng) TRY(device_draw(rl, ms));
            else HIP_TRY(ctx, hipMemcpyAsync(rl, rand_pin, n * 32, hipMemcpyHostToDevice, ms));      // (rand_pin: page-locked, the library's own)
            HIP_TRY(ctx, hipMemcpyAsync(wbuf.p, rl, n * sizeof(fe), hipMemcpyDeviceToDevice, ms));
        }
        TRY(dehalo_ntt_device(ctx, fid, (uint64_t*)rl, k, d.omega.v, 1, nullptr));
    }
    auto restore_random = [&]() -> int {      // (queued behind the MSM's kernels on the same stream)
        if (extra && !side) HIP_TRY(ctx, hipMemcpyAsync(cols.at((size_t)o_rand * n), wbuf.p, n * sizeof(fe), hipMemcpyDeviceToDevice, ms));
        return 0;
    };
    if (S + L == 0 && extra) {
        TRY(commit(tr, cols.at((size_t)o_rand * n), 1, true, [&]() { return restore_random(); }));
    }
    if (S + L) {
        std::vector<Fe> chal(std::max<uint32_t>(npc, 1));
        Fe dj = beta;
        for (uint32_t j = 0; j < npc; j++) {
            chal[j] = dj;
            dj = f->mul(dj, f->delta);
        }
        // every product's numerator and denominator columns in ONE launch (k_product_terms) instead of a GraphEvaluator program per column
        std::vector<const uint64_t*> pcolv, psig, pA, pS, pa, ps;
        for (auto& pc : cs.perm_cols) pcolv.push_back(pc.kind == DEHALO_COLUMN_ADVICE ? adv_v[pc.index] : pc.kind == DEHALO_COLUMN_FIXED ? fixed_v[pc.index] : inst_v[pc.index]);
        for (uint32_t j = 0; j < npc; j++) psig.push_back(col_ptr(pk->perm_values, j, n));
        for (uint32_t l = 0; l < L; l++) {
            pA.push_back(col_ptr(compressed, 2 * l, n));
            pS.push_back(col_ptr(compressed, 2 * table_rep[l] + 1, n));
            pa.push_back(col_ptr(cols, o_perm + 2 * l, n));
            ps.push_back(col_ptr(cols, o_perm + 2 * l + 1, n));
        }
        std::vector<Fe> set_factors(std::max<uint32_t>(S, 1));
        for (uint32_t s2 = 0; s2 < S; s2++) set_factors[s2] = chal[std::min<uint32_t>(s2 * cs.chunk_len(), npc ? npc - 1 : 0)];      // beta delta^(first column of the set)
        dehalo_product_inputs pin{};
        pin.columns = pcolv.data(); pin.sigma = psig.data(); pin.num_columns = npc; pin.chunk_len = cs.chunk_len();
        pin.omega_powers = omega_col.u64();
        pin.beta = beta.v; pin.gamma = gamma.v; pin.delta = f->delta.v;
        pin.set_factors = (const uint64_t*)set_factors.data();
        pin.compressed_input = pA.data(); pin.compressed_table = pS.data(); pin.permuted_input = pa.data(); pin.permuted_table = ps.data();
        pin.num_lookups = L;
        TRY(dehalo_product_terms_device(ctx, fid, &pin, n, num.u64(), den.u64(), n, nullptr));
        tk("product graphs queued");
        TRY(dehalo_grand_product_batch_device(ctx, fid, num.u64(), den.u64(), n, S + L, n, cols.u64((size_t)o_pz * n), nullptr));
        for (uint32_t s = 1; s < S; s++)      // z_s starts where z_{s-1} ended: z = vec![last_z]
            TRY(dehalo_scale_device(ctx, fid, cols.u64((size_t)(o_pz + s) * n), n, nullptr, 0, cols.u64((size_t)(o_pz + s - 1) * n + u), nullptr));
        // per column: bf blinding rows (n - bf .. n)
        k_place_rows<<<(unsigned)(((size_t)bf * (S + L) + 255) / 256), 256, 0, ms>>>(cols.at((size_t)o_pz * n + (n - bf)), n, bl_prod, bf, S + L);
        if (side) HIP_TRY(ctx, hipEventRecord(ev_ready[2], ms));
        auto after_products_queued = [&]() -> int {
            TRY(restore_random());
            if (!side) return 0;
            TRY(side_ntt(o_pz, S + L, ev_ready[2]));
            // the lookups' (compressed input + beta)(compressed table + gamma) over the extended domain need theta, beta, gamma and the advice /
            // fixed cosets: all there -- on the side context, beside the products' commitment, instead of after y
            if (L) {
                std::vector<const dehalo_graph*> graphs(pk->lookup_graphs.begin(), pk->lookup_graphs.end());
                std::vector<uint64_t*> outs;
                for (uint32_t l = 0; l < L; l++) outs.push_back(table_value.u64((size_t)l * m));
                EvalIn e;
                e.cols(fixed_c, adv_c, inst_c);
                e.in.beta = beta.v; e.in.gamma = gamma.v; e.in.theta = theta.v;
                e.in.form_flags = FF;
                TRY(dehalo_graph_evaluate_batch_device(side, graphs.data(), L, &e.in, ek, rot_scale, outs.data(), nullptr));
            }
            return 0;
        };
        TRY(commit(tr, cols.at((size_t)o_pz * n), S + L + extra, true, std::function<int()>(after_products_queued)));
    }
    mark(2);

    // ---- vanishing argument: a random polynomial
    tk("products done");
    if (helper.joinable()) helper.join();
    tk("helper joined");
    if (trace) fprintf(stderr, "  helper: draw %.3f, upload + commit queued %.3f, point on host %.3f ms after its start\n", helper_ms[0], helper_ms[1], helper_ms[2]);
    if (helper_rc.load()) return helper_rc.load();
    rng.skip(n);
    {
        uint64_t blind[4];
        TRY(rng.scalars(blind, 1));      // random_blind (unused by KZG)
    }
    if (!random_separate) {
        // (written with the products' commitments above)
    } else if (side) {
        if (!tr->write_point(rand_point)) return dh_fail(ctx, DEHALO_ERR_INVALID, "cannot write points at infinity to the transcript");
    } else {
        if (device_rng) TRY(device_draw(cols.at((size_t)o_rand * n), ms));
        else HIP_TRY(ctx, hipMemcpyAsync(cols.at((size_t)o_rand * n), rand_pin, n * 32, hipMemcpyHostToDevice, ms));      // (rand_pin: page-locked, the library's own)
        TRY(commit(tr, cols.at((size_t)o_rand * n), 1, false));
    }
    mark(3);
    const Fe y = tr->squeeze();
    tk("y");

    // ---- coefficient forms and cosets of everything committed so far
    if (!side) {
        TRY(dehalo_intt_scaled_device(ctx, fid, cols.u64(), k, d.omega_inv.v, d.ifft_divisor.v, nco, nullptr));
        TRY(dehalo_coset_ntt_form_device(ctx, fid, cols.u64(), k, ext.u64(), ek, d.ext_omega.v, d.g_coset.v, nco, DEHALO_FORM_OUT_INTERNAL, nullptr));
        if (I) TRY(dehalo_coset_ntt_form_device(ctx, fid, instance.u64(), k, ext.u64((size_t)nco * m), ek, d.ext_omega.v, d.g_coset.v, I, DEHALO_FORM_OUT_INTERNAL, nullptr));
    } else {      // queued phase by phase on the side context: wait for it
        HIP_TRY(side, hipEventRecord(ev_side, ss));
        HIP_TRY(ctx, hipStreamWaitEvent(ms, ev_side, 0));
    }

    // ---- evaluate_h, / t(X), extended_to_coeff
    const uint64_t *l0 = pk->l_ext.u64(0), *l_last = pk->l_ext.u64(m), *l_active = pk->l_ext.u64(2 * m);
    if (!gates_early) {
        EvalIn e;
        e.cols(fixed_c, adv_c, inst_c);
        e.in.y = y.v;
        e.in.form_flags = FF;
        TRY(dehalo_graph_evaluate_device(ctx, pk->custom_gates, &e.in, ek, rot_scale, nullptr, h.u64(), nullptr));
    }
    if (S) {
        std::vector<const uint64_t*> z, pcols, sigma;
        for (uint32_t s = 0; s < S; s++) z.push_back(col_ptr(ext, o_pz + s, m));
        for (auto& pc : cs.perm_cols) pcols.push_back(pc.kind == DEHALO_COLUMN_ADVICE ? adv_c[pc.index] : pc.kind == DEHALO_COLUMN_FIXED ? fixed_c[pc.index] : inst_c[pc.index]);
        for (uint32_t j = 0; j < npc; j++) sigma.push_back(col_ptr(pk->perm_cosets, j, m));
        const Fe beta_zeta = f->mul(beta, d.g_coset);
        dehalo_perm_inputs pi{};
        pi.z = z.data(); pi.num_sets = S;
        pi.columns = pcols.data(); pi.sigma = sigma.data(); pi.num_columns = npc;
        pi.chunk_len = cs.chunk_len();
        pi.last_rotation = -(int32_t)(bf + 1);
        pi.l0 = l0; pi.l_last = l_last; pi.l_active_row = l_active;
        pi.beta = beta.v; pi.gamma = gamma.v; pi.y = y.v; pi.delta = f->delta.v; pi.beta_zeta = beta_zeta.v; pi.extended_omega = d.ext_omega.v;
        pi.form_flags = FF;
        TRY(dehalo_permutation_h_device(ctx, fid, &pi, ek, rot_scale, h.u64(), nullptr));
    }
    if (L) {
        if (!side) {
            for (uint32_t l = 0; l < L; l++) {
                EvalIn e;
                e.cols(fixed_c, adv_c, inst_c);
                e.in.beta = beta.v; e.in.gamma = gamma.v; e.in.theta = theta.v;
                e.in.form_flags = FF;
                TRY(dehalo_graph_evaluate_device(ctx, pk->lookup_graphs[l], &e.in, ek, rot_scale, nullptr, table_value.u64((size_t)l * m), nullptr));
            }
        }
        for (uint32_t first = 0; first < L; first += 8) {
            std::vector<dehalo_lookup_inputs> li;
            for (uint32_t l = first; l < std::min(L, first + 8); l++) {
                dehalo_lookup_inputs x{};
                x.product_coset = col_ptr(ext, o_lz + l, m);
                x.permuted_input_coset = col_ptr(ext, o_perm + 2 * l, m);
                x.permuted_table_coset = col_ptr(ext, o_perm + 2 * l + 1, m);
                x.table_value = table_value.u64((size_t)l * m);
                x.l0 = l0; x.l_last = l_last; x.l_active_row = l_active;
                x.beta = beta.v; x.gamma = gamma.v; x.y = y.v;
                x.form_flags = FF;
                li.push_back(x);
            }
            TRY(dehalo_lookup_h_batch_device(ctx, fid, li.data(), (uint32_t)li.size(), ek, rot_scale, h.u64(), nullptr));
        }
    }
    TRY(dehalo_scale_device(ctx, fid, h.u64(), m, (const uint64_t*)d.t_inv.data(), (uint32_t)d.t_inv.size(), nullptr, nullptr));      // divide_by_vanishing_poly
    TRY(dehalo_coset_intt_form_device(ctx, fid, h.u64(), ek, d.ext_omega_inv.v, d.ext_ifft_divisor.v, d.g_coset.v, 1, DEHALO_FORM_IN_INTERNAL, nullptr));
    {
        std::vector<uint64_t> hb(4 * (size_t)pieces);
        TRY(rng.scalars(hb.data(), pieces));      // h_blinds (unused by KZG)
    }
    tk("quotient queued");
    TRY(commit(tr, h.p, pieces, false));
    mark(4);
    const Fe x = tr->squeeze();
    const Fe xn = f->pow_u64(x, (uint64_t)n);

    // ---- evaluations, in upstream's order: every opened polynomial at every rotation in ONE call
    std::vector<Fe> point(rots.size());
    for (size_t i = 0; i < rots.size(); i++) point[i] = d.rotate_omega(x, rots[i]);
    TRY(dehalo_eval_polynomial_multi_masked_device(ctx, fid, plist.data(), plist.size(), n, (const uint64_t*)point.data(), (uint32_t)rots.size(), eval_wanted.data(), evals.u64(), nullptr));
    // the folded quotient h(X) = sum_i x^(n i) h_i(X) (opened below; its value at x comes from the pieces' values)
    std::vector<Fe> xs(pieces);
    {
        Fe cur = f->one;
        for (uint32_t i = 0; i < pieces; i++) {
            xs[i] = cur;
            cur = f->mul(cur, xn);
        }
    }
    TRY(dehalo_lincomb_device(ctx, fid, hp_ptrs.data(), (const uint64_t*)xs.data(), pieces, n, hfold.u64(), nullptr, nullptr));
    tk("evaluations queued");
    TRY(dehalo_download(ctx, evals.p, eval_count * 32, host_evals.data()));
    tk("evaluations on host");
    const Fe* E = (const Fe*)host_evals.data();
    Fe hfold_eval = zero;
    for (uint32_t i = 0; i < pieces; i++) hfold_eval = f->add(hfold_eval, f->mul(xs[i], E[hpiece0 + i]));
    for (int64_t i : write_idx) tr->write_scalar(E[i]);
    mark(5);

    // ---- ProverGWC::create_proof: one witness polynomial per distinct point, in order of first appearance
    const Fe v = tr->squeeze();
    tk("v");
    HIP_TRY(ctx, hipMemsetAsync(wbuf.p, 0, 4 * n * 32, ms));
    std::vector<const uint64_t*> qptrs;
    std::vector<uint64_t*> wptrs;
    std::vector<Fe> qpoints;
    for (size_t gi = 0; gi < groups.size(); gi++) {
        const Group& g = groups[gi];
        std::vector<Fe> coefs(g.idx.size());
        Fe eval_batch = zero, pw = f->one;
        for (size_t i = 0; i < g.idx.size(); i++) {
            coefs[i] = pw;
            eval_batch = f->add(eval_batch, f->mul(pw, g.idx[i] >= 0 ? E[g.idx[i]] : hfold_eval));
            pw = f->mul(pw, v);
        }
        TRY(dehalo_lincomb_device(ctx, fid, g.ptrs.data(), (const uint64_t*)coefs.data(), g.ptrs.size(), n, qbuf.u64(gi * n), eval_batch.v, nullptr));
        qptrs.push_back(qbuf.u64(gi * n));
        wptrs.push_back(wbuf.u64(gi * n));
        qpoints.push_back(point[(size_t)(std::find(rots.begin(), rots.end(), g.rot) - rots.begin())]);
    }
    TRY(dehalo_kate_division_batch_device(ctx, fid, qptrs.data(), n, (const uint64_t*)qpoints.data(), wptrs.data(), groups.size(), nullptr));
    TRY(commit(tr, wbuf.p, groups.size(), false));
    mark(6);
    timings[7] = ms_since(t_start);
    if (trace) {
        double prev = 0;
        for (auto& t : ticks) {
            fprintf(stderr, "  %8.3f (+%6.3f) %s\n", t.second, t.second - prev, t.first);
            prev = t.second;
        }
        fprintf(stderr, "  %8.3f total\n", timings[7]);
    }
    // a PCG64 caller's generator moves past this proof's draws (upstream's `&mut rng`)
    if (rng_in && rng_in->kind == DEHALO_RNG_PCG64) {
        rng_in->pcg_state[0] = (uint64_t)rng.pcg.state;
        rng_in->pcg_state[1] = (uint64_t)(rng.pcg.state >> 64);
    }
    return 0;
}

extern "C" int dehalo_prover_create(dehalo_ctx* ctx, dehalo_ctx* side_ctx, const dehalo_params* params, const dehalo_pk* pk, dehalo_prover** out) try {
    if (!ctx || !params || !pk || !out) return dh_fail(ctx, DEHALO_ERR_INVALID, "prover_create: null argument");
    if (params->k != pk->k || params->curve != pk->curve) return dh_fail(ctx, DEHALO_ERR_INVALID, "prover_create: params and proving key disagree on k / curve");
    if (side_ctx && (side_ctx == ctx || side_ctx->device != ctx->device)) return dh_fail(ctx, DEHALO_ERR_INVALID, "prover_create: the side context must be another context of the same device");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    std::unique_ptr<dehalo_prover> p(new dehalo_prover);
    TRY(p->init(ctx, side_ctx, params, pk));
    *out = p.release();
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_prover_release(dehalo_prover* p) try {
    if (!p) return 0;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->side) (void)hipStreamSynchronize(p->side->stream);
    delete p;
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_create_proof(dehalo_prover* p, const uint64_t* advice, const uint64_t* const* instances, const size_t* instance_lens, uint32_t num_instance_columns,
                                   dehalo_rng* rng, dehalo_transcript* transcript, uint32_t flags) try {
    if (!p || !transcript) return DEHALO_ERR_INVALID;
    if (transcript->curve != p->pk->curve) return dh_fail(p->ctx, DEHALO_ERR_INVALID, "create_proof: the transcript's curve differs from the key's");
    std::lock_guard<std::mutex> lk(p->mu);
    const int rc = p->run(advice, instances, instance_lens, num_instance_columns, rng, transcript, flags);
    if (rc) {      // leave nothing of this proof in flight on either context
        (void)hipStreamSynchronize(p->ctx->stream);
        if (p->side) (void)hipStreamSynchronize(p->side->stream);
    }
    return rc;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_prover_set_shard(dehalo_prover* p, uint32_t rank, uint32_t world, dehalo_gather_fn gather, void* user) try {
    if (!p || world == 0 || rank >= world || (world > 1 && !gather)) return DEHALO_ERR_INVALID;
    p->shard_rank = rank; p->shard_world = world; p->shard_gather = world > 1 ? gather : nullptr; p->shard_user = user;
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_prover_last_timings(const dehalo_prover* p, double out[8]) try {
    if (!p || !out) return DEHALO_ERR_INVALID;
    memcpy(out, p->timings, sizeof p->timings);
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_pk_info(const dehalo_pk* pk, uint32_t out[8]) try {
    if (!pk || !out) return DEHALO_ERR_INVALID;
    const HostCS& cs = pk->cs;
    const uint32_t L = (uint32_t)cs.lookups.size(), S = cs.num_sets();
    out[0] = pk->k;
    out[1] = pk->dom.extended_k;
    out[2] = cs.blinding_factors();
    out[3] = cs.degree();
    out[4] = S;
    out[5] = cs.num_advice + 2 * L + S + L + 1 + (cs.degree() - 1);
    out[6] = (uint32_t)(cs.advice_q.size() + cs.fixed_q.size() + 1 + cs.perm_cols.size() + (S ? 3 * S - 1 : 0) + 5 * L);
    std::vector<int32_t> rs = {0, 1, -(int32_t)(cs.blinding_factors() + 1)};
    if (L) rs.push_back(-1);
    for (auto& q : cs.advice_q) rs.push_back(q.rotation);
    for (auto& q : cs.fixed_q) rs.push_back(q.rotation);
    std::sort(rs.begin(), rs.end());
    out[7] = (uint32_t)(std::unique(rs.begin(), rs.end()) - rs.begin());
    return 0;
} catch (...) { return DEHALO_ERR_OOM; }

extern "C" int dehalo_create_proof_circuit(dehalo_prover* p, const dehalo_circuit_inputs* in, dehalo_synthesis_info* info, const uint64_t* const* instances,
                                           const size_t* instance_lens, uint32_t num_instance_columns, dehalo_rng* rng, dehalo_transcript* transcript) try {
    if (!p || !transcript || !in) return DEHALO_ERR_INVALID;
    if (transcript->curve != p->pk->curve) return dh_fail(p->ctx, DEHALO_ERR_INVALID, "create_proof: the transcript's curve differs from the key's");
    std::lock_guard<std::mutex> lk(p->mu);
    const int rc = p->run(nullptr, instances, instance_lens, num_instance_columns, rng, transcript, 0, in, info);
    if (rc) {
        (void)hipStreamSynchronize(p->ctx->stream);
        if (p->side) (void)hipStreamSynchronize(p->side->stream);
    }
    return rc;
} catch (...) { return DEHALO_ERR_OOM; }

namespace {
// proof i on prover i mod num_provers, one library thread per prover; `one` makes proof i on prover p into `tr`
template <class One>
int proofs_on_threads(dehalo_prover* const* provers, uint32_t num_provers, uint32_t count, uint8_t* const* proofs_out, size_t proof_cap, size_t* proof_lens, One one) {
    for (uint32_t i = 0; i < num_provers; i++)
        if (!provers[i]) return DEHALO_ERR_INVALID;
    std::vector<int> rcs(num_provers, 0);
    auto work = [&](uint32_t t) {
        dehalo_prover* p = provers[t];
        for (uint32_t i = t; i < count && !rcs[t]; i += num_provers) {
            dehalo_transcript tr;
            tr.init(p->pk->curve);
            int rc = one(p, i, &tr);
            if (!rc && tr.proof.size() > proof_cap) rc = dh_fail(p->ctx, DEHALO_ERR_INVALID, "create_proofs: proof buffer too small");
            if (!rc) {
                memcpy(proofs_out[i], tr.proof.data(), tr.proof.size());
                proof_lens[i] = tr.proof.size();
            }
            rcs[t] = rc;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < num_provers; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    for (int rc : rcs)
        if (rc) return rc;
    return 0;
}
}   // namespace

extern "C" int dehalo_create_proofs(dehalo_prover* const* provers, uint32_t num_provers, const uint64_t* const* advice, uint32_t count, dehalo_rng* rngs, uint32_t flags,
                                    uint8_t* const* proofs_out, size_t proof_cap, size_t* proof_lens) try {
    if (!provers || !num_provers || (count && (!advice || !proofs_out || !proof_lens))) return DEHALO_ERR_INVALID;
    return proofs_on_threads(provers, num_provers, count, proofs_out, proof_cap, proof_lens, [&](dehalo_prover* p, uint32_t i, dehalo_transcript* tr) {
        return dehalo_create_proof(p, advice[i], nullptr, nullptr, p->I ? p->I : 0, rngs ? &rngs[i] : nullptr, tr, flags);
    });
} catch (...) { return DEHALO_ERR_OOM; }

// the same with every proof's circuit synthesized inside its call (dehalo_create_proof_circuit): inputs[i] -> proof i
extern "C" int dehalo_create_proofs_circuit(dehalo_prover* const* provers, uint32_t num_provers, const dehalo_circuit_inputs* inputs, uint32_t count, dehalo_rng* rngs,
                                            uint8_t* const* proofs_out, size_t proof_cap, size_t* proof_lens) try {
    if (!provers || !num_provers || (count && (!inputs || !proofs_out || !proof_lens))) return DEHALO_ERR_INVALID;
    return proofs_on_threads(provers, num_provers, count, proofs_out, proof_cap, proof_lens, [&](dehalo_prover* p, uint32_t i, dehalo_transcript* tr) {
        return dehalo_create_proof_circuit(p, &inputs[i], nullptr, nullptr, nullptr, p->I ? p->I : 0, rngs ? &rngs[i] : nullptr, tr);
    });
} catch (...) { return DEHALO_ERR_OOM; }
