/*
 * oracle.c -- CPU restatement of the reference's MSM / NTT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker
 * or as the reported CPU baseline ("kind": "port").  The product library
 * (libdehalo.so) never links, loads or calls it.
 *
 * PARITY UNPINNED at the MSM/NTT boundary: the reference holds no golden vector
 * for best_multiexp / best_fft and cannot be built here (no rustc/cargo; the
 * arithmetic lives in the un-vendored crates halo2_proofs @ tag v2023_04_20,
 * /root/reference/Cargo.toml:17, and halo2curves below it).  This file follows
 * the published algorithm of those crates as recorded in SURVEY.md Appendix A
 * (halo2_proofs/src/arithmetic.rs: best_multiexp, multiexp_serial, best_fft,
 * recursive_butterfly_arithmetic; halo2_proofs/src/poly/domain.rs:
 * lagrange_to_coeff, coeff_to_extended, extended_to_coeff) and Appendix B
 * (halo2curves 4 x u64 Montgomery fields, Jacobian a = 0 curves).  It is itself
 * pinned by oracle/pyoracle.py (Python big integers; constants re-derived from
 * the moduli; bn256::Fr arithmetic pinned by the reference's Poseidon KATs,
 * src/poseidon/permutation.rs:154-158,190-196) in tests/test_oracle.py.
 *
 * Reference call sites this stands in for: the create_proof calls at
 * benches/delay_enc.rs:123-131, benches/mod_pow.rs:201-209,
 * benches/pose_enc.rs:127-135 (which reach best_multiexp through
 * ParamsKZG::commit_lagrange and best_fft through EvaluationDomain).
 *
 * In-memory formats are halo2curves': field element = 4 little-endian u64 limbs
 * in Montgomery form (R = 2^256); affine point = {x, y} (64 B), identity = (0,0);
 * projective = Jacobian {x, y, z} (96 B), identity z = 0.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#include <math.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 v[4]; } fe;
typedef struct { fe x, y; } aff;
typedef struct { fe x, y, z; } jac;

typedef struct {
    u64 p[4];
    u64 inv;   /* -p^-1 mod 2^64 */
    fe r;      /* R mod p  (Montgomery one) */
    fe r2;     /* R^2 mod p */
} field_t;

/* ids shared with include/dehalo.h */
enum { F_BN254_FR = 0, F_BN254_FQ = 1, F_PASTA_FP = 2, F_PASTA_FQ = 3 };
enum { C_BN254 = 0, C_PALLAS = 1, C_VESTA = 2 };

static field_t FIELDS[4] = {
    { { 0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL }, 0, {{0}}, {{0}} },
    { { 0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL }, 0, {{0}}, {{0}} },
    { { 0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0000000000000000ULL, 0x4000000000000000ULL }, 0, {{0}}, {{0}} },
    { { 0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0000000000000000ULL, 0x4000000000000000ULL }, 0, {{0}}, {{0}} },
};
static int fields_ready = 0;

/* ---------- raw 256-bit helpers ---------- */
static inline int geq(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; }
    return 1;
}
static inline u64 add4(u64 o[4], const u64 a[4], const u64 b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; o[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static inline u64 sub4(u64 o[4], const u64 a[4], const u64 b[4]) {
    u64 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; o[i] = (u64)d; br = (u64)(d >> 64) & 1; }
    return br;
}

static inline void f_add(const field_t* F, fe* o, const fe* a, const fe* b) {
    u64 t[4]; u64 c = add4(t, a->v, b->v);
    if (c || geq(t, F->p)) sub4(t, t, F->p);
    memcpy(o->v, t, 32);
}
static inline void f_sub(const field_t* F, fe* o, const fe* a, const fe* b) {
    u64 t[4]; if (sub4(t, a->v, b->v)) add4(t, t, F->p);
    memcpy(o->v, t, 32);
}
static inline void f_neg(const field_t* F, fe* o, const fe* a) {
    u64 z = a->v[0] | a->v[1] | a->v[2] | a->v[3];
    if (!z) { memset(o, 0, 32); return; }
    u64 t[4]; sub4(t, F->p, a->v); memcpy(o->v, t, 32);
}
static inline void f_dbl(const field_t* F, fe* o, const fe* a) { f_add(F, o, a, a); }
static inline int f_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int f_eq(const fe* a, const fe* b) { return memcmp(a, b, 32) == 0; }

/* Montgomery product a*b*R^-1 mod p, CIOS over 64-bit limbs */
static inline void f_mul(const field_t* F, fe* o, const fe* a, const fe* b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        u64 m = t[0] * F->inv;
        c = (u128)m * F->p[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F->p[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq(t, F->p)) sub4(t, t, F->p);
    memcpy(o->v, t, 32);
}
static inline void f_sqr(const field_t* F, fe* o, const fe* a) { f_mul(F, o, a, a); }

static void f_from_mont(const field_t* F, u64 o[4], const fe* a) {
    fe one = {{1, 0, 0, 0}}, t; f_mul(F, &t, a, &one); memcpy(o, t.v, 32);
}
static void f_to_mont(const field_t* F, fe* o, const u64 a[4]) {
    fe t; memcpy(t.v, a, 32); f_mul(F, o, &t, &F->r2);
}
static void f_pow(const field_t* F, fe* o, const fe* a, const u64 e[4]) {
    fe acc = F->r, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) f_mul(F, &acc, &acc, &base);
        f_sqr(F, &base, &base);
    }
    *o = acc;
}
static void f_inv(const field_t* F, fe* o, const fe* a) {
    u64 e[4], two[4] = {2, 0, 0, 0}; sub4(e, F->p, two); f_pow(F, o, a, e);
}

static void init_fields(void) {
    if (fields_ready) return;
    for (int k = 0; k < 4; k++) {
        field_t* F = &FIELDS[k];
        u64 inv = 1; /* Newton: p^-1 mod 2^64, then negate */
        for (int i = 0; i < 63; i++) { inv *= inv; inv *= F->p[0]; }
        F->inv = (u64)0 - inv;
        /* R = 2^256 mod p by doubling 1 256 times; R2 by 256 more */
        u64 x[4] = {1, 0, 0, 0};
        for (int i = 0; i < 512; i++) {
            u64 c = add4(x, x, x);
            if (c || geq(x, F->p)) sub4(x, x, F->p);
            if (i == 255) memcpy(F->r.v, x, 32);
        }
        memcpy(F->r2.v, x, 32);
    }
    fields_ready = 1;
}

/* ---------- curve arithmetic (Jacobian, a = 0) ---------- */
typedef struct { const field_t* F; fe b; aff g; } curve_t;
static curve_t CURVES[3];
static int curves_ready = 0;

static void small_mont(const field_t* F, fe* o, u64 k) { u64 t[4] = {k, 0, 0, 0}; f_to_mont(F, o, t); }

static void init_curves(void) {
    if (curves_ready) return;
    init_fields();
    fe one, two;
    CURVES[C_BN254].F = &FIELDS[F_BN254_FQ];
    small_mont(CURVES[C_BN254].F, &CURVES[C_BN254].b, 3);
    small_mont(CURVES[C_BN254].F, &CURVES[C_BN254].g.x, 1);
    small_mont(CURVES[C_BN254].F, &CURVES[C_BN254].g.y, 2);
    CURVES[C_PALLAS].F = &FIELDS[F_PASTA_FP];
    CURVES[C_VESTA].F = &FIELDS[F_PASTA_FQ];
    for (int c = C_PALLAS; c <= C_VESTA; c++) {
        const field_t* F = CURVES[c].F;
        small_mont(F, &CURVES[c].b, 5);
        small_mont(F, &one, 1); small_mont(F, &two, 2);
        f_neg(F, &CURVES[c].g.x, &one);
        CURVES[c].g.y = two;
    }
    curves_ready = 1;
}

static inline int aff_is_id(const aff* p) { return f_is_zero(&p->x) && f_is_zero(&p->y); }
static inline void jac_set_id(jac* r) { memset(r, 0, sizeof(*r)); }
static inline int jac_is_id(const jac* p) { return f_is_zero(&p->z); }

static void jac_double(const curve_t* C, jac* r, const jac* p) {
    const field_t* F = C->F;
    if (jac_is_id(p)) { jac_set_id(r); return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    f_sqr(F, &a, &p->x); f_sqr(F, &b, &p->y); f_sqr(F, &c, &b);
    f_add(F, &t, &p->x, &b); f_sqr(F, &t, &t); f_sub(F, &t, &t, &a); f_sub(F, &t, &t, &c); f_dbl(F, &d, &t);
    f_dbl(F, &e, &a); f_add(F, &e, &e, &a);
    f_sqr(F, &f, &e);
    f_mul(F, &z3, &p->y, &p->z); f_dbl(F, &z3, &z3);
    f_dbl(F, &t, &d); f_sub(F, &x3, &f, &t);
    f_sub(F, &t, &d, &x3); f_mul(F, &y3, &e, &t);
    f_dbl(F, &c, &c); f_dbl(F, &c, &c); f_dbl(F, &c, &c); f_sub(F, &y3, &y3, &c);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_add(const curve_t* C, jac* r, const jac* p, const jac* q) {
    const field_t* F = C->F;
    if (jac_is_id(p)) { *r = *q; return; }
    if (jac_is_id(q)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t, x3, y3, z3;
    f_sqr(F, &z1z1, &p->z); f_sqr(F, &z2z2, &q->z);
    f_mul(F, &u1, &p->x, &z2z2); f_mul(F, &u2, &q->x, &z1z1);
    f_mul(F, &s1, &p->y, &q->z); f_mul(F, &s1, &s1, &z2z2);
    f_mul(F, &s2, &q->y, &p->z); f_mul(F, &s2, &s2, &z1z1);
    if (f_eq(&u1, &u2)) {
        if (f_eq(&s1, &s2)) { jac_double(C, r, p); return; }
        jac_set_id(r); return;
    }
    f_sub(F, &h, &u2, &u1); f_sub(F, &rr, &s2, &s1);
    f_sqr(F, &hh, &h); f_mul(F, &hhh, &h, &hh); f_mul(F, &v, &u1, &hh);
    f_sqr(F, &x3, &rr); f_sub(F, &x3, &x3, &hhh); f_dbl(F, &t, &v); f_sub(F, &x3, &x3, &t);
    f_sub(F, &t, &v, &x3); f_mul(F, &y3, &rr, &t); f_mul(F, &t, &s1, &hhh); f_sub(F, &y3, &y3, &t);
    f_mul(F, &z3, &p->z, &q->z); f_mul(F, &z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_add_mixed(const curve_t* C, jac* r, const jac* p, const aff* q) {
    const field_t* F = C->F;
    if (aff_is_id(q)) { *r = *p; return; }
    if (jac_is_id(p)) { r->x = q->x; r->y = q->y; r->z = F->r; return; }
    fe z1z1, u2, s2, h, rr, hh, hhh, v, t, x3, y3, z3;
    f_sqr(F, &z1z1, &p->z);
    f_mul(F, &u2, &q->x, &z1z1);
    f_mul(F, &s2, &q->y, &p->z); f_mul(F, &s2, &s2, &z1z1);
    if (f_eq(&p->x, &u2)) {
        if (f_eq(&p->y, &s2)) { jac_double(C, r, p); return; }
        jac_set_id(r); return;
    }
    f_sub(F, &h, &u2, &p->x); f_sub(F, &rr, &s2, &p->y);
    f_sqr(F, &hh, &h); f_mul(F, &hhh, &h, &hh); f_mul(F, &v, &p->x, &hh);
    f_sqr(F, &x3, &rr); f_sub(F, &x3, &x3, &hhh); f_dbl(F, &t, &v); f_sub(F, &x3, &x3, &t);
    f_sub(F, &t, &v, &x3); f_mul(F, &y3, &rr, &t); f_mul(F, &t, &p->y, &hhh); f_sub(F, &y3, &y3, &t);
    f_mul(F, &z3, &p->z, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_to_affine(const curve_t* C, aff* r, const jac* p) {
    const field_t* F = C->F;
    if (jac_is_id(p)) { memset(r, 0, sizeof(*r)); return; }
    fe zi, zi2, zi3;
    f_inv(F, &zi, &p->z); f_sqr(F, &zi2, &zi); f_mul(F, &zi3, &zi2, &zi);
    f_mul(F, &r->x, &p->x, &zi2); f_mul(F, &r->y, &p->y, &zi3);
}

/* ---------- multiexp_serial / best_multiexp ---------- */
static inline u64 get_at(int segment, int c, const unsigned char bytes[32]) {
    int skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    unsigned char v[8] = {0};
    int len = 32 - skip_bytes; if (len > 8) len = 8;
    memcpy(v, bytes + skip_bytes, len);
    u64 tmp; memcpy(&tmp, v, 8);
    tmp >>= skip_bits - skip_bytes * 8;
    return tmp % ((u64)1 << c);
}

typedef struct { int state; /* 0 None, 1 Affine, 2 Projective */ aff a; jac j; } bucket_t;

static void multiexp_serial(const curve_t* C, const field_t* FS, const fe* coeffs, const aff* bases, size_t n, jac* acc) {
    unsigned char (*bytes)[32] = malloc(n * 32 + 32);
    for (size_t i = 0; i < n; i++) { u64 t[4]; f_from_mont(FS, t, &coeffs[i]); memcpy(bytes[i], t, 32); }
    int c;
    if (n < 4) c = 1; else if (n < 32) c = 3; else c = (int)ceil(log((double)n));
    int segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    bucket_t* buckets = malloc(nb * sizeof(bucket_t));
    for (int seg = segments - 1; seg >= 0; seg--) {
        for (int k = 0; k < c; k++) jac_double(C, acc, acc);
        for (size_t b = 0; b < nb; b++) buckets[b].state = 0;
        for (size_t i = 0; i < n; i++) {
            u64 d = get_at(seg, c, bytes[i]);
            if (!d) continue;
            bucket_t* B = &buckets[d - 1];
            if (B->state == 0) { B->a = bases[i]; B->state = 1; }
            else if (B->state == 1) {
                jac t; if (aff_is_id(&B->a)) jac_set_id(&t); else { t.x = B->a.x; t.y = B->a.y; t.z = C->F->r; }
                jac_add_mixed(C, &B->j, &t, &bases[i]); B->state = 2;
            } else jac_add_mixed(C, &B->j, &B->j, &bases[i]);
        }
        jac running; jac_set_id(&running);
        for (size_t b = nb; b-- > 0;) {
            bucket_t* B = &buckets[b];
            if (B->state == 1) jac_add_mixed(C, &running, &running, &B->a);
            else if (B->state == 2) jac_add(C, &running, &running, &B->j);
            jac_add(C, acc, acc, &running);
        }
    }
    free(buckets); free(bytes);
}

typedef struct { const curve_t* C; const field_t* FS; const fe* coeffs; const aff* bases; size_t n; jac res; } msm_job;
static void* msm_worker(void* arg) { msm_job* j = arg; jac_set_id(&j->res); multiexp_serial(j->C, j->FS, j->coeffs, j->bases, j->n, &j->res); return NULL; }

static const field_t* scalar_field_of(int curve) {
    return curve == C_BN254 ? &FIELDS[F_BN254_FR] : curve == C_PALLAS ? &FIELDS[F_PASTA_FQ] : &FIELDS[F_PASTA_FP];
}

/* best_multiexp: chunk = n / threads (last chunk may be short; may give threads+1 chunks) */
int orc_best_multiexp(int curve, const u64* coeffs, const u64* bases, size_t n, int threads, u64 out_jac[12]) {
    if (curve < 0 || curve > 2 || threads < 1) return -1;
    init_curves();
    const curve_t* C = &CURVES[curve];
    const field_t* FS = scalar_field_of(curve);
    jac acc; jac_set_id(&acc);
    if (n > (size_t)threads) {
        size_t chunk = n / threads, nch = (n + chunk - 1) / chunk;
        msm_job* jobs = calloc(nch, sizeof(msm_job));
        pthread_t* th = calloc(nch, sizeof(pthread_t));
        for (size_t k = 0; k < nch; k++) {
            size_t off = k * chunk, len = (off + chunk <= n) ? chunk : n - off;
            jobs[k] = (msm_job){C, FS, (const fe*)coeffs + off, (const aff*)bases + off, len, {{{0}}}};
            pthread_create(&th[k], NULL, msm_worker, &jobs[k]);
        }
        for (size_t k = 0; k < nch; k++) { pthread_join(th[k], NULL); jac_add(C, &acc, &acc, &jobs[k].res); }
        free(jobs); free(th);
    } else {
        multiexp_serial(C, FS, (const fe*)coeffs, (const aff*)bases, n, &acc);
    }
    memcpy(out_jac, &acc, 96);
    return 0;
}

int orc_to_affine(int curve, const u64 jac_in[12], u64 out_xy[8]) {
    if (curve < 0 || curve > 2) return -1;
    init_curves();
    jac p; memcpy(&p, jac_in, 96); aff a; jac_to_affine(&CURVES[curve], &a, &p); memcpy(out_xy, &a, 64);
    return 0;
}

/* ---------- best_fft ---------- */
static inline uint32_t bitrev(uint32_t k, int l) { uint32_t r = 0; for (int i = 0; i < l; i++) { r = (r << 1) | (k & 1); k >>= 1; } return r; }

static void butterfly_rec(const field_t* F, fe* a, size_t n, size_t stride, const fe* tw) {
    if (n == 2) {
        fe t = a[1]; f_sub(F, &a[1], &a[0], &t); f_add(F, &a[0], &a[0], &t);
        return;
    }
    butterfly_rec(F, a, n / 2, stride * 2, tw);
    butterfly_rec(F, a + n / 2, n / 2, stride * 2, tw);
    fe* l = a; fe* r = a + n / 2;
    { fe t = r[0]; f_sub(F, &r[0], &l[0], &t); f_add(F, &l[0], &l[0], &t); }
    for (size_t i = 1; i < n / 2; i++) {
        fe t; f_mul(F, &t, &r[i], &tw[i * stride]);
        f_sub(F, &r[i], &l[i], &t); f_add(F, &l[i], &l[i], &t);
    }
}

typedef struct { const field_t* F; fe* a; size_t n; size_t stride; const fe* tw; size_t lo, hi; } fft_job;
static void* fft_leaf_worker(void* arg) { fft_job* j = arg; butterfly_rec(j->F, j->a, j->n, j->stride, j->tw); return NULL; }
static void* fft_comb_worker(void* arg) {
    fft_job* j = arg; const field_t* F = j->F;
    fe* l = j->a; fe* r = j->a + j->n / 2;
    for (size_t i = j->lo; i < j->hi; i++) {
        fe t;
        if (i == 0) t = r[0]; else f_mul(F, &t, &r[i], &j->tw[i * j->stride]);
        f_sub(F, &r[i], &l[i], &t); f_add(F, &l[i], &l[i], &t);
    }
    return NULL;
}

int orc_best_fft(int field, u64* a_, const u64 omega_[4], uint32_t log_n, int threads) {
    if (field < 0 || field > 3 || threads < 1 || log_n > 30) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    fe* a = (fe*)a_;
    size_t n = (size_t)1 << log_n;
    if (log_n == 0) return 0;
    for (size_t k = 0; k < n; k++) { size_t rk = bitrev((uint32_t)k, log_n); if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; } }
    fe omega; memcpy(&omega, omega_, 32);
    fe* tw = malloc((n / 2 + 1) * sizeof(fe));
    tw[0] = F->r;
    for (size_t i = 1; i < n / 2; i++) f_mul(F, &tw[i], &tw[i - 1], &omega);
    int lt = 0; while ((2 << lt) <= threads) lt++;        /* floor(log2(threads)) */
    if ((int)log_n <= lt + 1 || threads == 1) {
        butterfly_rec(F, a, n, 1, tw);
    } else {
        /* rayon::join recursion flattened: 2^lt leaves run in parallel, then the
         * lt combine levels run with the butterfly loop split across threads. */
        size_t leaves = (size_t)1 << lt, ln = n >> lt;
        pthread_t* th = calloc(leaves > (size_t)threads ? leaves : (size_t)threads, sizeof(pthread_t));
        fft_job* jobs = calloc(leaves > (size_t)threads ? leaves : (size_t)threads, sizeof(fft_job));
        for (size_t k = 0; k < leaves; k++) {
            jobs[k] = (fft_job){F, a + k * ln, ln, leaves, tw, 0, 0};
            pthread_create(&th[k], NULL, fft_leaf_worker, &jobs[k]);
        }
        for (size_t k = 0; k < leaves; k++) pthread_join(th[k], NULL);
        for (int lvl = lt - 1; lvl >= 0; lvl--) {
            size_t blocks = (size_t)1 << lvl, bn = n >> lvl, per = leaves / blocks; /* threads per block */
            size_t idx = 0;
            for (size_t b = 0; b < blocks; b++) {
                size_t half = bn / 2, step = (half + per - 1) / per;
                for (size_t q = 0; q < per; q++, idx++) {
                    size_t lo = q * step, hi = lo + step > half ? half : lo + step;
                    jobs[idx] = (fft_job){F, a + b * bn, bn, blocks, tw, lo, hi};
                    pthread_create(&th[idx], NULL, fft_comb_worker, &jobs[idx]);
                }
            }
            for (size_t k = 0; k < idx; k++) pthread_join(th[k], NULL);
        }
        free(th); free(jobs);
    }
    free(tw);
    return 0;
}

/* ---------- EvaluationDomain wrappers ---------- */
static void scale_pattern(const field_t* F, fe* a, size_t n, const fe* all, const fe pw[2]) {
    for (size_t i = 0; i < n; i++) {
        if (all) f_mul(F, &a[i], &a[i], all);
        if (pw) { size_t r = i % 3; if (r) f_mul(F, &a[i], &a[i], &pw[r - 1]); }
    }
}
/* lagrange_to_coeff: best_fft(omega_inv) then * ifft_divisor */
int orc_lagrange_to_coeff(int field, u64* a, uint32_t k, const u64 omega_inv[4], const u64 divisor[4], int threads) {
    int rc = orc_best_fft(field, a, omega_inv, k, threads); if (rc) return rc;
    fe d; memcpy(&d, divisor, 32);
    scale_pattern(&FIELDS[field], (fe*)a, (size_t)1 << k, &d, NULL);
    return 0;
}
/* coeff_to_extended: a[i] *= [1, zeta, zeta^2][i % 3]; zero-pad to 2^ext_k; best_fft(ext_omega) */
int orc_coeff_to_extended(int field, const u64* coeffs, uint32_t k, u64* ext, uint32_t ext_k, const u64 ext_omega[4], const u64 zeta[4], int threads) {
    if (field < 0 || field > 3 || ext_k < k) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    size_t n = (size_t)1 << k, en = (size_t)1 << ext_k;
    memcpy(ext, coeffs, n * 32); memset(ext + 4 * n, 0, (en - n) * 32);
    fe pw[2]; memcpy(&pw[0], zeta, 32); f_sqr(F, &pw[1], &pw[0]);
    scale_pattern(F, (fe*)ext, n, NULL, pw);
    return orc_best_fft(field, ext, ext_omega, ext_k, threads);
}
/* extended_to_coeff: best_fft(ext_omega_inv); * ext_ifft_divisor; a[i] *= [1, zeta^2, zeta][i % 3]
 * (caller truncates to n * quotient_poly_degree) */
int orc_extended_to_coeff(int field, u64* a, uint32_t ext_k, const u64 ext_omega_inv[4], const u64 divisor[4], const u64 zeta[4], int threads) {
    int rc = orc_best_fft(field, a, ext_omega_inv, ext_k, threads); if (rc) return rc;
    const field_t* F = &FIELDS[field];
    fe d; memcpy(&d, divisor, 32);
    fe z, pw[2]; memcpy(&z, zeta, 32); f_sqr(F, &pw[0], &z); pw[1] = z;
    scale_pattern(F, (fe*)a, (size_t)1 << ext_k, &d, pw);
    return 0;
}

/* ---------- field-vector primitives around the path (Montgomery in/out) ----------
 * eval_polynomial: [UPSTREAM halo2_proofs/src/arithmetic.rs] serial Horner below 2^? terms, else one
 * Horner per thread chunk, scaled by point^(chunk start) and summed. */
typedef struct { const field_t* F; const fe* c; size_t n; fe point; fe res; } evp_job;
static void* evp_worker(void* arg) {
    evp_job* j = arg;
    fe acc; memset(&acc, 0, sizeof(acc));
    for (size_t i = j->n; i-- > 0;) { f_mul(j->F, &acc, &acc, &j->point); f_add(j->F, &acc, &acc, &j->c[i]); }
    j->res = acc;
    return NULL;
}
int orc_eval_polynomial(int field, const u64* coeffs, size_t n, const u64 point[4], int threads, u64 out[4]) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (n < (size_t)threads || threads == 1) {
        evp_job j = {F, (const fe*)coeffs, n, *(const fe*)point, {{0}}};
        evp_worker(&j);
        memcpy(out, &j.res, 32);
        return 0;
    }
    size_t chunk = (n + threads - 1) / threads;
    evp_job jobs[256]; pthread_t th[256];
    int used = 0;
    for (size_t st = 0; st < n; st += chunk, used++) {
        jobs[used].F = F; jobs[used].c = (const fe*)coeffs + st; jobs[used].n = st + chunk <= n ? chunk : n - st;
        jobs[used].point = *(const fe*)point;
        pthread_create(&th[used], NULL, evp_worker, &jobs[used]);
    }
    fe step = F->r, sum, pw = F->r;               /* point^chunk, running power */
    { u64 e[4] = {chunk, 0, 0, 0}; f_pow(F, &step, (const fe*)point, e); }
    memset(&sum, 0, sizeof(sum));
    for (int i = 0; i < used; i++) {
        pthread_join(th[i], NULL);
        fe t; f_mul(F, &t, &jobs[i].res, &pw); f_add(F, &sum, &sum, &t);
        f_mul(F, &pw, &pw, &step);
    }
    memcpy(out, &sum, 32);
    return 0;
}
/* batch_invert: [UPSTREAM ff::BatchInvert] one inversion, zeros skipped.  scratch = n elements. */
int orc_batch_invert(int field, u64* v_, size_t n) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    fe* v = (fe*)v_;
    fe* prefix = malloc((n ? n : 1) * sizeof(fe));
    if (!prefix) return -3;
    fe acc = F->r;
    for (size_t i = 0; i < n; i++) { prefix[i] = acc; if (!f_is_zero(&v[i])) f_mul(F, &acc, &acc, &v[i]); }
    fe inv; f_inv(F, &inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (f_is_zero(&v[i])) continue;
        fe t = v[i];
        f_mul(F, &v[i], &inv, &prefix[i]);
        f_mul(F, &inv, &inv, &t);
    }
    free(prefix);
    return 0;
}
/* grand product: [UPSTREAM plonk/permutation/prover.rs, plonk/lookup/prover.rs] z[0] = 1, z[i] = z[i-1] num[i-1] / den[i-1] */
int orc_grand_product(int field, const u64* num_, const u64* den_, size_t n, u64* z_) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    fe* dinv = malloc((n ? n : 1) * sizeof(fe));
    if (!dinv) return -3;
    memcpy(dinv, den_, n * sizeof(fe));
    orc_batch_invert(field, (u64*)dinv, n);
    const fe* num = (const fe*)num_; fe* z = (fe*)z_;
    fe acc = F->r;
    for (size_t i = 0; i < n; i++) { z[i] = acc; fe t; f_mul(F, &t, &num[i], &dinv[i]); f_mul(F, &acc, &acc, &t); }
    free(dinv);
    return 0;
}

/* permute_expression_pair: [UPSTREAM halo2_proofs/src/plonk/lookup/prover.rs] sort the input, give every first
 * occurrence its own value from the table multiset, pour the leftover table values (ascending) into the
 * repeated rows from the end.  Returns 1 where upstream returns Err(ConstraintSystemFailure). */
typedef struct { u64 c[4]; fe m; } lp_item;     /* canonical key + the original Montgomery element */
static int lp_cmp(const void* a_, const void* b_) {
    const lp_item* a = a_; const lp_item* b = b_;
    for (int i = 3; i >= 0; i--) if (a->c[i] != b->c[i]) return a->c[i] < b->c[i] ? -1 : 1;
    return 0;
}
int orc_permute_expression_pair(int field, const u64* input_, const u64* table_, size_t usable, u64* out_input_, u64* out_table_) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    lp_item* A = malloc((usable ? usable : 1) * sizeof(lp_item)); lp_item* T = malloc((usable ? usable : 1) * sizeof(lp_item));
    size_t* repeated = malloc((usable ? usable : 1) * sizeof(size_t)); unsigned char* consumed = calloc(usable ? usable : 1, 1);
    for (size_t i = 0; i < usable; i++) {
        A[i].m = ((const fe*)input_)[i]; f_from_mont(F, A[i].c, &A[i].m);
        T[i].m = ((const fe*)table_)[i]; f_from_mont(F, T[i].c, &T[i].m);
    }
    qsort(A, usable, sizeof(lp_item), lp_cmp); qsort(T, usable, sizeof(lp_item), lp_cmp);
    fe* oi = (fe*)out_input_; fe* ot = (fe*)out_table_;
    size_t nrep = 0, tpos = 0;
    int fail = 0;
    for (size_t row = 0; row < usable; row++) {
        oi[row] = A[row].m;
        if (row == 0 || lp_cmp(&A[row], &A[row - 1]) != 0) {
            ot[row] = A[row].m;
            while (tpos < usable && lp_cmp(&T[tpos], &A[row]) < 0) tpos++;      /* first copy of the value in the sorted table */
            if (tpos < usable && lp_cmp(&T[tpos], &A[row]) == 0) consumed[tpos++] = 1; else { fail = 1; break; }
        } else repeated[nrep++] = row;
    }
    if (!fail)
        for (size_t j = 0; j < usable; j++)
            if (!consumed[j]) ot[repeated[--nrep]] = T[j].m;
    free(A); free(T); free(repeated); free(consumed);
    return fail;
}

/* ---------- quotient numerator: [UPSTREAM halo2_proofs/src/plonk/evaluation.rs @ v2023_04_20] ----------
 * GraphEvaluator::evaluate per row, and the permutation / lookup terms of Evaluator::evaluate_h, with upstream's
 * `parallelize` (contiguous row chunks, one per thread).  Montgomery in / out. */
typedef struct { uint32_t kind, index, rotation; } orc_source;
typedef struct { uint32_t op; orc_source a, b; uint32_t parts_begin, parts_len, target; } orc_calc;
typedef struct {
    const field_t* F;
    const fe* constants; const int32_t* rotations; uint32_t nrot; const orc_calc* calcs; uint32_t ncalcs; const orc_source* parts; uint32_t nint;
    const fe* const* fixed; const fe* const* advice; const fe* const* instance; const fe* challenges; fe beta, gamma, theta, y;
    uint32_t rot_scale; size_t rows; const fe* previous; fe* out;
    size_t start, end;
} geval_job;
static inline size_t rot_idx(size_t idx, int32_t rot, uint32_t rot_scale, size_t isize) {
    int64_t v = ((int64_t)idx + (int64_t)rot * (int64_t)rot_scale) % (int64_t)isize;
    return (size_t)(v < 0 ? v + (int64_t)isize : v);
}
static void* geval_worker(void* arg) {
    geval_job* j = arg;
    const field_t* F = j->F;
    fe* inter = malloc((j->nint ? j->nint : 1) * sizeof(fe));
    size_t* rots = malloc((j->nrot ? j->nrot : 1) * sizeof(size_t));
    fe zero; memset(&zero, 0, sizeof(zero));
    for (size_t idx = j->start; idx < j->end; idx++) {
        for (uint32_t r = 0; r < j->nrot; r++) rots[r] = rot_idx(idx, j->rotations[r], j->rot_scale, j->rows);
        const fe prev = j->previous ? j->previous[idx] : zero;
#define GET(S) ((S).kind == 0 ? j->constants[(S).index] : (S).kind == 1 ? inter[(S).index] : (S).kind == 2 ? j->fixed[(S).index][rots[(S).rotation]] : \
                (S).kind == 3 ? j->advice[(S).index][rots[(S).rotation]] : (S).kind == 4 ? j->instance[(S).index][rots[(S).rotation]] :                  \
                (S).kind == 5 ? j->challenges[(S).index] : (S).kind == 6 ? j->beta : (S).kind == 7 ? j->gamma : (S).kind == 8 ? j->theta :                \
                (S).kind == 9 ? j->y : prev)
        for (uint32_t c = 0; c < j->ncalcs; c++) {
            const orc_calc* k = &j->calcs[c];
            fe a = GET(k->a), b, v;
            switch (k->op) {
                case 0: b = GET(k->b); f_add(F, &v, &a, &b); break;
                case 1: b = GET(k->b); f_sub(F, &v, &a, &b); break;
                case 2: b = GET(k->b); f_mul(F, &v, &a, &b); break;
                case 3: f_sqr(F, &v, &a); break;
                case 4: f_dbl(F, &v, &a); break;
                case 5: f_neg(F, &v, &a); break;
                case 6:
                    b = GET(k->b); v = a;
                    for (uint32_t q = 0; q < k->parts_len; q++) { fe part = GET(j->parts[k->parts_begin + q]); f_mul(F, &v, &v, &b); f_add(F, &v, &v, &part); }
                    break;
                default: v = a; break;
            }
            inter[k->target] = v;
        }
#undef GET
        j->out[idx] = j->ncalcs ? inter[j->calcs[j->ncalcs - 1].target] : zero;
    }
    free(inter); free(rots);
    return NULL;
}
static void run_chunks(void* (*fn)(void*), void* jobs, size_t job_size, size_t rows, int threads, size_t start_off, size_t end_off) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((size_t)threads > rows) threads = (int)(rows ? rows : 1);
    pthread_t th[256];
    size_t chunk = (rows + threads - 1) / threads;
    char* base = jobs;
    for (int t = 0; t < threads; t++) {
        char* jt = base + (size_t)t * job_size;
        if (t) memcpy(jt, base, job_size);
        size_t st = (size_t)t * chunk, en = st + chunk < rows ? st + chunk : rows;
        if (st > rows) st = rows;
        *(size_t*)(jt + start_off) = st; *(size_t*)(jt + end_off) = en;
    }
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, fn, base + (size_t)t * job_size);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}
int orc_graph_evaluate(int field, const u64* constants, const int32_t* rotations, uint32_t nrot, const orc_calc* calcs, uint32_t ncalcs, const orc_source* parts,
                       uint32_t nint, const u64* const* fixed, const u64* const* advice, const u64* const* instance, const u64* challenges,
                       const u64* bgty /* beta, gamma, theta, y: 16 u64 */, uint32_t log_rows, uint32_t rot_scale, const u64* previous, u64* out, int threads) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    geval_job* jobs = calloc(256, sizeof(geval_job));
    geval_job* j = &jobs[0];
    j->F = &FIELDS[field]; j->constants = (const fe*)constants; j->rotations = rotations; j->nrot = nrot; j->calcs = calcs; j->ncalcs = ncalcs; j->parts = parts;
    j->nint = nint; j->fixed = (const fe* const*)fixed; j->advice = (const fe* const*)advice; j->instance = (const fe* const*)instance;
    j->challenges = (const fe*)challenges;
    memcpy(&j->beta, bgty, 32); memcpy(&j->gamma, bgty + 4, 32); memcpy(&j->theta, bgty + 8, 32); memcpy(&j->y, bgty + 12, 32);
    j->rot_scale = rot_scale; j->rows = (size_t)1 << log_rows; j->previous = (const fe*)previous; j->out = (fe*)out;
    run_chunks(geval_worker, jobs, sizeof(geval_job), j->rows, threads, offsetof(geval_job, start), offsetof(geval_job, end));
    free(jobs);
    return 0;
}

typedef struct {
    const field_t* F; const fe* const* z; uint32_t nsets; const fe* const* cols; const fe* const* sigma; uint32_t ncols, chunk_len; int32_t last_rotation;
    const fe *l0, *l_last, *l_active; fe beta, gamma, y, delta, beta_zeta, omega; uint32_t rot_scale; size_t rows; fe* values;
    size_t start, end;
} perm_job;
static void* perm_worker(void* arg) {
    perm_job* j = arg;
    const field_t* F = j->F;
    if (!j->nsets) return NULL;
    fe beta_term; { u64 e[4] = {j->start, 0, 0, 0}; f_pow(F, &beta_term, &j->omega, e); }
    const fe one = F->r;
    for (size_t idx = j->start; idx < j->end; idx++) {
        fe v = j->values[idx], t, u;
        size_t r_next = rot_idx(idx, 1, j->rot_scale, j->rows), r_last = rot_idx(idx, j->last_rotation, j->rot_scale, j->rows);
        f_mul(F, &v, &v, &j->y); f_sub(F, &t, &one, &j->z[0][idx]); f_mul(F, &t, &t, &j->l0[idx]); f_add(F, &v, &v, &t);
        const fe* zl = &j->z[j->nsets - 1][idx];
        f_mul(F, &v, &v, &j->y); f_sqr(F, &t, zl); f_sub(F, &t, &t, zl); f_mul(F, &t, &t, &j->l_last[idx]); f_add(F, &v, &v, &t);
        for (uint32_t s = 1; s < j->nsets; s++) {
            f_mul(F, &v, &v, &j->y); f_sub(F, &t, &j->z[s][idx], &j->z[s - 1][r_last]); f_mul(F, &t, &t, &j->l0[idx]); f_add(F, &v, &v, &t);
        }
        fe cur; f_mul(F, &cur, &j->beta_zeta, &beta_term);
        for (uint32_t s = 0; s < j->nsets; s++) {
            uint32_t c0 = s * j->chunk_len, c1 = c0 + j->chunk_len < j->ncols ? c0 + j->chunk_len : j->ncols;
            fe left = j->z[s][r_next], right = j->z[s][idx];
            for (uint32_t c = c0; c < c1; c++) {
                f_mul(F, &t, &j->beta, &j->sigma[c][idx]); f_add(F, &t, &t, &j->cols[c][idx]); f_add(F, &t, &t, &j->gamma); f_mul(F, &left, &left, &t);
            }
            for (uint32_t c = c0; c < c1; c++) {
                f_add(F, &u, &j->cols[c][idx], &cur); f_add(F, &u, &u, &j->gamma); f_mul(F, &right, &right, &u);
                f_mul(F, &cur, &cur, &j->delta);
            }
            f_mul(F, &v, &v, &j->y); f_sub(F, &t, &left, &right); f_mul(F, &t, &t, &j->l_active[idx]); f_add(F, &v, &v, &t);
        }
        f_mul(F, &beta_term, &beta_term, &j->omega);
        j->values[idx] = v;
    }
    return NULL;
}
int orc_permutation_h(int field, const u64* const* z, uint32_t nsets, const u64* const* cols, const u64* const* sigma, uint32_t ncols, uint32_t chunk_len,
                      int32_t last_rotation, const u64* l0, const u64* l_last, const u64* l_active, const u64* scalars /* beta gamma y delta beta_zeta omega */,
                      uint32_t log_rows, uint32_t rot_scale, u64* values, int threads) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    perm_job* jobs = calloc(256, sizeof(perm_job));
    perm_job* j = &jobs[0];
    j->F = &FIELDS[field]; j->z = (const fe* const*)z; j->nsets = nsets; j->cols = (const fe* const*)cols; j->sigma = (const fe* const*)sigma; j->ncols = ncols;
    j->chunk_len = chunk_len; j->last_rotation = last_rotation; j->l0 = (const fe*)l0; j->l_last = (const fe*)l_last; j->l_active = (const fe*)l_active;
    memcpy(&j->beta, scalars, 32); memcpy(&j->gamma, scalars + 4, 32); memcpy(&j->y, scalars + 8, 32); memcpy(&j->delta, scalars + 12, 32);
    memcpy(&j->beta_zeta, scalars + 16, 32); memcpy(&j->omega, scalars + 20, 32);
    j->rot_scale = rot_scale; j->rows = (size_t)1 << log_rows; j->values = (fe*)values;
    run_chunks(perm_worker, jobs, sizeof(perm_job), j->rows, threads, offsetof(perm_job, start), offsetof(perm_job, end));
    free(jobs);
    return 0;
}

typedef struct {
    const field_t* F; const fe *z, *a, *s, *tv, *l0, *l_last, *l_active; fe beta, gamma, y; uint32_t rot_scale; size_t rows; fe* values;
    size_t start, end;
} lookup_job;
static void* lookup_worker(void* arg) {
    lookup_job* j = arg;
    const field_t* F = j->F;
    const fe one = F->r;
    for (size_t idx = j->start; idx < j->end; idx++) {
        fe v = j->values[idx], t, u, ams;
        size_t r_next = rot_idx(idx, 1, j->rot_scale, j->rows), r_prev = rot_idx(idx, -1, j->rot_scale, j->rows);
        f_sub(F, &ams, &j->a[idx], &j->s[idx]);
        f_mul(F, &v, &v, &j->y); f_sub(F, &t, &one, &j->z[idx]); f_mul(F, &t, &t, &j->l0[idx]); f_add(F, &v, &v, &t);
        f_mul(F, &v, &v, &j->y); f_sqr(F, &t, &j->z[idx]); f_sub(F, &t, &t, &j->z[idx]); f_mul(F, &t, &t, &j->l_last[idx]); f_add(F, &v, &v, &t);
        f_add(F, &t, &j->a[idx], &j->beta); f_mul(F, &t, &t, &j->z[r_next]); f_add(F, &u, &j->s[idx], &j->gamma); f_mul(F, &t, &t, &u);
        f_mul(F, &u, &j->z[idx], &j->tv[idx]); f_sub(F, &t, &t, &u); f_mul(F, &t, &t, &j->l_active[idx]);
        f_mul(F, &v, &v, &j->y); f_add(F, &v, &v, &t);
        f_mul(F, &v, &v, &j->y); f_mul(F, &t, &ams, &j->l0[idx]); f_add(F, &v, &v, &t);
        f_sub(F, &t, &j->a[idx], &j->a[r_prev]); f_mul(F, &t, &t, &ams); f_mul(F, &t, &t, &j->l_active[idx]);
        f_mul(F, &v, &v, &j->y); f_add(F, &v, &v, &t);
        j->values[idx] = v;
    }
    return NULL;
}
int orc_lookup_h(int field, const u64* z, const u64* a, const u64* s_, const u64* tv, const u64* l0, const u64* l_last, const u64* l_active,
                 const u64* scalars /* beta gamma y */, uint32_t log_rows, uint32_t rot_scale, u64* values, int threads) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    lookup_job* jobs = calloc(256, sizeof(lookup_job));
    lookup_job* j = &jobs[0];
    j->F = &FIELDS[field]; j->z = (const fe*)z; j->a = (const fe*)a; j->s = (const fe*)s_; j->tv = (const fe*)tv; j->l0 = (const fe*)l0; j->l_last = (const fe*)l_last;
    j->l_active = (const fe*)l_active;
    memcpy(&j->beta, scalars, 32); memcpy(&j->gamma, scalars + 4, 32); memcpy(&j->y, scalars + 8, 32);
    j->rot_scale = rot_scale; j->rows = (size_t)1 << log_rows; j->values = (fe*)values;
    run_chunks(lookup_worker, jobs, sizeof(lookup_job), j->rows, threads, offsetof(lookup_job, start), offsetof(lookup_job, end));
    free(jobs);
    return 0;
}

/* ---------- field ops exposed for the KAT pin (Montgomery in/out) ---------- */
int orc_field_op(int field, int op, const u64* a, const u64* b, u64* out, size_t n) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    for (size_t i = 0; i < n; i++) {
        const fe* x = (const fe*)a + i; const fe* y = b ? (const fe*)b + i : x; fe* o = (fe*)out + i;
        switch (op) {
            case 0: f_add(F, o, x, y); break;
            case 1: f_sub(F, o, x, y); break;
            case 2: f_mul(F, o, x, y); break;
            case 3: f_inv(F, o, x); break;
            case 4: { fe t; f_to_mont(F, &t, x->v); *o = t; } break;   /* canonical -> Montgomery */
            case 5: { u64 t[4]; f_from_mont(F, t, x); memcpy(o, t, 32); } break;
            default: return -2;
        }
    }
    return 0;
}
int orc_field_info(int field, u64 p[4], u64 r[4], u64 r2[4], u64* inv) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    memcpy(p, FIELDS[field].p, 32); memcpy(r, FIELDS[field].r.v, 32); memcpy(r2, FIELDS[field].r2.v, 32); *inv = FIELDS[field].inv;
    return 0;
}

/* ---------- synthetic inputs ---------- */
typedef struct { u64 s[4]; } rng_t;
static inline u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
static void rng_seed(rng_t* g, u64 seed) {
    for (int i = 0; i < 4; i++) {
        seed += 0x9E3779B97F4A7C15ULL; u64 z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        g->s[i] = z ^ (z >> 31);
    }
}
static u64 rng_next(rng_t* g) {
    u64* s = g->s; u64 r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
}
/* uniform below a bound of `nb` bits given as limbs; rejection (matches pyoracle.Xoshiro.below) */
static void rng_below(rng_t* g, const u64 bound[4], int nb, u64 out[4]) {
    int words = (nb + 63) / 64;
    for (;;) {
        u64 v[4] = {0, 0, 0, 0};
        for (int i = 0; i < words; i++) v[i] = rng_next(g);
        if (nb % 64) v[words - 1] &= (((u64)1 << (nb % 64)) - 1);
        if (!geq(v, bound)) { memcpy(out, v, 32); return; }
    }
}
static int bitlen_minus1(const u64 b[4]) { /* bit_length(bound - 1), bound >= 2 */
    u64 one[4] = {1, 0, 0, 0}, t[4]; sub4(t, b, one);
    for (int i = 3; i >= 0; i--) if (t[i]) return 64 * i + 64 - __builtin_clzll(t[i]);
    return 1;
}
static void pow2_limbs(u64 o[4], int e) { memset(o, 0, 32); if (e < 256) o[e >> 6] = (u64)1 << (e & 63); }

/* dist: 0 uniform, 1 witness-like, 2 lookup-like.  Output in Montgomery form. */
int orc_fill_scalars(int field, int dist, u64 seed, size_t n, u64* out) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    rng_t g; rng_seed(&g, seed);
    int pb = bitlen_minus1(F->p);
    u64 table[340][4];
    if (dist == 2) for (int i = 0; i < 340; i++) rng_below(&g, F->p, pb, table[i]);
    u64 b8[4], b64m[4] = {0, 1, 0, 0}, b134[4];
    pow2_limbs(b8, 8); pow2_limbs(b134, 134);
    for (size_t i = 0; i < n; i++) {
        u64 v[4] = {0, 0, 0, 0};
        if (dist == 0) rng_below(&g, F->p, pb, v);
        else if (dist == 1) {
            u64 u = rng_next(&g) % 100;
            if (u < 30) { /* zero */ }
            else if (u < 45) rng_below(&g, b8, 8, v);
            else if (u < 70) rng_below(&g, b64m, 64, v);
            else if (u < 95) rng_below(&g, b134, 134, v);
            else rng_below(&g, F->p, pb, v);
        } else {
            if (rng_next(&g) % 100 < 60) { /* zero */ }
            else memcpy(v, table[rng_next(&g) % 340], 32);
        }
        fe m; f_to_mont(F, &m, v); memcpy(out + 4 * i, m.v, 32);
    }
    return 0;
}

/* P0 = G, P_i = P_{i-1} + [0x9e3779b97f4a7c15]G, normalised with one batch inversion */
int orc_synth_bases(int curve, size_t n, u64* out_xy) {
    if (curve < 0 || curve > 2) return -1;
    init_curves();
    const curve_t* C = &CURVES[curve]; const field_t* F = C->F;
    if (n == 0) return 0;
    jac g; g.x = C->g.x; g.y = C->g.y; g.z = F->r;
    jac gp; jac_set_id(&gp); jac base = g; u64 k = 0x9E3779B97F4A7C15ULL;
    while (k) { if (k & 1) jac_add(C, &gp, &gp, &base); jac_double(C, &base, &base); k >>= 1; }
    aff gpa; jac_to_affine(C, &gpa, &gp);
    jac* js = malloc(n * sizeof(jac)); fe* pre = malloc(n * sizeof(fe));
    js[0] = g;
    for (size_t i = 1; i < n; i++) jac_add_mixed(C, &js[i], &js[i - 1], &gpa);
    fe acc = F->r;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; f_mul(F, &acc, &acc, &js[i].z); }
    fe inv; f_inv(F, &inv, &acc);
    for (size_t i = n; i-- > 0;) {
        fe zi, zi2, zi3; f_mul(F, &zi, &inv, &pre[i]); f_mul(F, &inv, &inv, &js[i].z);
        f_sqr(F, &zi2, &zi); f_mul(F, &zi3, &zi2, &zi);
        aff a; f_mul(F, &a.x, &js[i].x, &zi2); f_mul(F, &a.y, &js[i].y, &zi3);
        memcpy(out_xy + 8 * i, &a, 64);
    }
    free(js); free(pre);
    return 0;
}

/* ---------- round 2: the element-wise steps of create_proof + SRS generation (test infrastructure) ---------- */
/* kate_division: [UPSTREAM halo2_proofs/src/arithmetic.rs kate_division] q = (a(X) - a(b)) / (X - b), len - 1 coefficients:
 * "for (q, r) in q.iter_mut().rev().zip(a.rev()) { lead = r - tmp; *q = lead; tmp = lead * (-b) }". */
int orc_kate_division(int field, const u64* a_, size_t n, const u64 point[4], u64* q_) {
    if (field < 0 || field > 3 || n == 0) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    const fe* a = (const fe*)a_; fe* q = (fe*)q_;
    fe nb, tmp; f_neg(F, &nb, (const fe*)point); memset(&tmp, 0, sizeof(tmp));
    for (size_t i = n - 1; i >= 1; i--) {            /* q[i-1] pairs with a[i] */
        fe lead; f_sub(F, &lead, &a[i], &tmp);
        q[i - 1] = lead;
        f_mul(F, &tmp, &lead, &nb);
    }
    return 0;
}
/* out[i] = sum_j coef[j] cols[j][i]; sub0 (may be NULL) subtracted from out[0]: the polynomial folds of
 * [UPSTREAM plonk/vanishing/prover.rs Constructed::evaluate, poly/kzg/multiopen/gwc/prover.rs]. */
int orc_lincomb(int field, const u64* const* cols, const u64* coefs, size_t count, size_t n, u64* out_, const u64* sub0) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    fe* out = (fe*)out_;
    memset(out, 0, n * sizeof(fe));
    for (size_t j = 0; j < count; j++) {
        const fe* c = (const fe*)cols[j]; const fe* k = (const fe*)(coefs + 4 * j);
        for (size_t i = 0; i < n; i++) { fe t; f_mul(F, &t, &c[i], k); f_add(F, &out[i], &out[i], &t); }
    }
    if (sub0 && n) f_sub(F, &out[0], &out[0], (const fe*)sub0);
    return 0;
}
/* a[i] *= pattern[i % period] ([UPSTREAM poly/domain.rs divide_by_vanishing_poly: "*h *= t_evaluations[index % len]"]) */
int orc_scale_periodic(int field, u64* a_, size_t n, const u64* pattern, size_t period) {
    if (field < 0 || field > 3 || period == 0) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    fe* a = (fe*)a_;
    for (size_t i = 0; i < n; i++) f_mul(F, &a[i], &a[i], (const fe*)(pattern + 4 * (i % period)));
    return 0;
}
/* out[i] = [scalars[i]] G (affine), G the curve's generator: ParamsKZG::setup's g = [s^i]G and g_lagrange = [L_i(s)]G
 * ([UPSTREAM poly/kzg/commitment.rs setup]: both are fixed-base multiplications once the scalars are known).  8-bit
 * windows over a 32 x 255 table, threads split the outputs. */
typedef struct { const curve_t* C; const field_t* FS; const aff* table; const fe* scalars; aff* out; size_t lo, hi; } fbm_job;
static void* fbm_worker(void* arg) {
    fbm_job* j = arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        u64 t[4]; f_from_mont(j->FS, t, &j->scalars[i]);
        const unsigned char* b = (const unsigned char*)t;
        jac acc; jac_set_id(&acc);
        for (int w = 0; w < 32; w++) if (b[w]) jac_add_mixed(j->C, &acc, &acc, &j->table[w * 255 + b[w] - 1]);
        jac_to_affine(j->C, &j->out[i], &acc);
    }
    return NULL;
}
int orc_fixed_base_mul(int curve, const u64* scalars, size_t n, int threads, u64* out_xy) {
    if (curve < 0 || curve > 2) return -1;
    init_curves();
    const curve_t* C = &CURVES[curve];
    aff* table = malloc(32 * 255 * sizeof(aff));
    jac base; base.x = C->g.x; base.y = C->g.y; base.z = C->F->r;
    for (int w = 0; w < 32; w++) {
        jac cur = base;
        for (int d = 1; d <= 255; d++) {
            jac_to_affine(C, &table[w * 255 + d - 1], &cur);
            jac nx; jac_add(C, &nx, &cur, &base); cur = nx;
        }
        base = cur;                                   /* 256 * base */
    }
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    fbm_job jobs[256]; pthread_t th[256];
    size_t chunk = (n + threads - 1) / threads; int used = 0;
    for (size_t lo = 0; lo < n; lo += chunk, used++) {
        jobs[used] = (fbm_job){C, scalar_field_of(curve), table, (const fe*)scalars, (aff*)out_xy, lo, lo + chunk < n ? lo + chunk : n};
        pthread_create(&th[used], NULL, fbm_worker, &jobs[used]);
    }
    for (int i = 0; i < used; i++) pthread_join(th[i], NULL);
    free(table);
    return 0;
}
/* out[i] = base^i * first (a column of powers: s^i, omega^i), serial like upstream's setup loop */
int orc_powers(int field, const u64 base[4], const u64 first[4], size_t n, u64* out_) {
    if (field < 0 || field > 3) return -1;
    init_fields();
    const field_t* F = &FIELDS[field];
    fe* out = (fe*)out_; fe cur = *(const fe*)first;
    for (size_t i = 0; i < n; i++) { out[i] = cur; f_mul(F, &cur, &cur, (const fe*)base); }
    return 0;
}
