// hostfield.hpp -- the O(1)-per-proof field arithmetic the HOST side of create_proof needs (challenges, opening points, powers of
// x and v, domain constants): 4 x u64 Montgomery limbs, R = 2^256, the in-memory form of halo2curves' Fr / Fq / Fp
// [UPSTREAM halo2curves bn256::{Fr,Fq}, pasta_curves::{Fp,Fq}].  Never used per element of a column: columns live on the device.
#pragma once
#include <cstdint>
#include <cstring>

#include "field_constants.h"

struct Fe {
    uint64_t v[4];
    bool operator==(const Fe& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    bool operator!=(const Fe& o) const { return !(*this == o); }
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
};

struct HostField {
    int id = -1;
    uint64_t p[4] = {}, inv = 0;       // modulus, -p^-1 mod 2^64
    Fe one{}, r2{}, r3{};              // R, R^2, R^3 mod p
    uint32_t two_adicity = 0, bits = 0;
    Fe gen{}, root_of_unity{}, zeta{}, delta{};   // Montgomery form

    static bool geq(const uint64_t a[4], const uint64_t b[4]) {
        for (int i = 3; i >= 0; i--) {
            if (a[i] > b[i]) return true;
            if (a[i] < b[i]) return false;
        }
        return true;
    }
    static uint64_t sub_limbs(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)a[i] - b[i] - (uint64_t)br;
            r[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
        return (uint64_t)br;
    }
    static uint64_t add_limbs(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (unsigned __int128)a[i] + b[i];
            r[i] = (uint64_t)c;
            c >>= 64;
        }
        return (uint64_t)c;
    }

    Fe add(const Fe& a, const Fe& b) const {
        Fe r;
        uint64_t c = add_limbs(r.v, a.v, b.v);
        if (c || geq(r.v, p)) sub_limbs(r.v, r.v, p);
        return r;
    }
    Fe sub(const Fe& a, const Fe& b) const {
        Fe r;
        if (sub_limbs(r.v, a.v, b.v)) add_limbs(r.v, r.v, p);
        return r;
    }
    Fe neg(const Fe& a) const { return a.is_zero() ? a : sub(Fe{{0, 0, 0, 0}}, a); }
    // Montgomery product a * b / R mod p (CIOS); a < 2^256 arbitrary, b < p
    Fe mul(const Fe& a, const Fe& b) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (unsigned __int128)a.v[j] * b.v[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv;
            c = (unsigned __int128)m * p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (unsigned __int128)m * p[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fe r{{t[0], t[1], t[2], t[3]}};
        if (t[4] || geq(r.v, p)) sub_limbs(r.v, r.v, p);
        return r;
    }
    Fe sqr(const Fe& a) const { return mul(a, a); }
    Fe from_canonical(const Fe& a) const { return mul(a, r2); }      // a < 2^256 (reduced on the way)
    Fe to_canonical(const Fe& a) const { return mul(a, Fe{{1, 0, 0, 0}}); }
    Fe from_u64(uint64_t x) const { return from_canonical(Fe{{x, 0, 0, 0}}); }
    Fe pow(const Fe& a, const uint64_t e[4]) const {
        Fe r = one;
        for (int i = 255; i >= 0; i--) {
            r = sqr(r);
            if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a);
        }
        return r;
    }
    Fe pow_u64(const Fe& a, uint64_t e) const {
        const uint64_t ee[4] = {e, 0, 0, 0};
        return pow(a, ee);
    }
    Fe invert(const Fe& a) const {      // a^(p - 2); 0 -> 0
        uint64_t e[4];
        const uint64_t two[4] = {2, 0, 0, 0};
        sub_limbs(e, p, two);
        return pow(a, e);
    }
    // 64 little-endian bytes taken as an integer, reduced mod p (FromUniformBytes<64>::from_uniform_bytes, Challenge255)
    Fe from_u512(const uint8_t b[64]) const {
        Fe lo, hi;
        memcpy(lo.v, b, 32);
        memcpy(hi.v, b + 32, 32);
        return add(mul(lo, r2), mul(hi, r3));       // lo R + hi 2^256 R
    }
    void to_bytes(const Fe& a, uint8_t out[32]) const {      // PrimeField::to_repr: canonical, little-endian
        const Fe c = to_canonical(a);
        memcpy(out, c.v, 32);
    }
};

namespace hostfield_detail {
template <class F>
inline HostField make(uint64_t gen) {
    HostField f;
    f.id = F::ID;
    f.two_adicity = F::TWO_ADICITY;
    for (int i = 0; i < 4; i++) f.p[i] = (uint64_t)F::P[2 * i] | ((uint64_t)F::P[2 * i + 1] << 32);
    uint64_t x = 1;                                   // Newton: p^-1 mod 2^64
    for (int i = 0; i < 6; i++) x *= 2 - f.p[0] * x;
    f.inv = (uint64_t)0 - x;
    f.bits = 256;
    while (f.bits && !((f.p[(f.bits - 1) >> 6] >> ((f.bits - 1) & 63)) & 1)) f.bits--;
    // R mod p by 256 doublings of 1, R^2 by 256 more (no constant is trusted that this file can compute)
    Fe a{{1, 0, 0, 0}};
    auto dbl = [&](Fe& z) {
        uint64_t c = HostField::add_limbs(z.v, z.v, z.v);
        if (c || HostField::geq(z.v, f.p)) HostField::sub_limbs(z.v, z.v, f.p);
    };
    for (int i = 0; i < 256; i++) dbl(a);
    f.one = a;
    for (int i = 0; i < 256; i++) dbl(a);
    f.r2 = a;
    f.r3 = f.mul(f.r2, f.r2);
    f.gen = f.from_u64(gen);
    // ROOT_OF_UNITY = gen^((p - 1) >> S), DELTA = gen^(2^S) [UPSTREAM ff::PrimeField]; ZETA is the curve crates' literal
    uint64_t t[4] = {f.p[0] - 1, f.p[1], f.p[2], f.p[3]};
    for (uint32_t s = 0; s < f.two_adicity; s++) {
        for (int i = 0; i < 3; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 63);
        t[3] >>= 1;
    }
    f.root_of_unity = f.pow(f.gen, t);
    f.delta = f.gen;
    for (uint32_t s = 0; s < f.two_adicity; s++) f.delta = f.sqr(f.delta);
    for (int i = 0; i < 4; i++) f.zeta.v[i] = (uint64_t)F::ZETA_M[2 * i] | ((uint64_t)F::ZETA_M[2 * i + 1] << 32);
    return f;
}
}   // namespace hostfield_detail

// dehalo_field id -> host field (MULTIPLICATIVE_GENERATOR: 7 bn256::Fr, 3 bn256::Fq, 5 pasta::Fp / Fq)
inline const HostField* host_field(int id) {
    static const HostField fields[4] = {hostfield_detail::make<Bn254Fr>(7), hostfield_detail::make<Bn254Fq>(3), hostfield_detail::make<PastaFp>(5),
                                        hostfield_detail::make<PastaFq>(5)};
    return (id >= 0 && id < 4) ? &fields[id] : nullptr;
}
// dehalo_curve id -> (base field id, scalar field id)
inline int curve_base_field(int curve) { return curve == 0 ? 1 : curve == 1 ? 2 : curve == 2 ? 3 : -1; }
inline int curve_scalar_field(int curve) { return curve == 0 ? 0 : curve == 1 ? 3 : curve == 2 ? 2 : -1; }
