// halo2_backend.hpp -- C++ host-side mirror of the upstream interface the reference reaches
// through create_proof (benches/delay_enc.rs:123-131): halo2_proofs::arithmetic::{best_multiexp,
// best_fft}, poly::EvaluationDomain::{lagrange_to_coeff, coeff_to_extended, extended_to_coeff}
// and ParamsKZG::{commit, commit_lagrange} [UPSTREAM halo2_proofs @ v2023_04_20], as thin
// wrappers over the C ABI (include/dehalo.h).  Same names and argument meaning; upstream's
// assert_eq! panics become std::invalid_argument, library errors std::runtime_error.
// Header-only; link with -ldehalo.  No CPU path: constructing Backend without a gfx950 throws.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dehalo.h"

namespace halo2_amd {

using Fe = std::array<uint64_t, 4>;        // halo2curves field element (Montgomery, 4 x u64 LE)
using Affine = std::array<uint64_t, 8>;    // {x, y}; identity = all zero
using Projective = std::array<uint64_t, 12>;  // Jacobian {x, y, z}; identity z = 0

class Backend {
  public:
    explicit Backend(int device = 0) {
        int rc = dehalo_ctx_create(device, &ctx_);
        if (rc != 0) throw std::runtime_error("dehalo_ctx_create failed (" + std::to_string(rc) + "): no gfx950 device; there is no CPU fallback");
    }
    ~Backend() { dehalo_ctx_destroy(ctx_); }
    Backend(const Backend&) = delete;
    Backend& operator=(const Backend&) = delete;
    dehalo_ctx* raw() const { return ctx_; }

    // arithmetic::best_multiexp(coeffs, bases) -> C::Curve
    Projective best_multiexp(dehalo_curve curve, const std::vector<Fe>& coeffs, const std::vector<Affine>& bases) const {
        if (coeffs.size() != bases.size()) throw std::invalid_argument("best_multiexp: coeffs.len() != bases.len()");
        Projective out{};
        check(dehalo_best_multiexp(ctx_, curve, coeffs.empty() ? nullptr : coeffs[0].data(), bases.empty() ? nullptr : bases[0].data(), coeffs.size(), out.data()));
        return out;
    }
    // arithmetic::best_fft(a, omega, log_n): in place
    void best_fft(dehalo_field field, std::vector<Fe>& a, const Fe& omega, uint32_t log_n) const {
        if (a.size() != (size_t(1) << log_n)) throw std::invalid_argument("best_fft: a.len() != 1 << log_n");
        check(dehalo_ntt(ctx_, field, a[0].data(), log_n, omega.data()));
    }
    // arithmetic::eval_polynomial(poly, point) -> F
    Fe eval_polynomial(dehalo_field field, const std::vector<Fe>& poly, const Fe& point) const {
        Fe out{};
        check(dehalo_eval_polynomial(ctx_, field, poly.empty() ? nullptr : poly[0].data(), poly.size(), point.data(), out.data()));
        return out;
    }
    // ff::BatchInvert: in place, zeros stay zero
    void batch_invert(dehalo_field field, std::vector<Fe>& values) const {
        check(dehalo_batch_invert(ctx_, field, values.empty() ? nullptr : values[0].data(), values.size()));
    }
    // z[0] = 1, z[i] = prod_{j<i} num[j] / den[j]  (permutation::commit / lookup::commit_product without the blinding rows)
    std::vector<Fe> grand_product(dehalo_field field, const std::vector<Fe>& num, const std::vector<Fe>& den) const {
        if (num.size() != den.size()) throw std::invalid_argument("grand_product: num.len() != den.len()");
        std::vector<Fe> z(num.size());
        if (!num.empty()) check(dehalo_grand_product(ctx_, field, num[0].data(), den[0].data(), num.size(), z[0].data()));
        return z;
    }
    // arithmetic::kate_division(a, point) -> (a(X) - a(point)) / (X - point), a.len() - 1 coefficients
    std::vector<Fe> kate_division(dehalo_field field, const std::vector<Fe>& a, const Fe& point) const {
        if (a.empty()) throw std::invalid_argument("kate_division: empty polynomial");
        std::vector<Fe> q(a.size() - 1);
        check(dehalo_kate_division(ctx_, field, a[0].data(), a.size(), point.data(), q.empty() ? nullptr : q[0].data()));
        return q;
    }
    // plonk::lookup::prover::permute_expression_pair over the first `usable_rows` values (the caller appends the blinding rows);
    // Err(ConstraintSystemFailure) -> std::runtime_error carrying DEHALO_ERR_NOT_IN_TABLE
    std::pair<std::vector<Fe>, std::vector<Fe>> permute_expression_pair(dehalo_field field, const std::vector<Fe>& input, const std::vector<Fe>& table,
                                                                        size_t usable_rows) const {
        if (input.size() < usable_rows || table.size() < usable_rows) throw std::invalid_argument("permute_expression_pair: fewer than usable_rows values");
        std::vector<Fe> pi(usable_rows), pt(usable_rows);
        if (usable_rows) check(dehalo_permute_expression_pair(ctx_, field, input[0].data(), table[0].data(), usable_rows, pi[0].data(), pt[0].data()));
        return {pi, pt};
    }
    void check(int rc) const {
        if (rc != 0) throw std::runtime_error(std::string("dehalo error ") + std::to_string(rc) + ": " + dehalo_last_error(ctx_));
    }

  private:
    dehalo_ctx* ctx_ = nullptr;
};

// poly::EvaluationDomain -- the caller supplies the domain constants it already holds
// (omega_inv, ifft_divisor, extended_omega, ..., g_coset = F::ZETA), exactly upstream's fields.
struct EvaluationDomain {
    const Backend& be;
    dehalo_field field;
    uint32_t k, extended_k, quotient_poly_degree;
    Fe omega_inv, ifft_divisor, extended_omega, extended_omega_inv, extended_ifft_divisor, g_coset;

    void lagrange_to_coeff(std::vector<Fe>& a) const {
        if (a.size() != (size_t(1) << k)) throw std::invalid_argument("lagrange_to_coeff: wrong length");
        be.check(dehalo_intt_scaled(be.raw(), field, a[0].data(), k, omega_inv.data(), ifft_divisor.data()));
    }
    std::vector<Fe> coeff_to_extended(const std::vector<Fe>& a) const {
        if (a.size() != (size_t(1) << k)) throw std::invalid_argument("coeff_to_extended: wrong length");
        std::vector<Fe> ext(size_t(1) << extended_k);
        be.check(dehalo_coset_ntt(be.raw(), field, a[0].data(), k, ext[0].data(), extended_k, extended_omega.data(), g_coset.data()));
        return ext;
    }
    void extended_to_coeff(std::vector<Fe>& a) const {
        if (a.size() != (size_t(1) << extended_k)) throw std::invalid_argument("extended_to_coeff: wrong length");
        be.check(dehalo_coset_intt(be.raw(), field, a[0].data(), extended_k, extended_omega_inv.data(), extended_ifft_divisor.data(), g_coset.data()));
        a.resize((size_t(1) << k) * quotient_poly_degree);   // upstream truncates
    }
};

// ParamsKZG / ParamsIPA: g and g_lagrange resident in HBM
class Params {
  public:
    Params(const Backend& be, dehalo_curve curve, const std::vector<Affine>& g, const std::vector<Affine>* g_lagrange = nullptr) : be_(be) {
        be_.check(dehalo_bases_register(be_.raw(), curve, g[0].data(), g.size(), 64, 0, 1, &g_));
        if (g_lagrange) be_.check(dehalo_bases_register(be_.raw(), curve, (*g_lagrange)[0].data(), g_lagrange->size(), 64, 0, 1, &gl_));
    }
    ~Params() {
        if (g_) dehalo_bases_release(be_.raw(), g_);
        if (gl_) dehalo_bases_release(be_.raw(), gl_);
    }
    Projective commit(const std::vector<Fe>& poly) const { return msm(g_, poly); }
    Projective commit_lagrange(const std::vector<Fe>& poly) const {
        if (!gl_) throw std::invalid_argument("commit_lagrange: no g_lagrange registered");
        return msm(gl_, poly);
    }

  private:
    Projective msm(dehalo_bases* b, const std::vector<Fe>& poly) const {
        Projective out{};
        be_.check(dehalo_msm(be_.raw(), b, poly.empty() ? nullptr : poly[0].data(), poly.size(), out.data()));
        return out;
    }
    const Backend& be_;
    dehalo_bases* g_ = nullptr;
    dehalo_bases* gl_ = nullptr;
};

// ParamsIPA::{commit, commit_lagrange}(poly, r) = <poly, g> + r * w: tables registered over g || w (n + 1 points)
class ParamsIPA {
  public:
    ParamsIPA(const Backend& be, dehalo_curve curve, std::vector<Affine> g, std::vector<Affine> g_lagrange, const Affine& w) : be_(be), n_(g.size()) {
        if (g_lagrange.size() != n_) throw std::invalid_argument("ParamsIPA: g and g_lagrange differ in length");
        g.push_back(w);
        g_lagrange.push_back(w);
        be_.check(dehalo_bases_register(be_.raw(), curve, g[0].data(), g.size(), 64, 0, 1, &g_));
        be_.check(dehalo_bases_register(be_.raw(), curve, g_lagrange[0].data(), g_lagrange.size(), 64, 0, 1, &gl_));
    }
    ~ParamsIPA() {
        if (g_) dehalo_bases_release(be_.raw(), g_);
        if (gl_) dehalo_bases_release(be_.raw(), gl_);
    }
    Projective commit(const std::vector<Fe>& poly, const Fe& blind) const { return msm(g_, poly, blind); }
    Projective commit_lagrange(const std::vector<Fe>& poly, const Fe& blind) const { return msm(gl_, poly, blind); }

  private:
    Projective msm(dehalo_bases* b, std::vector<Fe> poly, const Fe& blind) const {
        if (poly.size() != n_) throw std::invalid_argument("commit: poly.len() != params.n");   // upstream: assert_eq!
        poly.push_back(blind);
        Projective out{};
        be_.check(dehalo_msm(be_.raw(), b, poly[0].data(), poly.size(), out.data()));
        return out;
    }
    const Backend& be_;
    size_t n_;
    dehalo_bases* g_ = nullptr;
    dehalo_bases* gl_ = nullptr;
};

}  // namespace halo2_amd
