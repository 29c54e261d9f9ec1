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

// ---- the whole call: plonk::{ConstraintSystem, keygen_vk, keygen_pk, create_proof}, ParamsKZG, Blake2bWrite ---------------------------
// (include/dehalo.h "the whole call"; reference call sites benches/delay_enc.rs:41-54, 84-115, 120-134)

// plonk::ConstraintSystem as the prover reads it, built the way a circuit's configure() builds it; expressions are handles.
class ConstraintSystem {
  public:
    using Expr = uint32_t;
    ConstraintSystem(uint32_t num_advice, uint32_t num_fixed, uint32_t num_instance) { d_.num_advice = num_advice; d_.num_fixed = num_fixed; d_.num_instance = num_instance; }
    Expr constant(const Fe& c) { return node(DEHALO_EXPR_CONSTANT, add_constant(c), 0, 0); }
    Expr query_advice(uint32_t col, int32_t rot = 0) { query(aq_, DEHALO_COLUMN_ADVICE, col, rot); return node(DEHALO_EXPR_ADVICE, col, 0, rot); }
    Expr query_fixed(uint32_t col, int32_t rot = 0) { query(fq_, DEHALO_COLUMN_FIXED, col, rot); return node(DEHALO_EXPR_FIXED, col, 0, rot); }
    Expr query_instance(uint32_t col, int32_t rot = 0) { query(iq_, DEHALO_COLUMN_INSTANCE, col, rot); return node(DEHALO_EXPR_INSTANCE, col, 0, rot); }
    Expr neg(Expr a) { return node(DEHALO_EXPR_NEGATED, a, 0, 0); }
    Expr sum(Expr a, Expr b) { return node(DEHALO_EXPR_SUM, a, b, 0); }
    Expr product(Expr a, Expr b) { return node(DEHALO_EXPR_PRODUCT, a, b, 0); }
    Expr scaled(Expr a, const Fe& c) { return node(DEHALO_EXPR_SCALED, a, add_constant(c), 0); }
    void enable_equality(dehalo_column_kind kind, uint32_t col) {
        std::vector<dehalo_column_query>& q = kind == DEHALO_COLUMN_ADVICE ? aq_ : kind == DEHALO_COLUMN_FIXED ? fq_ : iq_;
        query(q, kind, col, 0);
        for (auto& c : perm_) if (c.kind == (uint32_t)kind && c.index == col) return;
        perm_.push_back(dehalo_column_query{(uint32_t)kind, col, 0});
    }
    void create_gate(const std::vector<Expr>& polys) { gates_.insert(gates_.end(), polys.begin(), polys.end()); }
    void lookup(const std::vector<std::pair<Expr, Expr>>& pairs) {
        lens_.push_back((uint32_t)pairs.size());
        for (auto& pr : pairs) { lin_.push_back(pr.first); ltab_.push_back(pr.second); }
    }
    void set_minimum_degree(uint32_t d) { d_.minimum_degree = d; }
    const dehalo_constraint_system* descriptor() {
        d_.nodes = nodes_.data(); d_.num_nodes = (uint32_t)nodes_.size();
        d_.constants = consts_.empty() ? nullptr : consts_[0].data(); d_.num_constants = (uint32_t)consts_.size();
        d_.gates = gates_.data(); d_.num_gates = (uint32_t)gates_.size();
        d_.lookup_lens = lens_.data(); d_.num_lookups = (uint32_t)lens_.size(); d_.lookup_inputs = lin_.data(); d_.lookup_tables = ltab_.data();
        d_.permutation_columns = perm_.data(); d_.num_permutation_columns = (uint32_t)perm_.size();
        d_.advice_queries = aq_.data(); d_.num_advice_queries = (uint32_t)aq_.size();
        d_.fixed_queries = fq_.data(); d_.num_fixed_queries = (uint32_t)fq_.size();
        d_.instance_queries = iq_.data(); d_.num_instance_queries = (uint32_t)iq_.size();
        return &d_;
    }

  private:
    Expr node(uint32_t kind, uint32_t a, uint32_t b, int32_t rot) { nodes_.push_back(dehalo_expr_node{kind, a, b, rot}); return (Expr)nodes_.size() - 1; }
    uint32_t add_constant(const Fe& c) {
        for (size_t i = 0; i < consts_.size(); i++) if (consts_[i] == c) return (uint32_t)i;
        consts_.push_back(c);
        return (uint32_t)consts_.size() - 1;
    }
    static void query(std::vector<dehalo_column_query>& q, uint32_t kind, uint32_t col, int32_t rot) {
        for (auto& e : q) if (e.index == col && e.rotation == rot) return;
        q.push_back(dehalo_column_query{kind, col, rot});
    }
    dehalo_constraint_system d_{};
    std::vector<dehalo_expr_node> nodes_;
    std::vector<Fe> consts_;
    std::vector<uint32_t> gates_, lens_, lin_, ltab_;
    std::vector<dehalo_column_query> perm_, aq_, fq_, iq_;
};

// ParamsKZG<Bn256>
class ParamsKZG {
  public:
    ParamsKZG(const Backend& be, dehalo_curve curve, const std::vector<uint8_t>& raw_bytes) : be_(be) {      // ParamsKZG::read (RawBytes)
        be_.check(dehalo_params_read(be_.raw(), curve, raw_bytes.data(), raw_bytes.size(), &p_));
    }
    // ParamsKZG::setup(k, rng) with the rng's draw handed over: `s` = the secret scalar in Montgomery form (benches/delay_enc.rs:43); made on the device
    ParamsKZG(const Backend& be, dehalo_curve curve, uint32_t k, const Fe& s) : be_(be) { be_.check(dehalo_params_setup(be_.raw(), curve, k, s.data(), &p_)); }
    ~ParamsKZG() { dehalo_params_release(be_.raw(), p_); }
    ParamsKZG(const ParamsKZG&) = delete;
    std::vector<uint8_t> write() const {
        std::vector<uint8_t> out(dehalo_params_size(p_));
        be_.check(dehalo_params_write(p_, out.data(), out.size()));
        return out;
    }
    dehalo_params* raw() const { return p_; }

  private:
    const Backend& be_;
    dehalo_params* p_ = nullptr;
};

// ProvingKey<G1Affine> (with its VerifyingKey)
class ProvingKey {
  public:
    // keygen_vk + keygen_pk: fixed = num_fixed x 2^k elements (Montgomery), mapping = permutation assembly (cell -> cell)
    ProvingKey(const Backend& be, const ParamsKZG& params, ConstraintSystem& cs, const std::vector<Fe>& fixed, const std::vector<uint64_t>& mapping,
               const std::vector<std::vector<uint8_t>>& selectors = {}, uint32_t flags = 0) : be_(be) {
        std::vector<const uint8_t*> sp;
        for (auto& s : selectors) sp.push_back(s.data());
        be_.check(dehalo_keygen(be_.raw(), params.raw(), cs.descriptor(), fixed.empty() ? nullptr : fixed[0].data(), mapping.data(), sp.data(), (uint32_t)sp.size(), flags, &pk_));
    }
    // ProvingKey::read::<_, Circuit>(.., SerdeFormat::RawBytes)
    ProvingKey(const Backend& be, dehalo_curve curve, ConstraintSystem& cs, const std::vector<uint8_t>& raw_bytes, uint32_t num_selectors) : be_(be) {
        be_.check(dehalo_pk_read(be_.raw(), curve, cs.descriptor(), raw_bytes.data(), raw_bytes.size(), num_selectors, &pk_));
    }
    ~ProvingKey() { dehalo_pk_release(be_.raw(), pk_); }
    ProvingKey(const ProvingKey&) = delete;
    std::vector<uint8_t> write() const {
        std::vector<uint8_t> out(dehalo_pk_size(pk_));
        be_.check(dehalo_pk_write(be_.raw(), pk_, out.data(), out.size()));
        return out;
    }
    std::vector<uint8_t> vk_write() const {
        std::vector<uint8_t> out(dehalo_vk_size(pk_));
        be_.check(dehalo_vk_write(pk_, out.data(), out.size()));
        return out;
    }
    void set_transcript_repr(const Fe& r) { be_.check(dehalo_pk_set_transcript_repr(pk_, r.data())); }
    dehalo_pk* raw() const { return pk_; }

  private:
    const Backend& be_;
    dehalo_pk* pk_ = nullptr;
};

// Blake2bWrite<Vec<u8>, G1Affine, Challenge255<_>>
class Blake2bWrite {
  public:
    explicit Blake2bWrite(dehalo_curve curve) {
        if (dehalo_transcript_create(curve, &t_) != 0) throw std::runtime_error("dehalo_transcript_create failed");
    }
    ~Blake2bWrite() { dehalo_transcript_release(t_); }
    Blake2bWrite(const Blake2bWrite&) = delete;
    std::vector<uint8_t> finalize() const {
        std::vector<uint8_t> out(dehalo_transcript_len(t_));
        if (dehalo_transcript_finalize(t_, out.data(), out.size()) != 0) throw std::runtime_error("dehalo_transcript_finalize failed");
        return out;
    }
    dehalo_transcript* raw() const { return t_; }

  private:
    dehalo_transcript* t_ = nullptr;
};

// create_proof(&params, &pk, &[circuit], &[instances], rng, &mut transcript): `advice` is what circuit.synthesize assigned
class Prover {
  public:
    Prover(const Backend& be, const ParamsKZG& params, const ProvingKey& pk, const Backend* side = nullptr) : be_(be) {
        be_.check(dehalo_prover_create(be_.raw(), side ? side->raw() : nullptr, params.raw(), pk.raw(), &p_));
    }
    ~Prover() { dehalo_prover_release(p_); }
    Prover(const Prover&) = delete;
    // rng == nullptr: operating-system entropy (the reference's OsRng)
    void create_proof(const std::vector<Fe>& advice, const std::vector<std::vector<Fe>>& instances, dehalo_rng* rng, Blake2bWrite& transcript) const {
        std::vector<const uint64_t*> ip;
        std::vector<size_t> il;
        for (auto& col : instances) { ip.push_back(col.empty() ? nullptr : col[0].data()); il.push_back(col.size()); }
        be_.check(dehalo_create_proof(p_, advice[0].data(), ip.data(), il.data(), (uint32_t)ip.size(), rng, transcript.raw(), 0));
    }

  private:
    const Backend& be_;
    dehalo_prover* p_ = nullptr;
};

}  // namespace halo2_amd
