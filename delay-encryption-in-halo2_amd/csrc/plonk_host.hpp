// plonk_host.hpp -- what create_proof / keygen read of a circuit, on the host: ConstraintSystem (the dehalo_constraint_system handed
// over the C ABI, validated and owned), its derived quantities [UPSTREAM halo2_proofs/src/plonk/circuit.rs: degree, blinding_factors;
// plonk/permutation.rs: required_degree, chunk length], Expression -> GraphEvaluator [UPSTREAM plonk/evaluation.rs
// GraphEvaluator::add_expression, Evaluator::new] and EvaluationDomain::new's constants [UPSTREAM poly/domain.rs].
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dehalo.h"
#include "hostfield.hpp"

struct HostCS {
    uint32_t num_advice = 0, num_fixed = 0, num_instance = 0, minimum_degree = 0;
    std::vector<dehalo_expr_node> nodes;
    std::vector<Fe> constants;
    std::vector<uint32_t> gates;
    struct Lookup { std::vector<uint32_t> inputs, tables; };
    std::vector<Lookup> lookups;
    std::vector<dehalo_column_query> perm_cols, advice_q, fixed_q, instance_q;

    // copies and validates; "" on success, otherwise what is wrong
    std::string load(const dehalo_constraint_system* d) {
        if (!d) return "null constraint system";
        num_advice = d->num_advice; num_fixed = d->num_fixed; num_instance = d->num_instance; minimum_degree = d->minimum_degree;
        if ((d->num_nodes && !d->nodes) || (d->num_constants && !d->constants) || (d->num_gates && !d->gates) ||
            (d->num_lookups && (!d->lookup_lens || !d->lookup_inputs || !d->lookup_tables)) || (d->num_permutation_columns && !d->permutation_columns) ||
            (d->num_advice_queries && !d->advice_queries) || (d->num_fixed_queries && !d->fixed_queries) || (d->num_instance_queries && !d->instance_queries))
            return "constraint system: null array with a non-zero count";
        nodes.assign(d->nodes, d->nodes + d->num_nodes);
        constants.resize(d->num_constants);
        for (uint32_t i = 0; i < d->num_constants; i++) memcpy(constants[i].v, d->constants + 4 * i, 32);
        for (uint32_t i = 0; i < d->num_nodes; i++) {
            const dehalo_expr_node& e = nodes[i];
            switch (e.kind) {
                case DEHALO_EXPR_CONSTANT: if (e.a >= d->num_constants) return "expression: constant index out of range"; break;
                case DEHALO_EXPR_FIXED: if (e.a >= num_fixed) return "expression: fixed column out of range"; break;
                case DEHALO_EXPR_ADVICE: if (e.a >= num_advice) return "expression: advice column out of range"; break;
                case DEHALO_EXPR_INSTANCE: if (e.a >= num_instance) return "expression: instance column out of range"; break;
                case DEHALO_EXPR_NEGATED: if (e.a >= i) return "expression: child must precede its parent"; break;
                case DEHALO_EXPR_SUM: case DEHALO_EXPR_PRODUCT: if (e.a >= i || e.b >= i) return "expression: child must precede its parent"; break;
                case DEHALO_EXPR_SCALED: if (e.a >= i || e.b >= d->num_constants) return "expression: bad scaled node"; break;
                default: return "expression: unknown node kind";
            }
        }
        gates.assign(d->gates, d->gates + d->num_gates);
        for (uint32_t g : gates) if (g >= d->num_nodes) return "gate root out of range";
        lookups.clear();
        size_t off = 0;
        for (uint32_t l = 0; l < d->num_lookups; l++) {
            Lookup lk;
            for (uint32_t i = 0; i < d->lookup_lens[l]; i++, off++) {
                if (d->lookup_inputs[off] >= d->num_nodes || d->lookup_tables[off] >= d->num_nodes) return "lookup expression root out of range";
                lk.inputs.push_back(d->lookup_inputs[off]);
                lk.tables.push_back(d->lookup_tables[off]);
            }
            if (lk.inputs.empty()) return "lookup without expressions";
            lookups.push_back(lk);
        }
        auto copyq = [](std::vector<dehalo_column_query>& dst, const dehalo_column_query* src, uint32_t cnt) { dst.assign(src, src + cnt); };
        copyq(perm_cols, d->permutation_columns, d->num_permutation_columns);
        copyq(advice_q, d->advice_queries, d->num_advice_queries);
        copyq(fixed_q, d->fixed_queries, d->num_fixed_queries);
        copyq(instance_q, d->instance_queries, d->num_instance_queries);
        for (auto& q : perm_cols) {
            const uint32_t lim = q.kind == DEHALO_COLUMN_ADVICE ? num_advice : q.kind == DEHALO_COLUMN_FIXED ? num_fixed : q.kind == DEHALO_COLUMN_INSTANCE ? num_instance : 0;
            if (q.index >= lim) return "permutation column out of range";
        }
        for (auto& q : advice_q) if (q.index >= num_advice) return "advice query out of range";
        for (auto& q : fixed_q) if (q.index >= num_fixed) return "fixed query out of range";
        for (auto& q : instance_q) if (q.index >= num_instance) return "instance query out of range";
        return "";
    }

    uint32_t expr_degree(uint32_t i) const {      // Expression::degree
        const dehalo_expr_node& e = nodes[i];
        switch (e.kind) {
            case DEHALO_EXPR_CONSTANT: return 0;
            case DEHALO_EXPR_FIXED: case DEHALO_EXPR_ADVICE: case DEHALO_EXPR_INSTANCE: return 1;
            case DEHALO_EXPR_NEGATED: case DEHALO_EXPR_SCALED: return expr_degree(e.a);
            case DEHALO_EXPR_SUM: return std::max(expr_degree(e.a), expr_degree(e.b));
            default: return expr_degree(e.a) + expr_degree(e.b);
        }
    }
    bool expr_equal(uint32_t a, uint32_t b) const {      // structural equality of two expressions
        if (a == b) return true;
        const dehalo_expr_node &x = nodes[a], &y = nodes[b];
        if (x.kind != y.kind) return false;
        switch (x.kind) {
            case DEHALO_EXPR_CONSTANT: return constants[x.a] == constants[y.a];
            case DEHALO_EXPR_FIXED: case DEHALO_EXPR_ADVICE: case DEHALO_EXPR_INSTANCE: return x.a == y.a && x.rotation == y.rotation;
            case DEHALO_EXPR_NEGATED: return expr_equal(x.a, y.a);
            case DEHALO_EXPR_SCALED: return constants[x.b] == constants[y.b] && expr_equal(x.a, y.a);
            default: return expr_equal(x.a, y.a) && expr_equal(x.b, y.b);
        }
    }
    bool expr_fixed_only(uint32_t i) const {      // no advice / instance query anywhere below: the expression's values belong to the proving key
        const dehalo_expr_node& e = nodes[i];
        switch (e.kind) {
            case DEHALO_EXPR_CONSTANT: case DEHALO_EXPR_FIXED: return true;
            case DEHALO_EXPR_ADVICE: case DEHALO_EXPR_INSTANCE: return false;
            case DEHALO_EXPR_NEGATED: case DEHALO_EXPR_SCALED: return expr_fixed_only(e.a);
            default: return expr_fixed_only(e.a) && expr_fixed_only(e.b);
        }
    }
    void expr_fixed_columns(uint32_t i, std::vector<uint32_t>& out) const {      // the fixed columns an expression reads
        const dehalo_expr_node& e = nodes[i];
        switch (e.kind) {
            case DEHALO_EXPR_FIXED: if (std::find(out.begin(), out.end(), e.a) == out.end()) out.push_back(e.a); break;
            case DEHALO_EXPR_NEGATED: case DEHALO_EXPR_SCALED: expr_fixed_columns(e.a, out); break;
            case DEHALO_EXPR_SUM: case DEHALO_EXPR_PRODUCT: expr_fixed_columns(e.a, out); expr_fixed_columns(e.b, out); break;
            default: break;
        }
    }
    // the first lookup whose table expressions equal lookup l's (l itself if none before it): their theta-compressed table columns are
    // the same column, computed and sorted once
    uint32_t table_representative(uint32_t l) const {
        for (uint32_t m = 0; m < l; m++) {
            if (lookups[m].tables.size() != lookups[l].tables.size()) continue;
            bool same = true;
            for (size_t i = 0; i < lookups[l].tables.size() && same; i++) same = expr_equal(lookups[m].tables[i], lookups[l].tables[i]);
            if (same) return m;
        }
        return l;
    }
    uint32_t blinding_factors() const {           // max(3, most queries to one advice column) + 2
        std::vector<uint32_t> cnt(num_advice, 0);
        for (auto& q : advice_q) cnt[q.index]++;
        uint32_t factors = 1;
        for (uint32_t c : cnt) factors = std::max(factors, c);
        if (cnt.empty()) factors = 1;
        return std::max<uint32_t>(3, factors) + 2;
    }
    uint32_t degree() const {                     // permutation: 3; a lookup: max(4, 2 + input + table); gates: their own
        uint32_t d = 3;
        for (auto& lk : lookups) {
            uint32_t di = 1, dt = 1;
            for (uint32_t e : lk.inputs) di = std::max(di, expr_degree(e));
            for (uint32_t e : lk.tables) dt = std::max(dt, expr_degree(e));
            d = std::max(d, std::max<uint32_t>(4, 2 + di + dt));
        }
        for (uint32_t g : gates) d = std::max(d, expr_degree(g));
        return std::max(d, minimum_degree);
    }
    uint32_t chunk_len() const { return degree() - 2; }
    uint32_t num_sets() const {
        const uint32_t c = chunk_len();
        return ((uint32_t)perm_cols.size() + c - 1) / c;
    }
    // a binary encoding of everything above: hashed into the substitute transcript_repr
    void encode(std::vector<uint8_t>& out) const {
        auto u32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) out.push_back((uint8_t)(v >> (8 * i))); };
        u32(num_advice); u32(num_fixed); u32(num_instance); u32(minimum_degree);
        u32((uint32_t)nodes.size());
        for (auto& e : nodes) { u32(e.kind); u32(e.a); u32(e.b); u32((uint32_t)e.rotation); }
        u32((uint32_t)constants.size());
        for (auto& c : constants) out.insert(out.end(), (const uint8_t*)c.v, (const uint8_t*)c.v + 32);
        u32((uint32_t)gates.size());
        for (uint32_t g : gates) u32(g);
        u32((uint32_t)lookups.size());
        for (auto& lk : lookups) { u32((uint32_t)lk.inputs.size()); for (size_t i = 0; i < lk.inputs.size(); i++) { u32(lk.inputs[i]); u32(lk.tables[i]); } }
        for (auto* v : {&perm_cols, &advice_q, &fixed_q, &instance_q}) { u32((uint32_t)v->size()); for (auto& q : *v) { u32(q.kind); u32(q.index); u32((uint32_t)q.rotation); } }
    }
};

// ---- GraphEvaluator under construction (plonk/evaluation.rs) ----
struct GSrc {
    uint32_t kind, index, rot;
    bool operator==(const GSrc& o) const { return kind == o.kind && index == o.index && rot == o.rot; }
    bool operator<=(const GSrc& o) const {      // upstream derives PartialOrd on ValueSource: (variant, fields) lexicographic
        if (kind != o.kind) return kind < o.kind;
        if (index != o.index) return index < o.index;
        return rot <= o.rot;
    }
};
struct GCalc { uint32_t op; GSrc a, b; std::vector<GSrc> parts; uint32_t target; };

struct GraphBuilder {
    const HostField* f;
    std::vector<Fe> constants;
    std::vector<int32_t> rotations;
    std::vector<GCalc> calcs;
    uint32_t num_intermediates = 0;
    static constexpr GSrc ZERO{DEHALO_SRC_CONSTANT, 0, 0}, ONE{DEHALO_SRC_CONSTANT, 1, 0}, TWO{DEHALO_SRC_CONSTANT, 2, 0};

    explicit GraphBuilder(const HostField* field) : f(field) {      // GraphEvaluator::default(): constants [0, 1, 2]
        constants = {Fe{{0, 0, 0, 0}}, f->one, f->add(f->one, f->one)};
    }
    uint32_t add_rotation(int32_t r) {
        for (size_t i = 0; i < rotations.size(); i++) if (rotations[i] == r) return (uint32_t)i;
        rotations.push_back(r);
        return (uint32_t)rotations.size() - 1;
    }
    GSrc add_constant(const Fe& c) {
        for (size_t i = 0; i < constants.size(); i++) if (constants[i] == c) return GSrc{DEHALO_SRC_CONSTANT, (uint32_t)i, 0};
        constants.push_back(c);
        return GSrc{DEHALO_SRC_CONSTANT, (uint32_t)constants.size() - 1, 0};
    }
    GSrc add_calc(uint32_t op, GSrc a, GSrc b = ZERO, std::vector<GSrc> parts = {}) {
        for (auto& c : calcs)
            if (c.op == op && c.a == a && c.b == b && c.parts == parts) return GSrc{DEHALO_SRC_INTERMEDIATE, c.target, 0};
        const uint32_t target = num_intermediates++;
        calcs.push_back(GCalc{op, a, b, std::move(parts), target});
        return GSrc{DEHALO_SRC_INTERMEDIATE, target, 0};
    }
    GSrc column(uint32_t kind, uint32_t index, int32_t rot = 0) { return GSrc{kind, index, add_rotation(rot)}; }

    GSrc add_expression(const HostCS& cs, uint32_t i) {
        const dehalo_expr_node& e = cs.nodes[i];
        switch (e.kind) {
            case DEHALO_EXPR_CONSTANT: return add_constant(cs.constants[e.a]);
            case DEHALO_EXPR_FIXED: return add_calc(DEHALO_CALC_STORE, column(DEHALO_SRC_FIXED, e.a, e.rotation));
            case DEHALO_EXPR_ADVICE: return add_calc(DEHALO_CALC_STORE, column(DEHALO_SRC_ADVICE, e.a, e.rotation));
            case DEHALO_EXPR_INSTANCE: return add_calc(DEHALO_CALC_STORE, column(DEHALO_SRC_INSTANCE, e.a, e.rotation));
            case DEHALO_EXPR_NEGATED: {
                if (cs.nodes[e.a].kind == DEHALO_EXPR_CONSTANT) return add_constant(f->neg(cs.constants[cs.nodes[e.a].a]));
                const GSrc a = add_expression(cs, e.a);
                return a == ZERO ? a : add_calc(DEHALO_CALC_NEGATE, a);
            }
            case DEHALO_EXPR_SUM: {
                if (cs.nodes[e.b].kind == DEHALO_EXPR_NEGATED) {      // a - b
                    const GSrc a = add_expression(cs, e.a), b = add_expression(cs, cs.nodes[e.b].a);
                    if (a == ZERO) return add_calc(DEHALO_CALC_NEGATE, b);
                    return b == ZERO ? a : add_calc(DEHALO_CALC_SUB, a, b);
                }
                const GSrc a = add_expression(cs, e.a), b = add_expression(cs, e.b);
                if (a == ZERO) return b;
                if (b == ZERO) return a;
                return a <= b ? add_calc(DEHALO_CALC_ADD, a, b) : add_calc(DEHALO_CALC_ADD, b, a);
            }
            case DEHALO_EXPR_PRODUCT: {
                const GSrc a = add_expression(cs, e.a), b = add_expression(cs, e.b);
                if (a == ZERO || b == ZERO) return ZERO;
                if (a == ONE) return b;
                if (b == ONE) return a;
                if (a == TWO) return add_calc(DEHALO_CALC_DOUBLE, b);
                if (b == TWO) return add_calc(DEHALO_CALC_DOUBLE, a);
                if (a == b) return add_calc(DEHALO_CALC_SQUARE, a);
                return a <= b ? add_calc(DEHALO_CALC_MUL, a, b) : add_calc(DEHALO_CALC_MUL, b, a);
            }
            default: {      // SCALED
                const Fe& c = cs.constants[e.b];
                if (c.is_zero()) return ZERO;
                if (c == f->one) return add_expression(cs, e.a);
                const GSrc cst = add_constant(c);
                const GSrc a = add_expression(cs, e.a);
                return add_calc(DEHALO_CALC_MUL, a, cst);
            }
        }
    }

    int compile(dehalo_ctx* ctx, dehalo_graph** out) const {
        std::vector<dehalo_calculation> cc;
        std::vector<dehalo_source> parts;
        auto src = [](const GSrc& s) { return dehalo_source{s.kind, s.index, s.rot}; };
        for (auto& c : calcs) {
            dehalo_calculation d{};
            d.op = c.op;
            d.a = src(c.a);
            d.b = src(c.b);
            d.parts_begin = (uint32_t)parts.size();
            d.parts_len = (uint32_t)c.parts.size();
            d.target = c.target;
            for (auto& p : c.parts) parts.push_back(src(p));
            cc.push_back(d);
        }
        return dehalo_graph_create(ctx, f->id, (const uint64_t*)constants.data(), (uint32_t)constants.size(), rotations.data(), (uint32_t)rotations.size(), cc.data(),
                                   (uint32_t)cc.size(), parts.data(), (uint32_t)parts.size(), num_intermediates, out);
    }
};

// Evaluator::new, custom gates: value = Horner(previous, gate polynomials, y)
inline GraphBuilder custom_gates_graph(const HostCS& cs, const HostField* f) {
    GraphBuilder g(f);
    std::vector<GSrc> parts;
    for (uint32_t poly : cs.gates) parts.push_back(g.add_expression(cs, poly));
    g.add_calc(DEHALO_CALC_HORNER, GSrc{DEHALO_SRC_PREVIOUS, 0, 0}, GSrc{DEHALO_SRC_Y, 0, 0}, parts);
    return g;
}
// Evaluator::new, one lookup: (theta-compressed input + beta) * (theta-compressed table + gamma)
inline GraphBuilder lookup_table_value_graph(const HostCS& cs, const HostCS::Lookup& lk, const HostField* f) {
    GraphBuilder g(f);
    std::vector<GSrc> pi, pt;
    for (uint32_t e : lk.inputs) pi.push_back(g.add_expression(cs, e));
    const GSrc ci = g.add_calc(DEHALO_CALC_HORNER, GraphBuilder::ZERO, GSrc{DEHALO_SRC_THETA, 0, 0}, pi);
    for (uint32_t e : lk.tables) pt.push_back(g.add_expression(cs, e));
    const GSrc ct = g.add_calc(DEHALO_CALC_HORNER, GraphBuilder::ZERO, GSrc{DEHALO_SRC_THETA, 0, 0}, pt);
    const GSrc right = g.add_calc(DEHALO_CALC_ADD, ct, GSrc{DEHALO_SRC_GAMMA, 0, 0});
    const GSrc left = g.add_calc(DEHALO_CALC_ADD, ci, GSrc{DEHALO_SRC_BETA, 0, 0});
    g.add_calc(DEHALO_CALC_MUL, left, right);
    return g;
}
// lookup::Argument::commit_permuted's compress_expressions over the rows of the ORIGINAL domain
inline GraphBuilder compress_graph(const HostCS& cs, const std::vector<uint32_t>& exprs, const HostField* f) {
    GraphBuilder g(f);
    std::vector<GSrc> parts;
    for (uint32_t e : exprs) parts.push_back(g.add_expression(cs, e));
    g.add_calc(DEHALO_CALC_HORNER, GraphBuilder::ZERO, GSrc{DEHALO_SRC_THETA, 0, 0}, parts);
    return g;
}

// ---- EvaluationDomain::new(j, k) ----
struct HostDomain {
    const HostField* f = nullptr;
    uint32_t k = 0, j = 0, quotient_poly_degree = 0, extended_k = 0;
    size_t n = 0, m = 0;
    Fe omega, omega_inv, ext_omega, ext_omega_inv, ifft_divisor, ext_ifft_divisor, g_coset, g_coset_inv;
    std::vector<Fe> t_inv;      // 1 / t(X) on the coset: 2^(extended_k - k) values

    bool init(const HostField* field, uint32_t j_, uint32_t k_) {
        f = field; j = j_; k = k_;
        n = (size_t)1 << k;
        quotient_poly_degree = j - 1;
        extended_k = k;
        while (((size_t)1 << extended_k) < n * quotient_poly_degree) extended_k++;
        if (extended_k > f->two_adicity) return false;
        m = (size_t)1 << extended_k;
        ext_omega = f->root_of_unity;
        for (uint32_t i = extended_k; i < f->two_adicity; i++) ext_omega = f->sqr(ext_omega);
        omega = ext_omega;
        for (uint32_t i = k; i < extended_k; i++) omega = f->sqr(omega);
        omega_inv = f->invert(omega);
        ext_omega_inv = f->invert(ext_omega);
        ifft_divisor = f->invert(f->from_u64((uint64_t)n));
        ext_ifft_divisor = f->invert(f->from_u64((uint64_t)m));
        g_coset = f->zeta;
        g_coset_inv = f->sqr(g_coset);
        const Fe orig = f->pow_u64(g_coset, (uint64_t)n), step = f->pow_u64(ext_omega, (uint64_t)n);
        t_inv.clear();
        Fe cur = orig;
        for (size_t i = 0; i < ((size_t)1 << (extended_k - k)); i++) {
            t_inv.push_back(f->invert(f->sub(cur, f->one)));
            cur = f->mul(cur, step);
        }
        return true;
    }
    Fe rotate_omega(const Fe& x, int32_t rot) const {
        return f->mul(x, f->pow_u64(rot >= 0 ? omega : omega_inv, (uint64_t)(rot >= 0 ? rot : -rot)));
    }
};
