// capi.hip -- C ABI (include/dehalo.h) over the gfx950 kernels.  Host logic only: argument
// checks, HBM workspace, dispatch to the per-curve / per-field translation units
// (msm_*.hip, ntt_*.hip), HIP-event timing.  No CPU arithmetic path exists here: every field
// or group operation runs on the device, and context creation fails without one.
#include <algorithm>
#include <atomic>

#include <map>

#include "evalh_types.hpp"
#include "internal.hpp"

namespace {

hipStream_t pick_stream(dehalo_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

// Window bits by measurement on MI355X (tools/sweep_c.py, tools/profile_prover.py with WINDOW_BITS): the bucket
// reduction costs ~ 2^(c-1) group operations on a latency chain, the accumulation n * ceil(256 / c) additions.
// 2^20 and up: 16; 2^17 .. 2^19: 15; 2^10 .. 2^16: 13 (prover-shaped schedule at k = 14: 2.9 ms of MSMs with
// c = 13, 3.1 with 15, 3.5 with 14 -- even c leaves a top window of few bits whose buckets are hot).
uint32_t choose_window(size_t n) {
    uint32_t l = log2_ceil(n ? n : 1);
    // 2^20 and up: 17 bits -- 15 rows instead of 16 (6 % fewer additions), 2^16 buckets whose histogram fits the LDS as packed 16-bit counters; four alternating pairs
    // at 2^20: 760.7 -> 779.1 Mpoints/s in the step, one MSM alone 1.55 -> 1.48 ms (profiles/r06_window_17.txt)
    if (l >= 20) return 17;
    if (l >= 19) return 16;      // k = 19 proofs 25.7 -> 25.0 ms against 15 bits (17: 26.4); k = 17 / 18 stay at 15 (7.25 / 12.9 ms against 7.5 / 13.0 at 16): profiles/r06_window_sweep_proofs.txt
    if (l >= 17) return 15;
    if (l >= 10) return 13;
    return std::max<uint32_t>(6, l + 1);
}

// Single-row tables (the one-shot, unregistered path: every window keeps its own buckets and the window sums are combined by
// c (W - 1) doublings): fewer buckets per window pay for the extra windows.  Measured on MI355X (tools/sweep_single_row.py, Pallas,
// uniform scalars, device time): 2^14 c = 10 0.81 ms (13: 0.95, 16: 1.12); 2^17 c = 13 1.15 (10: 1.21, 15: 1.39); 2^20 c = 13 3.13 (16: 3.51).
uint32_t choose_window_single(size_t n) {
    uint32_t l = log2_ceil(n ? n : 1);
    if (l >= 16) return 13;
    if (l >= 12) return 10;
    return choose_window(n);
}

// Windows of the signed-digit recoding: the smallest W for which no canonical scalar s < r leaves a carry after window W - 1
// (msm.cuh for_each_digit drops it).  With top = (r - 1) >> c(W - 1) that holds when top + 1 <= 2^(c-1), and also when top == 2^(c-1)
// exactly while the c bits of r - 1 just below the top window are all zero (then s with that top digit has a zero digit in window
// W - 2, which absorbs any carry).  E.g. BN254 Fr, c = 15: 17 windows instead of ceil(256 / 15) = 18; Pasta Fq, c = 17: 15.
uint32_t signed_windows(const uint32_t r_words[8], uint32_t c) {
    uint64_t r1[4];                                                  // r - 1 (r is odd)
    for (int i = 0; i < 4; i++) r1[i] = (uint64_t)r_words[2 * i] | ((uint64_t)r_words[2 * i + 1] << 32);
    r1[0] -= 1;
    auto bits_at = [&](uint32_t lo, uint32_t count) -> uint64_t {   // bits [lo, lo + count) of r - 1, count <= 32
        uint64_t v = 0;
        for (uint32_t b = 0; b < count; b++) {
            uint32_t pos = lo + b;
            if (pos < 256 && ((r1[pos >> 6] >> (pos & 63)) & 1)) v |= 1ull << b;
        }
        return v;
    };
    for (uint32_t W = (254 + c - 1) / c; W <= (256 + c - 1) / c; W++) {
        if (W < 2) continue;
        const uint32_t shift = c * (W - 1);
        bool above = false;                                          // anything of r - 1 above the top window?
        for (uint32_t pos = shift + c; pos < 256; pos++) above |= ((r1[pos >> 6] >> (pos & 63)) & 1) != 0;
        if (above) continue;
        const uint64_t top = bits_at(shift, c), half = 1ull << (c - 1);
        if (top + 1 <= half) return W;
        if (top == half && bits_at(shift - c, c) == 0) return W;
    }
    return (256 + c - 1) / c;
}
const uint32_t* scalar_modulus_words(int curve) {
    return curve == DEHALO_CURVE_BN254_G1 ? Bn254Fr::P : curve == DEHALO_CURVE_PALLAS ? PastaFq::P : PastaFp::P;
}

int do_msm(dehalo_ctx* ctx, const dehalo_bases* bases, const fe* d_scalars, size_t len, size_t batch, jacobian_t* d_out, hipStream_t s) {
    switch (bases->curve) {
        case DEHALO_CURVE_BN254_G1: return run_msm_bn254(ctx, bases, d_scalars, len, batch, d_out, s);
        case DEHALO_CURVE_PALLAS: return run_msm_pallas(ctx, bases, d_scalars, len, batch, d_out, s);
        case DEHALO_CURVE_VESTA: return run_msm_vesta(ctx, bases, d_scalars, len, batch, d_out, s);
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    }
}
int do_build_table(dehalo_ctx* ctx, dehalo_bases* b, const affine_t* d_std, hipStream_t s) {
    switch (b->curve) {
        case DEHALO_CURVE_BN254_G1: return build_table_bn254(ctx, b, d_std, s);
        case DEHALO_CURVE_PALLAS: return build_table_pallas(ctx, b, d_std, s);
        case DEHALO_CURVE_VESTA: return build_table_vesta(ctx, b, d_std, s);
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    }
}
int do_to_affine(dehalo_ctx* ctx, int curve, const jacobian_t* d_in, affine_t* d_out, uint32_t count, hipStream_t s) {
    switch (curve) {
        case DEHALO_CURVE_BN254_G1: return to_affine_bn254(ctx, d_in, d_out, count, s);
        case DEHALO_CURVE_PALLAS: return to_affine_pallas(ctx, d_in, d_out, count, s);
        case DEHALO_CURVE_VESTA: return to_affine_vesta(ctx, d_in, d_out, count, s);
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    }
}
int do_point_sum(dehalo_ctx* ctx, int curve, const jacobian_t* d_in, uint32_t count, jacobian_t* d_out, hipStream_t s) {
    switch (curve) {
        case DEHALO_CURVE_BN254_G1: return point_sum_bn254(ctx, d_in, count, d_out, s);
        case DEHALO_CURVE_PALLAS: return point_sum_pallas(ctx, d_in, count, d_out, s);
        case DEHALO_CURVE_VESTA: return point_sum_vesta(ctx, d_in, count, d_out, s);
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    }
}
int do_ntt(dehalo_ctx* ctx, int field, const fe* src, uint64_t src_len, uint64_t src_stride, fe* dst, uint64_t dst_stride, uint32_t log_n,
           const uint64_t omega[4], size_t batch, const NttScale& sc, hipStream_t s) {
    switch (field) {
        case DEHALO_FIELD_BN254_FR: return run_ntt_bn254_fr(ctx, src, src_len, src_stride, dst, dst_stride, log_n, omega, batch, sc, s);
        case DEHALO_FIELD_BN254_FQ: return run_ntt_bn254_fq(ctx, src, src_len, src_stride, dst, dst_stride, log_n, omega, batch, sc, s);
        case DEHALO_FIELD_PASTA_FP: return run_ntt_pasta_fp(ctx, src, src_len, src_stride, dst, dst_stride, log_n, omega, batch, sc, s);
        case DEHALO_FIELD_PASTA_FQ: return run_ntt_pasta_fq(ctx, src, src_len, src_stride, dst, dst_stride, log_n, omega, batch, sc, s);
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    }
}
int do_field_op(dehalo_ctx* ctx, int field, int op, const fe* a, const fe* b, fe* out, uint64_t n, hipStream_t s) {
    switch (field) {
        case DEHALO_FIELD_BN254_FR: return field_op_bn254_fr(ctx, op, a, b, out, n, s);
        case DEHALO_FIELD_BN254_FQ: return field_op_bn254_fq(ctx, op, a, b, out, n, s);
        case DEHALO_FIELD_PASTA_FP: return field_op_pasta_fp(ctx, op, a, b, out, n, s);
        case DEHALO_FIELD_PASTA_FQ: return field_op_pasta_fq(ctx, op, a, b, out, n, s);
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    }
}

#define FIELD_SWITCH(ctx, field, CALL)                                                  \
    switch (field) {                                                                    \
        case DEHALO_FIELD_BN254_FR: return CALL(bn254_fr);                              \
        case DEHALO_FIELD_BN254_FQ: return CALL(bn254_fq);                              \
        case DEHALO_FIELD_PASTA_FP: return CALL(pasta_fp);                              \
        case DEHALO_FIELD_PASTA_FQ: return CALL(pasta_fq);                              \
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");           \
    }
int do_eval_poly(dehalo_ctx* ctx, int field, const fe* c, uint64_t len, uint64_t stride, size_t batch, const uint64_t pt[4], fe* out, hipStream_t s) {
#define CALL(N) eval_poly_##N(ctx, c, len, stride, batch, pt, out, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_eval_poly_multi(dehalo_ctx* ctx, int field, const fe* const* polys, size_t count, uint64_t len, const uint64_t* pts, uint32_t npts, fe* out, hipStream_t s,
                       const uint8_t* masks = nullptr) {
#define CALL(N) eval_poly_multi_##N(ctx, polys, count, len, pts, npts, out, s, masks)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_batch_invert(dehalo_ctx* ctx, int field, fe* v, uint64_t len, hipStream_t s) {
#define CALL(N) batch_invert_##N(ctx, v, len, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_prefix_product(dehalo_ctx* ctx, int field, const fe* in, uint64_t len, fe* out, hipStream_t s) {
#define CALL(N) prefix_product_##N(ctx, in, len, out, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_grand_product(dehalo_ctx* ctx, int field, const fe* num, const fe* den, uint64_t len, size_t batch, uint64_t stride, fe* z, hipStream_t s) {
#define CALL(N) grand_product_##N(ctx, num, den, len, batch, stride, z, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}

int do_lincomb(dehalo_ctx* ctx, int field, const fe* const* cols, const uint64_t* coefs, size_t count, uint64_t len, fe* out, const uint64_t* sub0, hipStream_t s) {
#define CALL(N) lincomb_##N(ctx, cols, coefs, count, len, out, sub0, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_scale(dehalo_ctx* ctx, int field, fe* a, uint64_t len, const uint64_t* pattern, uint32_t period, const fe* d_factor, hipStream_t s) {
#define CALL(N) scale_##N(ctx, a, len, pattern, period, d_factor, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_kate_division(dehalo_ctx* ctx, int field, const fe* a, uint64_t len, const uint64_t pt[4], fe* q, hipStream_t s) {
#define CALL(N) kate_division_##N(ctx, a, len, pt, q, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}

int do_kate_division_batch(dehalo_ctx* ctx, int field, const fe* const* a, uint64_t len, const uint64_t* pts, fe* const* q, size_t count, hipStream_t s) {
#define CALL(N) kate_division_batch_##N(ctx, a, len, pts, q, count, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}

int do_convert_form(dehalo_ctx* ctx, int field, const fe* in, fe* out, uint64_t n, int to_internal, hipStream_t s) {
#define CALL(N) convert_form_##N(ctx, in, out, n, to_internal, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_graph_upload(dehalo_ctx* ctx, int field, dehalo_graph* g, const uint64_t* constants, hipStream_t s) {
#define CALL(N) graph_upload_##N(ctx, g, constants, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_graph_evaluate(dehalo_ctx* ctx, const dehalo_graph* g, const dehalo_eval_inputs* in, uint32_t log_rows, uint32_t rot_scale, const fe* prev, fe* out,
                      hipStream_t s) {
#define CALL(N) graph_evaluate_##N(ctx, g, in, log_rows, rot_scale, prev, out, s)
    FIELD_SWITCH(ctx, g->field, CALL)
#undef CALL
}
int do_perm_h(dehalo_ctx* ctx, int field, const dehalo_perm_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s) {
#define CALL(N) perm_h_##N(ctx, in, log_rows, rot_scale, v, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}
int do_lookup_h(dehalo_ctx* ctx, int field, const dehalo_lookup_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s) {
#define CALL(N) lookup_h_##N(ctx, in, log_rows, rot_scale, v, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}

int do_lookup_h_batch(dehalo_ctx* ctx, int field, const dehalo_lookup_inputs* in, uint32_t count, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s) {
#define CALL(N) lookup_h_batch_##N(ctx, in, count, log_rows, rot_scale, v, s)
    FIELD_SWITCH(ctx, field, CALL)
#undef CALL
}

// Host-side compilation of upstream's GraphEvaluator into the device program: sources are
// resolved to table indices, and every intermediate gets a slot from a liveness scan (a slot is
// reused as soon as its value has been read for the last time; the first EVH_MAX_LDS_SLOTS slots
// live in LDS, the rest in an HBM scratch column).
int compile_graph(dehalo_ctx* ctx, dehalo_graph* g, const int32_t* rotations, uint32_t num_rotations, const dehalo_calculation* calcs, uint32_t num_calcs,
                  const dehalo_source* parts, uint32_t num_parts, uint32_t num_intermediates, std::vector<DevCalc>& out_calcs, std::vector<DevSrc>& out_parts,
                  bool propagate = true) {
    const uint32_t NEVER = 0xffffffffu;
    if (propagate) {
        // Copy propagation.  Upstream's GraphEvaluator::add_expression wraps EVERY column query in Calculation::Store(source) so that its
        // CPU loop loads a cell once; on the device a Store is a calculation of its own (fetch, slot write) and every later use a slot
        // read, while reading the column directly costs the same fetch.  A Store of a non-intermediate source whose target is written
        // exactly once is therefore dropped and its uses read the source (MainGate's gate: 34 calculations / 7 slots -> 20 / 3; fewer
        // slots is more resident waves: the kernel's occupancy is bounded by its LDS slots).  The original program is validated first.
        {
            std::vector<DevCalc> tc;
            std::vector<DevSrc> tp;
            dehalo_graph probe = *g;
            TRY(compile_graph(ctx, &probe, rotations, num_rotations, calcs, num_calcs, parts, num_parts, num_intermediates, tc, tp, false));
        }
        std::vector<uint32_t> defs(num_intermediates, 0);
        for (uint32_t i = 0; i < num_calcs; i++) defs[calcs[i].target]++;
        std::vector<int> aliased(num_intermediates, 0);
        std::vector<dehalo_source> alias(num_intermediates);
        std::vector<dehalo_calculation> cc;
        std::vector<dehalo_source> pp(parts, parts + num_parts);
        auto sub = [&](dehalo_source s) { return (s.kind == DEHALO_SRC_INTERMEDIATE && aliased[s.index]) ? alias[s.index] : s; };
        for (uint32_t i = 0; i < num_calcs; i++) {
            dehalo_calculation c = calcs[i];
            c.a = sub(c.a);
            c.b = sub(c.b);
            if (c.op == DEHALO_CALC_HORNER)
                for (uint32_t k = 0; k < c.parts_len; k++) pp[c.parts_begin + k] = sub(pp[c.parts_begin + k]);
            if (c.op == DEHALO_CALC_STORE && c.a.kind != DEHALO_SRC_INTERMEDIATE && i + 1 != num_calcs && defs[c.target] == 1) {
                aliased[c.target] = 1;
                alias[c.target] = c.a;
                continue;
            }
            cc.push_back(c);
        }
        g->num_calcs = (uint32_t)cc.size();
        return compile_graph(ctx, g, rotations, num_rotations, cc.data(), (uint32_t)cc.size(), pp.data(), num_parts, num_intermediates, out_calcs, out_parts, false);
    }
    std::vector<uint32_t> last_use(num_intermediates, NEVER), first_def(num_intermediates, NEVER);
    auto check_src = [&](const dehalo_source& src, uint32_t at) -> int {
        switch (src.kind) {
            case DEHALO_SRC_CONSTANT: if (src.index >= g->num_constants) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: constant index out of range"); break;
            case DEHALO_SRC_INTERMEDIATE:
                if (src.index >= num_intermediates) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: intermediate index out of range");
                if (first_def[src.index] == NEVER || first_def[src.index] >= at) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: intermediate read before it is written");
                last_use[src.index] = at;
                break;
            case DEHALO_SRC_FIXED: case DEHALO_SRC_ADVICE: case DEHALO_SRC_INSTANCE:
                if (src.rotation >= num_rotations) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: rotation index out of range");
                break;
            case DEHALO_SRC_CHALLENGE: case DEHALO_SRC_BETA: case DEHALO_SRC_GAMMA: case DEHALO_SRC_THETA: case DEHALO_SRC_Y: case DEHALO_SRC_PREVIOUS: break;
            default: return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: unknown value source kind");
        }
        return 0;
    };
    auto binary = [](uint32_t op) { return op == DEHALO_CALC_ADD || op == DEHALO_CALC_SUB || op == DEHALO_CALC_MUL || op == DEHALO_CALC_HORNER; };
    // pass 1: validation, definitions and last uses
    for (uint32_t i = 0; i < num_calcs; i++) {
        const dehalo_calculation& c = calcs[i];
        if (c.op > DEHALO_CALC_STORE) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: unknown calculation");
        if (c.target >= num_intermediates) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: target out of range");
        TRY(check_src(c.a, i));
        if (binary(c.op)) TRY(check_src(c.b, i));
        if (c.op == DEHALO_CALC_HORNER) {
            if ((uint64_t)c.parts_begin + c.parts_len > num_parts) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph: horner parts out of range");
            for (uint32_t k = 0; k < c.parts_len; k++) TRY(check_src(parts[c.parts_begin + k], i));
        }
        if (first_def[c.target] == NEVER) first_def[c.target] = i;
    }
    if (num_calcs) last_use[calcs[num_calcs - 1].target] = num_calcs;   // the result stays live to the end
    // pass 2: slots
    std::vector<uint32_t> slot_of(num_intermediates, NEVER), free_slots;
    uint32_t next_slot = 0;
    auto conv = [&](const dehalo_source& src) {
        DevSrc d{EVS_SCALAR, 0, 0};
        switch (src.kind) {
            case DEHALO_SRC_BETA: d.index = 0; break;
            case DEHALO_SRC_GAMMA: d.index = 1; break;
            case DEHALO_SRC_THETA: d.index = 2; break;
            case DEHALO_SRC_Y: d.index = 3; break;
            case DEHALO_SRC_CONSTANT: d.index = 4 + src.index; break;
            case DEHALO_SRC_CHALLENGE: d.index = 4 + g->num_constants + src.index; g->max_challenge = std::max(g->max_challenge, src.index + 1); break;
            case DEHALO_SRC_INTERMEDIATE: {
                uint32_t sl = slot_of[src.index];
                d.kind = sl < EVH_MAX_LDS_SLOTS ? EVS_SLOT_LDS : EVS_SLOT_HBM;
                d.index = sl < EVH_MAX_LDS_SLOTS ? sl : sl - EVH_MAX_LDS_SLOTS;
            } break;
            case DEHALO_SRC_FIXED: d.kind = EVS_FIXED; d.index = src.index; d.rot = rotations[src.rotation]; g->max_fixed = std::max(g->max_fixed, src.index + 1); break;
            case DEHALO_SRC_ADVICE: d.kind = EVS_ADVICE; d.index = src.index; d.rot = rotations[src.rotation]; g->max_advice = std::max(g->max_advice, src.index + 1); break;
            case DEHALO_SRC_INSTANCE: d.kind = EVS_INSTANCE; d.index = src.index; d.rot = rotations[src.rotation]; g->max_instance = std::max(g->max_instance, src.index + 1); break;
            default: d.kind = EVS_PREVIOUS; g->uses_previous = true; break;
        }
        return d;
    };
    std::vector<std::vector<uint32_t>> dying(num_calcs + 1);
    for (uint32_t v = 0; v < num_intermediates; v++)
        if (last_use[v] != NEVER) dying[last_use[v]].push_back(v);
    out_calcs.resize(num_calcs);
    out_parts.resize(num_parts ? num_parts : 1);
    for (uint32_t i = 0; i < num_calcs; i++) {
        const dehalo_calculation& c = calcs[i];
        DevCalc d{};
        d.op = c.op;
        d.a = conv(c.a);
        if (binary(c.op)) d.b = conv(c.b);
        d.parts_begin = c.parts_begin; d.parts_len = c.op == DEHALO_CALC_HORNER ? c.parts_len : 0;
        for (uint32_t k = 0; k < d.parts_len; k++) out_parts[c.parts_begin + k] = conv(parts[c.parts_begin + k]);
        // operands are all read before the result is written: slots whose last use is this calculation are free for its target
        for (uint32_t v : dying[i])
            if (slot_of[v] != NEVER && v != c.target) { free_slots.push_back(slot_of[v]); slot_of[v] = NEVER; }
        if (slot_of[c.target] == NEVER) {
            if (!free_slots.empty()) {
                auto it = std::min_element(free_slots.begin(), free_slots.end());   // lowest slot first: LDS before HBM
                slot_of[c.target] = *it;
                free_slots.erase(it);
            } else slot_of[c.target] = next_slot++;
        }
        uint32_t sl = slot_of[c.target];
        d.target_kind = sl < EVH_MAX_LDS_SLOTS ? EVS_SLOT_LDS : EVS_SLOT_HBM;
        d.target_slot = sl < EVH_MAX_LDS_SLOTS ? sl : sl - EVH_MAX_LDS_SLOTS;
        if (last_use[c.target] == NEVER || last_use[c.target] <= i) { /* dead value: its slot is released at once */
            if (!(i + 1 == num_calcs)) { free_slots.push_back(sl); slot_of[c.target] = NEVER; }
        }
        out_calcs[i] = d;
    }
    g->lds_slots = std::min<uint32_t>(next_slot, EVH_MAX_LDS_SLOTS);
    g->hbm_slots = next_slot > EVH_MAX_LDS_SLOTS ? next_slot - EVH_MAX_LDS_SLOTS : 0;
    if (num_calcs) {
        dehalo_source r{DEHALO_SRC_INTERMEDIATE, calcs[num_calcs - 1].target, 0};
        g->result = conv(r);
    }
    return 0;
}

int register_impl(dehalo_ctx* ctx, int curve, const uint64_t* affine_xy, size_t n, size_t stride_bytes, int window_bits, int precompute,
                  dehalo_bases** out, bool on_device = false) {
    if (!affine_xy || !out || n == 0 || stride_bytes < 64 || n >= (1ull << 30)) return dh_fail(ctx, DEHALO_ERR_INVALID, "bases_register: bad argument");
    if (window_bits != 0 && (window_bits < 4 || window_bits > (precompute ? 17 : 16)))
        return dh_fail(ctx, DEHALO_ERR_INVALID, "window_bits must be 0 or in [4, 16] (17 with precomputed rows)");
    uint32_t c = window_bits ? (uint32_t)window_bits : (precompute ? choose_window(n) : choose_window_single(n));
    if (!window_bits && precompute) {      // DEHALO_WINDOW_BITS: tuning experiments (results never depend on the window)
        const char* e = DH_EXPERIMENT_ENV("DEHALO_WINDOW_BITS");
        if (e && atoi(e) >= 4 && atoi(e) <= 17) c = (uint32_t)atoi(e);
    }
    if (c < 4) c = 4;
    uint32_t W = signed_windows(scalar_modulus_words(curve), c);
    if (precompute && (uint64_t)n * W >= (1ull << 30)) return dh_fail(ctx, DEHALO_ERR_INVALID, "precomputed table too large");      // 30-bit table indices in the sorted list (msm.cuh)
    // stage the caller's points (standard Montgomery form) on the device, then build the table
    if (!on_device) TRY(dh_ensure(ctx, ctx->ws_tmp_bases, n * sizeof(affine_t)));
    dehalo_bases* b = new dehalo_bases();
    b->curve = curve; b->n = n; b->c = c; b->W = W; b->precomp = precompute ? 1 : 0; b->table = nullptr;
    size_t rows = precompute ? W : 1;
    hipError_t e = hipMalloc((void**)&b->table, rows * n * sizeof(affine_t));
    if (e != hipSuccess) { delete b; return dh_fail(ctx, DEHALO_ERR_OOM, std::string("bases table: ") + hipGetErrorString(e)); }
    // 64 MiB of SRS points at 2^20: DMA straight from the caller's pages -- only when these bytes are what is copied (host memory, contiguous points)
    HostPin pin_bases(on_device || stride_bytes != 64 ? nullptr : affine_xy, n * stride_bytes);
    // (contiguous points: a plain copy -- the 2-D path took 3 of the 4.1 ms of registering 2^20 points)
    int rc = 0;
    std::vector<uint64_t> packed;      // (points with a trailing flag byte: gathered on the host, so that the upload is one contiguous copy)
    if (stride_bytes != 64 && !on_device) {
        packed.resize(n * 8);
        for (size_t i = 0; i < n; i++) memcpy(&packed[i * 8], (const char*)affine_xy + i * stride_bytes, 64);
    }
    if (!on_device) rc = dh_h2d(ctx, ctx->ws_tmp_bases.p, stride_bytes == 64 ? (const void*)affine_xy : (const void*)packed.data(), n * 64, ctx->stream);
    if (rc != 0) { (void)hipFree(b->table); delete b; return rc; }
    if (e == hipSuccess) rc = do_build_table(ctx, b, on_device ? (const affine_t*)affine_xy : (const affine_t*)ctx->ws_tmp_bases.p, ctx->stream);
    if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || rc != 0) {
        (void)hipFree(b->table);
        delete b;
        return rc ? rc : dh_fail(ctx, DEHALO_ERR_HIP, std::string("bases upload: ") + hipGetErrorString(e));
    }
    *out = b;
    return 0;
}

}  // namespace

// the limit of a precomputed table registered with window_bits = 0: n x windows < 2^30 (30-bit table indices in the sorted list, msm.cuh)
bool dh_precomputed_table_fits(int curve, size_t n) {
    return n < (1ull << 30) && (uint64_t)n * signed_windows(scalar_modulus_words(curve), std::max<uint32_t>(4, choose_window(n))) < (1ull << 30);
}

// ==========================================================================================
extern "C" {

const char* dehalo_version(void) { return "dehalo 0.2 gfx950"; }

int dehalo_ctx_create(int device, dehalo_ctx** out) { return dehalo_ctx_create_with_priority(device, 0, out); }

int dehalo_ctx_create_with_priority(int device, int priority, dehalo_ctx** out) {
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
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {   // priority > 0: the device's highest stream priority (a context of small kernels beside another context's long ones), < 0: lowest
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int prio = priority > 0 ? greatest : (priority < 0 ? least : 0);
        // Experiment (DEHALO_CU_PARTITION = P, measurements only): the i-th context created by this process gets a stream that may only use the (i mod P)-th
        // P-th of the compute units (hipExtStreamCreateWithCUMask), so that the provers of a batch do not wait for each other's workgroups to leave a CU.
        static const int cu_parts = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_CU_PARTITION"); return e ? atoi(e) : 0; }();
        static std::atomic<int> cu_next{0};
        hipError_t e;
        if (cu_parts > 1 && cu_parts <= 16) {
            const int part = cu_next.fetch_add(1) % cu_parts, ncu = ctx->num_cus, per = ncu / cu_parts;
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            static const bool interleave = DH_EXPERIMENT_ENV("DEHALO_CU_PARTITION_INTERLEAVE") != nullptr;
            for (int cu = 0; cu < ncu; cu++) {
                const bool mine = interleave ? (cu % cu_parts) == part : (cu / per) == part;
                if (mine) mask[cu / 32] |= 1u << (cu % 32);
            }
            e = hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)mask.size(), mask.data());
            if (e == hipSuccess) ctx->num_cus = per;
        } else
        e = priority == 0 ? hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)
                          : hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio);
        if (e != hipSuccess) { delete ctx; return DEHALO_ERR_HIP; }
    }
    if (const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_ACC_MIN_LAYERS")) ctx->msm_acc_min_layers = std::max(1, std::min(4, atoi(e)));
    if (const char* e = DH_EXPERIMENT_ENV("DEHALO_HOST_SPIN_US")) ctx->host_wait_spin_us = std::max(0, std::min(1000000, atoi(e)));
    if (const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_ACC_BLOCK")) ctx->msm_acc_block = atoi(e) == 768 ? 768 : 128;                                      // launch geometry only
    if (const char* e = DH_EXPERIMENT_ENV("DEHALO_MSM_ACC_POINTS")) ctx->msm_acc_points = std::max(0, std::min(4096, atoi(e)));   // launch geometry only (dehalo_ctx_set_tuning)
    *out = ctx;
    return 0;
}

void dehalo_ctx_destroy(dehalo_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    DevBuf* bufs[] = {&ctx->ws_scalars, &ctx->ws_out, &ctx->ws_count, &ctx->ws_counters, &ctx->ws_off, &ctx->ws_records, &ctx->ws_merge_lists, &ctx->ws_merge_parts,
                      &ctx->ws_bhist, &ctx->ws_pcount, &ctx->ws_pairs, &ctx->ws_bsum, &ctx->ws_idx, &ctx->ws_partial0, &ctx->ws_buckets, &ctx->ws_contrib, &ctx->ws_tree, &ctx->ws_bred_cnt,
                      &ctx->ws_gsums, &ctx->ws_ntt_scratch, &ctx->ws_ntt_io, &ctx->ws_ntt_io2, &ctx->ws_fop[0], &ctx->ws_fop[1], &ctx->ws_fop[2],
                      &ctx->ws_tmp_bases, &ctx->ws_poly[0], &ctx->ws_poly[1], &ctx->ws_poly[2], &ctx->ws_poly[3], &ctx->ws_poly[4], &ctx->ws_poly_io[0], &ctx->ws_poly_io[1],
                      &ctx->ws_poly_io[2], &ctx->ws_evh[0], &ctx->ws_evh[1], &ctx->ws_evh[2], &ctx->ws_evh[3], &ctx->ws_lookup};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& t : ctx->twiddles) (void)hipFree(t.tw);
    for (auto& r : ctx->regions) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (int i = 0; i < 2; i++) {
        if (ctx->stage.buf[i]) (void)hipHostFree(ctx->stage.buf[i]);
        if (ctx->stage.ev[i]) (void)hipEventDestroy(ctx->stage.ev[i]);
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* dehalo_last_error(const dehalo_ctx* ctx) {
    if (!ctx) return "null context";
    // a copy per calling thread, taken under the lock: another thread's error path may replace ctx->err at any time
    static thread_local std::string copy;
    {
        std::lock_guard<std::mutex> lk(const_cast<dehalo_ctx*>(ctx)->err_mu);
        copy = ctx->err;
    }
    return copy.c_str();
}

int dehalo_ctx_set_tuning(dehalo_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return DEHALO_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!strcmp(key, "msm_acc_points")) {
        if (value < 0 || value > 4096) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm_acc_points must be in [0, 4096]");
        ctx->msm_acc_points = value;
        return 0;
    }
    if (!strcmp(key, "msm_acc_waves")) {
        if (value < 1 || value > 4) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm_acc_waves must be in [1, 4]");
        ctx->msm_acc_waves = value;
        return 0;
    }
    if (!strcmp(key, "host_wait_spin_us")) {
        if (value < 0 || value > 1000000) return dh_fail(ctx, DEHALO_ERR_INVALID, "host_wait_spin_us must be in [0, 1000000]");
        ctx->host_wait_spin_us = value;
        return 0;
    }
    if (!strcmp(key, "msm_sort_block")) {
        if (value != 512 && value != 1024) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm_sort_block must be 512 or 1024");
        ctx->msm_sort_block = value;
        return 0;
    }
    if (!strcmp(key, "msm_acc_block")) {
        if (value != 128 && value != 768) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm_acc_block must be 128 or 768");
        ctx->msm_acc_block = value;
        return 0;
    }
    if (!strcmp(key, "ntt_full_table_log")) {
        if (value < 0 || value > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "ntt_full_table_log must be in [0, 30]");
        ctx->ntt_full_table_log = value;
        return 0;
    }
    return dh_fail(ctx, DEHALO_ERR_INVALID, std::string("unknown tuning key: ") + key);
}

void* dehalo_ctx_stream(dehalo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int dehalo_ctx_synchronize(dehalo_ctx* ctx) {
    if (!ctx) return DEHALO_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_download(dehalo_ctx* ctx, const void* d_src, size_t bytes, void* host_dst) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_src || !host_dst) && bytes) return dh_fail(ctx, DEHALO_ERR_INVALID, "download: null argument");
    (void)hipSetDevice(ctx->device);
    if (bytes) TRY(dh_d2h(ctx, host_dst, d_src, bytes, ctx->stream));
    HIP_TRY(ctx, dh_stream_wait(ctx, ctx->stream));
    return 0;
}

int dehalo_upload(dehalo_ctx* ctx, const void* host_src, size_t bytes, void* d_dst) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!host_src || !d_dst) && bytes) return dh_fail(ctx, DEHALO_ERR_INVALID, "upload: null argument");
    (void)hipSetDevice(ctx->device);
    {
        HostPin pin(host_src, bytes);      // 4 MiB and more: one DMA from the caller's pages, released below, after the stream has drained
        TRY(dh_h2d(ctx, d_dst, host_src, bytes, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int dehalo_bases_register(dehalo_ctx* ctx, int curve, const uint64_t* affine_xy, size_t n, size_t stride_bytes, int window_bits, int precompute,
                          dehalo_bases** out) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (curve < 0 || curve > 2) return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return register_impl(ctx, curve, affine_xy, n, stride_bytes, window_bits, precompute, out);
}

int dehalo_bases_register_device(dehalo_ctx* ctx, int curve, const uint64_t* d_affine_xy, size_t n, int window_bits, int precompute, dehalo_bases** out) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (curve < 0 || curve > 2) return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return register_impl(ctx, curve, d_affine_xy, n, 64, window_bits, precompute, out, true);
}

int dehalo_bases_release(dehalo_ctx* ctx, dehalo_bases* bases) {
    if (!ctx || !bases) return DEHALO_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(bases->table);
    delete bases;
    return 0;
}

size_t dehalo_bases_len(const dehalo_bases* bases) { return bases ? bases->n : 0; }

int dehalo_bases_info(const dehalo_bases* bases, uint32_t* window_bits, uint32_t* windows, int* precomputed) {
    if (!bases) return DEHALO_ERR_INVALID;
    if (window_bits) *window_bits = bases->c;
    if (windows) *windows = bases->W;
    if (precomputed) *precomputed = bases->precomp;
    return 0;
}

int dehalo_msm_device(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* d_scalars, size_t len, size_t batch, uint64_t* d_out_jacobian,
                      void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!bases || (!d_scalars && len) || !d_out_jacobian) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: null argument");
    if (len > bases->n) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: more scalars than registered bases");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_msm(ctx, bases, (const fe*)d_scalars, len, batch, (jacobian_t*)d_out_jacobian, pick_stream(ctx, stream));
}

int dehalo_msm_device_affine(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* d_scalars, size_t len, size_t batch, uint64_t* d_out_jacobian,
                             uint64_t* d_out_affine, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!bases || (!d_scalars && len) || !d_out_affine) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: null argument");
    if (len > bases->n) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: more scalars than registered bases");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    hipStream_t s = pick_stream(ctx, stream);
    if (len == 0 || batch == 0) {   // the empty sum: identity = (0, 0)
        if (batch) HIP_TRY(ctx, hipMemsetAsync(d_out_affine, 0, batch * sizeof(affine_t), s));
        if (batch && d_out_jacobian) HIP_TRY(ctx, hipMemsetAsync(d_out_jacobian, 0, batch * sizeof(jacobian_t), s));
        return 0;
    }
    ctx->msm_affine_out = (affine_t*)d_out_affine;
    // (without a Jacobian destination the kernel skips that form; run_msm_t only passes the pointer on)
    int rc = do_msm(ctx, bases, (const fe*)d_scalars, len, batch, (jacobian_t*)d_out_jacobian, s);
    ctx->msm_affine_out = nullptr;
    return rc;
}

int dehalo_msm_batch(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* const* scalars, size_t len, size_t batch, uint64_t* out_jacobian) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!bases || !scalars || !out_jacobian) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: null argument");
    if (len > bases->n) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: more scalars than registered bases");
    if (batch && len > (SIZE_MAX / 32) / batch) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: batch * len overflows");
    for (size_t b = 0; b < batch; b++)
        if (!scalars[b] && len) return dh_fail(ctx, DEHALO_ERR_INVALID, "msm: null scalar column");
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);   // one critical section per host-buffer call: staging, kernels, download
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        TRY(dh_ensure(ctx, ctx->ws_scalars, std::max<size_t>(32, batch * len * 32)));
        TRY(dh_ensure(ctx, ctx->ws_out, std::max<size_t>(96, batch * 96)));
        for (size_t b = 0; b < batch && len; b++) {
            HostPin pin(scalars[b], len * 32);
            TRY(dh_h2d(ctx, (char*)ctx->ws_scalars.p + b * len * 32, scalars[b], len * 32, ctx->stream));
            if (pin.p) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the pin ends with this scope
        }
    }
    TRY(dehalo_msm_device(ctx, bases, (const uint64_t*)ctx->ws_scalars.p, len, batch, (uint64_t*)ctx->ws_out.p, nullptr));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(dh_d2h(ctx, out_jacobian, ctx->ws_out.p, batch * 96, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_msm(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* scalars, size_t len, uint64_t out_jacobian[12]) {
    const uint64_t* cols[1] = {scalars};
    return dehalo_msm_batch(ctx, bases, cols, len, 1, out_jacobian);
}

int dehalo_best_multiexp(dehalo_ctx* ctx, int curve, const uint64_t* scalars, const uint64_t* affine_xy, size_t len, uint64_t out_jacobian[12]) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!out_jacobian || ((!scalars || !affine_xy) && len)) return dh_fail(ctx, DEHALO_ERR_INVALID, "best_multiexp: null argument");
    if (curve < 0 || curve > 2) return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown curve id");
    if (len == 0) { memset(out_jacobian, 0, 96); return 0; }
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);
    dehalo_bases* b = nullptr;
    TRY(dehalo_bases_register(ctx, curve, affine_xy, len, 64, 0, 0, &b));
    int rc = dehalo_msm(ctx, b, scalars, len, out_jacobian);
    dehalo_bases_release(ctx, b);
    return rc;
}

int dehalo_point_sum_device(dehalo_ctx* ctx, int curve, const uint64_t* d_jacobian, size_t count, uint64_t* d_out_jacobian, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_jacobian && count) || !d_out_jacobian) return dh_fail(ctx, DEHALO_ERR_INVALID, "point_sum: null argument");
    if (count >= (1ull << 31)) return dh_fail(ctx, DEHALO_ERR_INVALID, "point_sum: too many points");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_point_sum(ctx, curve, (const jacobian_t*)d_jacobian, (uint32_t)count, (jacobian_t*)d_out_jacobian, pick_stream(ctx, stream));
}

int dehalo_to_affine_device(dehalo_ctx* ctx, int curve, const uint64_t* d_jacobian, size_t count, uint64_t* d_affine_xy, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_jacobian || !d_affine_xy) && count) return dh_fail(ctx, DEHALO_ERR_INVALID, "to_affine: null argument");
    if (count == 0) return 0;
    if (count >= (1ull << 31)) return dh_fail(ctx, DEHALO_ERR_INVALID, "to_affine: too many points");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_to_affine(ctx, curve, (const jacobian_t*)d_jacobian, (affine_t*)d_affine_xy, (uint32_t)count, pick_stream(ctx, stream));
}

int dehalo_to_affine(dehalo_ctx* ctx, int curve, const uint64_t* jacobian, size_t count, uint64_t* affine_xy) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!jacobian || !affine_xy) && count) return dh_fail(ctx, DEHALO_ERR_INVALID, "to_affine: null argument");
    if (count == 0) return 0;
    if (count >= (1ull << 31)) return dh_fail(ctx, DEHALO_ERR_INVALID, "to_affine: too many points");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    TRY(dh_ensure(ctx, ctx->ws_fop[0], count * 96));
    TRY(dh_ensure(ctx, ctx->ws_fop[1], count * 64));
    TRY(dh_h2d(ctx, ctx->ws_fop[0].p, jacobian, count * 96, ctx->stream));
    TRY(do_to_affine(ctx, curve, (const jacobian_t*)ctx->ws_fop[0].p, (affine_t*)ctx->ws_fop[1].p, (uint32_t)count, ctx->stream));
    TRY(dh_d2h(ctx, affine_xy, ctx->ws_fop[1].p, count * 64, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- NTT family -----------------------------------------------------------------------------
static int ntt_device_impl(dehalo_ctx* ctx, int field, const uint64_t* d_src, uint64_t src_len, uint64_t src_stride, uint64_t* d_dst,
                           uint64_t dst_stride, uint32_t log_n, const uint64_t omega[4], size_t batch, const NttScale& sc, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!d_src || !d_dst || !omega) return dh_fail(ctx, DEHALO_ERR_INVALID, "ntt: null argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_ntt(ctx, field, (const fe*)d_src, src_len, src_stride, (fe*)d_dst, dst_stride, log_n, omega, batch, sc, pick_stream(ctx, stream));
}

int dehalo_ntt_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_n, const uint64_t omega[4], size_t batch, void* stream) {
    NttScale sc;
    uint64_t N = 1ull << (log_n & 63);
    return ntt_device_impl(ctx, field, d_a, N, N, d_a, N, log_n, omega, batch, sc, stream);
}

int dehalo_intt_scaled_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_n, const uint64_t omega_inv[4], const uint64_t n_inv[4],
                              size_t batch, void* stream) {
    if (!n_inv) return dh_fail(ctx, DEHALO_ERR_INVALID, "intt: null n_inv");
    NttScale sc;
    sc.post_mode = 1; sc.post0 = fe_from_u64(n_inv);
    uint64_t N = 1ull << (log_n & 63);
    return ntt_device_impl(ctx, field, d_a, N, N, d_a, N, log_n, omega_inv, batch, sc, stream);
}

int dehalo_lagrange_to_coeff_device(dehalo_ctx* ctx, int field, const uint64_t* d_values, uint64_t* d_coeffs, uint32_t log_n, const uint64_t omega_inv[4],
                                    const uint64_t n_inv[4], size_t batch, void* stream) {
    if (!n_inv) return dh_fail(ctx, DEHALO_ERR_INVALID, "lagrange_to_coeff: null n_inv");
    NttScale sc;
    sc.post_mode = 1; sc.post0 = fe_from_u64(n_inv);
    uint64_t N = 1ull << (log_n & 63);
    return ntt_device_impl(ctx, field, d_values, N, N, d_coeffs, N, log_n, omega_inv, batch, sc, stream);
}

static int form_shift_of(uint32_t flags) {   // OUT_INTERNAL: x 2^5; IN_INTERNAL: x 2^-5; both: no change of scale
    return (flags & DEHALO_FORM_OUT_INTERNAL ? 1 : 0) - (flags & DEHALO_FORM_IN_INTERNAL ? 1 : 0);
}

int dehalo_coset_ntt_device(dehalo_ctx* ctx, int field, const uint64_t* d_coeffs, uint32_t log_n, uint64_t* d_ext_out, uint32_t log_ext,
                            const uint64_t omega_ext[4], const uint64_t zeta[4], size_t batch, void* stream) {
    return dehalo_coset_ntt_form_device(ctx, field, d_coeffs, log_n, d_ext_out, log_ext, omega_ext, zeta, batch, 0, stream);
}

int dehalo_coset_ntt_form_device(dehalo_ctx* ctx, int field, const uint64_t* d_coeffs, uint32_t log_n, uint64_t* d_ext_out, uint32_t log_ext,
                                 const uint64_t omega_ext[4], const uint64_t zeta[4], size_t batch, uint32_t form_flags, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!zeta || log_ext < log_n) return dh_fail(ctx, DEHALO_ERR_INVALID, "coset_ntt: bad argument");
    if (d_coeffs == d_ext_out) return dh_fail(ctx, DEHALO_ERR_INVALID, "coset_ntt: coeffs and ext_out may not alias");
    NttScale sc;
    sc.form_shift = form_shift_of(form_flags);
    sc.pre_mode = 1; sc.pre_z = fe_from_u64(zeta);  // zeta^2 is formed inside the kernel
    uint64_t n = 1ull << (log_n & 63), N = 1ull << (log_ext & 63);
    return ntt_device_impl(ctx, field, d_coeffs, n, n, d_ext_out, N, log_ext, omega_ext, batch, sc, stream);
}

int dehalo_coset_intt_form_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_ext, const uint64_t omega_ext_inv[4],
                                  const uint64_t ext_n_inv[4], const uint64_t zeta[4], size_t batch, uint32_t form_flags, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!zeta || !ext_n_inv) return dh_fail(ctx, DEHALO_ERR_INVALID, "coset_intt: null argument");
    NttScale sc;
    sc.post_mode = 2; sc.post0 = fe_from_u64(ext_n_inv); sc.post_z = fe_from_u64(zeta);
    sc.form_shift = form_shift_of(form_flags);
    uint64_t N = 1ull << (log_ext & 63);
    return ntt_device_impl(ctx, field, d_a, N, N, d_a, N, log_ext, omega_ext_inv, batch, sc, stream);
}

int dehalo_coset_intt_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_ext, const uint64_t omega_ext_inv[4],
                             const uint64_t ext_n_inv[4], const uint64_t zeta[4], size_t batch, void* stream) {
    return dehalo_coset_intt_form_device(ctx, field, d_a, log_ext, omega_ext_inv, ext_n_inv, zeta, batch, 0, stream);
}

int dehalo_convert_form_device(dehalo_ctx* ctx, int field, const uint64_t* d_in, uint64_t* d_out, size_t n, int to_internal, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_in || !d_out) && n) return dh_fail(ctx, DEHALO_ERR_INVALID, "convert_form: null argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_convert_form(ctx, field, (const fe*)d_in, (fe*)d_out, n, to_internal, pick_stream(ctx, stream));
}

// host-buffer forms: upload, transform, download
static int with_host_io(dehalo_ctx* ctx, const uint64_t* in, size_t in_elems, uint64_t* out, size_t out_elems, bool inplace,
                        int (*fn)(dehalo_ctx*, uint64_t*, uint64_t*, void*), void* arg) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!in || !out) return dh_fail(ctx, DEHALO_ERR_INVALID, "null buffer");
    HostPin pin_in(in, in_elems * 32), pin_out(out == in ? nullptr : out, out_elems * 32);
    uint64_t *d_in, *d_out;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);   // one critical section per host-buffer call: staging, kernels, download
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        TRY(dh_ensure(ctx, ctx->ws_ntt_io, std::max<size_t>(32, in_elems * 32)));
        d_in = (uint64_t*)ctx->ws_ntt_io.p;
        if (inplace) d_out = d_in;
        else {
            TRY(dh_ensure(ctx, ctx->ws_ntt_io2, std::max<size_t>(32, out_elems * 32)));
            d_out = (uint64_t*)ctx->ws_ntt_io2.p;
        }
        TRY(dh_h2d(ctx, d_in, in, in_elems * 32, ctx->stream));
    }
    TRY(fn(ctx, d_in, d_out, arg));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(dh_d2h(ctx, out, d_out, out_elems * 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

struct NttArgs { int field; uint32_t log_n, log_ext; const uint64_t *w, *s, *z; };

int dehalo_ntt(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_n, const uint64_t omega[4]) {
    if (log_n > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "log_n > 30");
    NttArgs A{field, log_n, 0, omega, nullptr, nullptr};
    size_t N = (size_t)1 << log_n;
    return with_host_io(ctx, a, N, a, N, true, [](dehalo_ctx* c, uint64_t* di, uint64_t*, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_ntt_device(c, A->field, di, A->log_n, A->w, 1, nullptr);
    }, &A);
}

int dehalo_intt_scaled(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_n, const uint64_t omega_inv[4], const uint64_t n_inv[4]) {
    if (log_n > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "log_n > 30");
    NttArgs A{field, log_n, 0, omega_inv, n_inv, nullptr};
    size_t N = (size_t)1 << log_n;
    return with_host_io(ctx, a, N, a, N, true, [](dehalo_ctx* c, uint64_t* di, uint64_t*, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_intt_scaled_device(c, A->field, di, A->log_n, A->w, A->s, 1, nullptr);
    }, &A);
}

int dehalo_coset_ntt(dehalo_ctx* ctx, int field, const uint64_t* coeffs, uint32_t log_n, uint64_t* ext_out, uint32_t log_ext,
                     const uint64_t omega_ext[4], const uint64_t zeta[4]) {
    if (log_ext > 30 || log_ext < log_n) return dh_fail(ctx, DEHALO_ERR_INVALID, "coset_ntt: bad sizes");
    NttArgs A{field, log_n, log_ext, omega_ext, nullptr, zeta};
    return with_host_io(ctx, coeffs, (size_t)1 << log_n, ext_out, (size_t)1 << log_ext, false, [](dehalo_ctx* c, uint64_t* di, uint64_t* dout, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_coset_ntt_device(c, A->field, di, A->log_n, dout, A->log_ext, A->w, A->z, 1, nullptr);
    }, &A);
}

int dehalo_coset_intt(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_ext, const uint64_t omega_ext_inv[4], const uint64_t ext_n_inv[4],
                      const uint64_t zeta[4]) {
    if (log_ext > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "log_ext > 30");
    NttArgs A{field, 0, log_ext, omega_ext_inv, ext_n_inv, zeta};
    size_t N = (size_t)1 << log_ext;
    return with_host_io(ctx, a, N, a, N, true, [](dehalo_ctx* c, uint64_t* di, uint64_t*, void* p) {
        NttArgs* A = (NttArgs*)p;
        return dehalo_coset_intt_device(c, A->field, di, A->log_ext, A->w, A->s, A->z, 1, nullptr);
    }, &A);
}

int dehalo_field_op(dehalo_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!a || !out || op < 0 || op > 6) return dh_fail(ctx, DEHALO_ERR_INVALID, "field_op: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    TRY(dh_ensure(ctx, ctx->ws_fop[0], n * 32));
    TRY(dh_ensure(ctx, ctx->ws_fop[1], n * 32));
    TRY(dh_ensure(ctx, ctx->ws_fop[2], n * 32));
    TRY(dh_h2d(ctx, ctx->ws_fop[0].p, a, n * 32, ctx->stream));
    if (b) TRY(dh_h2d(ctx, ctx->ws_fop[1].p, b, n * 32, ctx->stream));
    TRY(do_field_op(ctx, field, op, (const fe*)ctx->ws_fop[0].p, b ? (const fe*)ctx->ws_fop[1].p : nullptr, (fe*)ctx->ws_fop[2].p, n, ctx->stream));
    TRY(dh_d2h(ctx, out, ctx->ws_fop[2].p, n * 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_field_op_device(dehalo_ctx* ctx, int field, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (((!d_a || !d_out) && n) || op < 0 || op > 6) return dh_fail(ctx, DEHALO_ERR_INVALID, "field_op: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_field_op(ctx, field, op, (const fe*)d_a, (const fe*)d_b, (fe*)d_out, n, pick_stream(ctx, stream));
}

// ---- field-vector primitives (poly.cuh) ---------------------------------------------------------
int dehalo_eval_polynomial_device(dehalo_ctx* ctx, int field, const uint64_t* d_coeffs, size_t len, size_t stride_elems, size_t batch,
                                  const uint64_t point[4], uint64_t* d_out, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_coeffs && len) || !point || !d_out) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial: null argument");
    if (batch > 1 && stride_elems < len) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial: stride shorter than the polynomial");
    if (batch >= 65536) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial: batch too large");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_eval_poly(ctx, field, (const fe*)d_coeffs, len, stride_elems, batch, point, (fe*)d_out, pick_stream(ctx, stream));
}

int dehalo_eval_polynomial_multi_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_polys, size_t count, size_t len, const uint64_t* points,
                                        uint32_t num_points, uint64_t* d_out, void* stream) {
    return dehalo_eval_polynomial_multi_masked_device(ctx, field, d_polys, count, len, points, num_points, nullptr, d_out, stream);
}

int dehalo_eval_polynomial_multi_masked_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_polys, size_t count, size_t len, const uint64_t* points,
                                               uint32_t num_points, const uint8_t* wanted, uint64_t* d_out, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((count && !d_polys) || !points || !d_out) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial_multi: null argument");
    for (size_t j = 0; j < count; j++)
        if (!d_polys[j] && len) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial_multi: null polynomial");
    if (num_points == 0 || num_points > 4) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial_multi: 1 to 4 points");
    if (count >= 65536) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial_multi: too many polynomials");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_eval_poly_multi(ctx, field, (const fe* const*)d_polys, count, len, points, num_points, (fe*)d_out, pick_stream(ctx, stream), wanted);
}

int dehalo_eval_polynomial(dehalo_ctx* ctx, int field, const uint64_t* coeffs, size_t len, const uint64_t point[4], uint64_t out[4]) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!coeffs && len) || !point || !out) return dh_fail(ctx, DEHALO_ERR_INVALID, "eval_polynomial: null argument");
    HostPin pin_c(coeffs, len * 32);
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);   // one critical section per host-buffer call: staging, kernels, download
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        TRY(dh_ensure(ctx, ctx->ws_poly_io[0], std::max<size_t>(32, len * 32)));
        TRY(dh_ensure(ctx, ctx->ws_poly_io[1], 32));
        if (len) TRY(dh_h2d(ctx, ctx->ws_poly_io[0].p, coeffs, len * 32, ctx->stream));
    }
    TRY(dehalo_eval_polynomial_device(ctx, field, (const uint64_t*)ctx->ws_poly_io[0].p, len, len, 1, point, (uint64_t*)ctx->ws_poly_io[1].p, nullptr));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(dh_d2h(ctx, out, ctx->ws_poly_io[1].p, 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_batch_invert_device(dehalo_ctx* ctx, int field, uint64_t* d_values, size_t len, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!d_values && len) return dh_fail(ctx, DEHALO_ERR_INVALID, "batch_invert: null argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_batch_invert(ctx, field, (fe*)d_values, len, pick_stream(ctx, stream));
}

int dehalo_batch_invert(dehalo_ctx* ctx, int field, uint64_t* values, size_t len) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!values && len) return dh_fail(ctx, DEHALO_ERR_INVALID, "batch_invert: null argument");
    HostPin pin_v(values, len * 32);
    if (len == 0) return 0;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);   // one critical section per host-buffer call: staging, kernels, download
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        TRY(dh_ensure(ctx, ctx->ws_poly_io[0], len * 32));
        TRY(dh_h2d(ctx, ctx->ws_poly_io[0].p, values, len * 32, ctx->stream));
    }
    TRY(dehalo_batch_invert_device(ctx, field, (uint64_t*)ctx->ws_poly_io[0].p, len, nullptr));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(dh_d2h(ctx, values, ctx->ws_poly_io[0].p, len * 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_prefix_product_device(dehalo_ctx* ctx, int field, const uint64_t* d_in, size_t len, uint64_t* d_out, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_in || !d_out) && len) return dh_fail(ctx, DEHALO_ERR_INVALID, "prefix_product: null argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_prefix_product(ctx, field, (const fe*)d_in, len, (fe*)d_out, pick_stream(ctx, stream));
}

int dehalo_grand_product_batch_device(dehalo_ctx* ctx, int field, const uint64_t* d_num, const uint64_t* d_den, size_t len, size_t batch, size_t stride_elems,
                                      uint64_t* d_z, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_num || !d_den || !d_z) && len && batch) return dh_fail(ctx, DEHALO_ERR_INVALID, "grand_product: null argument");
    if (batch > 1 && stride_elems < len) return dh_fail(ctx, DEHALO_ERR_INVALID, "grand_product: stride shorter than the columns");
    if (batch >= 65536) return dh_fail(ctx, DEHALO_ERR_INVALID, "grand_product: batch too large");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_grand_product(ctx, field, (const fe*)d_num, (const fe*)d_den, len, batch, stride_elems, (fe*)d_z, pick_stream(ctx, stream));
}

int dehalo_grand_product_device(dehalo_ctx* ctx, int field, const uint64_t* d_num, const uint64_t* d_den, size_t len, uint64_t* d_z, void* stream) {
    return dehalo_grand_product_batch_device(ctx, field, d_num, d_den, len, 1, len, d_z, stream);
}

int dehalo_grand_product(dehalo_ctx* ctx, int field, const uint64_t* num, const uint64_t* den, size_t len, uint64_t* z) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!num || !den || !z) && len) return dh_fail(ctx, DEHALO_ERR_INVALID, "grand_product: null argument");
    HostPin pin_n(num, len * 32), pin_d(den, len * 32), pin_z(z, len * 32);
    if (len == 0) return 0;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);   // one critical section per host-buffer call: staging, kernels, download
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        for (int i = 0; i < 3; i++) TRY(dh_ensure(ctx, ctx->ws_poly_io[i], len * 32));
        TRY(dh_h2d(ctx, ctx->ws_poly_io[0].p, num, len * 32, ctx->stream));
        TRY(dh_h2d(ctx, ctx->ws_poly_io[1].p, den, len * 32, ctx->stream));
    }
    TRY(dehalo_grand_product_device(ctx, field, (const uint64_t*)ctx->ws_poly_io[0].p, (const uint64_t*)ctx->ws_poly_io[1].p, len,
                                    (uint64_t*)ctx->ws_poly_io[2].p, nullptr));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(dh_d2h(ctx, z, ctx->ws_poly_io[2].p, len * 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_lincomb_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_cols, const uint64_t* coefs, size_t count, size_t len, uint64_t* d_out,
                          const uint64_t* sub_const, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((count && (!d_cols || !coefs)) || (!d_out && len)) return dh_fail(ctx, DEHALO_ERR_INVALID, "lincomb: null argument");
    for (size_t j = 0; j < count; j++)
        if (!d_cols[j] && len) return dh_fail(ctx, DEHALO_ERR_INVALID, "lincomb: null column");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_lincomb(ctx, field, (const fe* const*)d_cols, coefs, count, len, (fe*)d_out, sub_const, pick_stream(ctx, stream));
}

int dehalo_scale_device(dehalo_ctx* ctx, int field, uint64_t* d_a, size_t len, const uint64_t* pattern, uint32_t period, const uint64_t* d_factor, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_a && len) || (period && !pattern)) return dh_fail(ctx, DEHALO_ERR_INVALID, "scale: null argument");
    if (period > 8 || (period & (period - 1))) return dh_fail(ctx, DEHALO_ERR_INVALID, "scale: period must be 0, 1, 2, 4 or 8");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_scale(ctx, field, (fe*)d_a, len, pattern, period, (const fe*)d_factor, pick_stream(ctx, stream));
}

int dehalo_kate_division_device(dehalo_ctx* ctx, int field, const uint64_t* d_a, size_t len, const uint64_t point[4], uint64_t* d_q, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (((!d_a || !d_q) && len > 1) || !point) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division: null argument");
    if (d_a == d_q) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division: a and q may not alias");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_kate_division(ctx, field, (const fe*)d_a, len, point, (fe*)d_q, pick_stream(ctx, stream));
}

int dehalo_kate_division_batch_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_a, size_t len, const uint64_t* points, uint64_t* const* d_q,
                                      size_t count, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (count && (!d_a || !d_q || !points)) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division_batch: null argument");
    if (count > 8) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division_batch: at most 8 divisions per call");
    if (len > (1ull << 22)) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division_batch: polynomials of at most 2^22 coefficients");
    for (size_t y = 0; y < count; y++) {
        if ((!d_a[y] || !d_q[y]) && len > 1) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division_batch: null polynomial");
        if (d_a[y] == d_q[y]) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division_batch: a and q may not alias");
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_kate_division_batch(ctx, field, (const fe* const*)d_a, len, points, (fe* const*)d_q, count, pick_stream(ctx, stream));
}

int dehalo_kate_division(dehalo_ctx* ctx, int field, const uint64_t* a, size_t len, const uint64_t point[4], uint64_t* q) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (((!a || !q) && len > 1) || !point) return dh_fail(ctx, DEHALO_ERR_INVALID, "kate_division: null argument");
    HostPin pin_a(a, len * 32), pin_q(q, (len - 1) * 32);
    if (len <= 1) return 0;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);
    (void)hipSetDevice(ctx->device);
    TRY(dh_ensure(ctx, ctx->ws_poly_io[0], len * 32));
    TRY(dh_ensure(ctx, ctx->ws_poly_io[1], len * 32));
    TRY(dh_h2d(ctx, ctx->ws_poly_io[0].p, a, len * 32, ctx->stream));
    TRY(dehalo_kate_division_device(ctx, field, (const uint64_t*)ctx->ws_poly_io[0].p, len, point, (uint64_t*)ctx->ws_poly_io[1].p, nullptr));
    TRY(dh_d2h(ctx, q, ctx->ws_poly_io[1].p, (len - 1) * 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dehalo_permute_expression_pair_batch_device(dehalo_ctx* ctx, int field, const uint64_t* d_inputs, const uint64_t* d_tables, size_t usable_rows, size_t batch,
                                                size_t stride_elems, uint64_t* d_permuted_inputs, uint64_t* d_permuted_tables, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_inputs || !d_tables || !d_permuted_inputs || !d_permuted_tables) && usable_rows && batch)
        return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: null argument");
    if (d_permuted_inputs == d_inputs || d_permuted_tables == d_tables || d_permuted_inputs == d_tables || d_permuted_tables == d_inputs)
        return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: outputs may not alias inputs");
    if (batch > 1 && stride_elems < usable_rows) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: stride shorter than the columns");
    if (batch >= 4096) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: batch too large");
    if (field < 0 || field > 3) return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return lookup_permute_impl(ctx, field, (const fe*)d_inputs, (const fe*)d_tables, usable_rows, batch, stride_elems, (fe*)d_permuted_inputs,
                               (fe*)d_permuted_tables, pick_stream(ctx, stream));
}

static int permute_ptrs(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows, size_t batch,
                        uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, int* d_status, void* stream, const LookupDistinct* distinct);
int dehalo_permute_expression_pair_ptrs_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows, size_t batch,
                                               uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, void* stream) {
    return permute_ptrs(ctx, field, d_inputs, d_tables, usable_rows, batch, d_permuted_inputs, d_permuted_tables, nullptr, stream, nullptr);
}
int dehalo_permute_expression_pair_ptrs_deferred_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows,
                                                        size_t batch, uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, int32_t* d_status, void* stream) {
    if (ctx && !d_status && batch) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: null status array");
    return permute_ptrs(ctx, field, d_inputs, d_tables, usable_rows, batch, d_permuted_inputs, d_permuted_tables, (int*)d_status, stream, nullptr);
}
static int permute_ptrs(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows, size_t batch,
                        uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, int* d_status, void* stream, const LookupDistinct* distinct) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!d_inputs || !d_tables || !d_permuted_inputs || !d_permuted_tables) && batch) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: null argument");
    if (batch >= 4096) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: batch too large");
    if (field < 0 || field > 3) return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    for (size_t y = 0; y < batch && usable_rows; y++) {
        if (!d_inputs[y] || !d_tables[y] || !d_permuted_inputs[y] || !d_permuted_tables[y]) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: null column");
        for (size_t z = 0; z < batch; z++)
            if (d_permuted_inputs[y] == d_inputs[z] || d_permuted_inputs[y] == d_tables[z] || d_permuted_tables[y] == d_inputs[z] || d_permuted_tables[y] == d_tables[z])
                return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: outputs may not alias inputs");
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return lookup_permute_ptrs(ctx, field, (const fe* const*)d_inputs, (const fe* const*)d_tables, usable_rows, batch, (fe* const*)d_permuted_inputs, (fe* const*)d_permuted_tables,
                               pick_stream(ctx, stream), d_status, distinct);
}
int dehalo_permute_expression_pair_distinct_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows, size_t batch,
                                                   uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, const uint32_t* const* d_rep_rows,
                                                   const uint32_t* const* d_multiplicities, const uint32_t* distinct_count, int32_t* d_status, void* stream) {
    if (ctx && batch && (!d_rep_rows || !d_multiplicities || !distinct_count)) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair_distinct: null argument");
    std::vector<LookupDistinct> dist(batch);
    for (size_t y = 0; y < batch; y++) {
        dist[y] = LookupDistinct{d_rep_rows[y], d_multiplicities[y], distinct_count[y]};
        if (distinct_count[y] && (!d_rep_rows[y] || !d_multiplicities[y])) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair_distinct: null distinct-row array");
        if (distinct_count[y] > usable_rows) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair_distinct: more distinct rows than usable rows");
        for (size_t z = 0; z < y; z++)      // lookups of one table must describe it alike
            if (d_tables && d_tables[z] == d_tables[y] && (d_rep_rows[z] != d_rep_rows[y] || d_multiplicities[z] != d_multiplicities[y] || distinct_count[z] != distinct_count[y]))
                return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair_distinct: lookups that share a table must pass the same distinct-row arrays");
    }
    return permute_ptrs(ctx, field, d_inputs, d_tables, usable_rows, batch, d_permuted_inputs, d_permuted_tables, (int*)d_status, stream, dist.data());
}

int dehalo_permute_expression_pair_device(dehalo_ctx* ctx, int field, const uint64_t* d_input, const uint64_t* d_table, size_t usable_rows,
                                          uint64_t* d_permuted_input, uint64_t* d_permuted_table, void* stream) {
    return dehalo_permute_expression_pair_batch_device(ctx, field, d_input, d_table, usable_rows, 1, usable_rows, d_permuted_input, d_permuted_table, stream);
}

int dehalo_permute_expression_pair(dehalo_ctx* ctx, int field, const uint64_t* input, const uint64_t* table, size_t usable_rows, uint64_t* permuted_input,
                                   uint64_t* permuted_table) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((!input || !table || !permuted_input || !permuted_table) && usable_rows) return dh_fail(ctx, DEHALO_ERR_INVALID, "permute_expression_pair: null argument");
    HostPin pin_i(input, usable_rows * 32), pin_t(table, usable_rows * 32), pin_pi(permuted_input, usable_rows * 32), pin_pt(permuted_table, usable_rows * 32);
    if (usable_rows == 0) return 0;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);   // one critical section per host-buffer call: staging, kernels, download
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        TRY(dh_ensure(ctx, ctx->ws_poly_io[0], usable_rows * 64));
        TRY(dh_ensure(ctx, ctx->ws_poly_io[1], usable_rows * 64));
        TRY(dh_h2d(ctx, ctx->ws_poly_io[0].p, input, usable_rows * 32, ctx->stream));
        TRY(dh_h2d(ctx, (char*)ctx->ws_poly_io[0].p + usable_rows * 32, table, usable_rows * 32, ctx->stream));
    }
    uint64_t* d_in = (uint64_t*)ctx->ws_poly_io[0].p;
    uint64_t* d_out = (uint64_t*)ctx->ws_poly_io[1].p;
    TRY(dehalo_permute_expression_pair_device(ctx, field, d_in, d_in + usable_rows * 4, usable_rows, d_out, d_out + usable_rows * 4, nullptr));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(dh_d2h(ctx, permuted_input, d_out, usable_rows * 32, ctx->stream));
    TRY(dh_d2h(ctx, permuted_table, d_out + usable_rows * 4, usable_rows * 32, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- quotient numerator (evalh.cuh) -----------------------------------------------------------------
int dehalo_graph_create(dehalo_ctx* ctx, int field, const uint64_t* constants, uint32_t num_constants, const int32_t* rotations, uint32_t num_rotations,
                        const dehalo_calculation* calcs, uint32_t num_calcs, const dehalo_source* horner_parts, uint32_t num_horner_parts,
                        uint32_t num_intermediates, dehalo_graph** out) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!out || (num_constants && !constants) || (num_rotations && !rotations) || (num_calcs && !calcs) || (num_horner_parts && !horner_parts))
        return dh_fail(ctx, DEHALO_ERR_INVALID, "graph_create: null argument");
    if (field < 0 || field > 3) return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    if (num_calcs > (1u << 20) || num_intermediates > (1u << 20)) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph_create: program too large");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    dehalo_graph* g = new dehalo_graph();
    memset(g, 0, sizeof(*g));
    g->field = field; g->num_calcs = num_calcs; g->num_parts = num_horner_parts; g->num_constants = num_constants;
    std::vector<DevCalc> dc;
    std::vector<DevSrc> dp;
    int rc = compile_graph(ctx, g, rotations, num_rotations, calcs, num_calcs, horner_parts, num_horner_parts, num_intermediates, dc, dp);
    hipError_t e = hipSuccess;
    if (rc == 0) e = hipMalloc((void**)&g->d_calcs, std::max<size_t>(1, dc.size()) * sizeof(DevCalc));
    if (rc == 0 && e == hipSuccess) e = hipMalloc((void**)&g->d_parts, dp.size() * sizeof(DevSrc));
    if (rc == 0 && e == hipSuccess) e = hipMalloc((void**)&g->d_constants, std::max<size_t>(1, num_constants) * sizeof(fe));
    if (rc == 0 && e == hipSuccess && !dc.empty()) e = hipMemcpy(g->d_calcs, dc.data(), dc.size() * sizeof(DevCalc), hipMemcpyHostToDevice);
    if (rc == 0 && e == hipSuccess) e = hipMemcpy(g->d_parts, dp.data(), dp.size() * sizeof(DevSrc), hipMemcpyHostToDevice);
    if (rc == 0 && e == hipSuccess) rc = do_graph_upload(ctx, field, g, constants, ctx->stream);
    if (rc != 0 || e != hipSuccess) {
        (void)hipFree(g->d_calcs); (void)hipFree(g->d_parts); (void)hipFree(g->d_constants);
        delete g;
        return rc ? rc : dh_fail(ctx, e == hipErrorOutOfMemory ? DEHALO_ERR_OOM : DEHALO_ERR_HIP, std::string("graph_create: ") + hipGetErrorString(e));
    }
    *out = g;
    return 0;
}

int dehalo_graph_release(dehalo_ctx* ctx, dehalo_graph* g) {
    if (!ctx || !g) return DEHALO_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(g->d_calcs); (void)hipFree(g->d_parts); (void)hipFree(g->d_constants);
    delete g;
    return 0;
}

int dehalo_graph_evaluate_device(dehalo_ctx* ctx, const dehalo_graph* g, const dehalo_eval_inputs* in, uint32_t log_rows, uint32_t rot_scale,
                                 const uint64_t* d_previous, uint64_t* d_out, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!g || !in || !d_out) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph_evaluate: null argument");
    if (log_rows > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph_evaluate: log_rows > 30");
    if (in->num_fixed < g->max_fixed || in->num_advice < g->max_advice || in->num_instance < g->max_instance || in->num_challenges < g->max_challenge)
        return dh_fail(ctx, DEHALO_ERR_INVALID, "graph_evaluate: the program reads a column or challenge that was not supplied");
    if ((in->num_fixed && !in->fixed) || (in->num_advice && !in->advice) || (in->num_instance && !in->instance) || (in->num_challenges && !in->challenges))
        return dh_fail(ctx, DEHALO_ERR_INVALID, "graph_evaluate: null column table");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_graph_evaluate(ctx, g, in, log_rows, rot_scale, (const fe*)d_previous, (fe*)d_out, pick_stream(ctx, stream));
}

int dehalo_graph_evaluate_batch_device(dehalo_ctx* ctx, const dehalo_graph* const* graphs, uint32_t count, const dehalo_eval_inputs* in, uint32_t log_rows,
                                       uint32_t rot_scale, uint64_t* const* d_outs, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if ((count && (!graphs || !d_outs)) || !in) return dh_fail(ctx, DEHALO_ERR_INVALID, "graph_evaluate_batch: null argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    // DEHALO_GRAPH_BATCH=0: one staging + one evaluation launch per program, as before round 4 (A/B measurements)
    static const bool batched = [] { const char* e = DH_EXPERIMENT_ENV("DEHALO_GRAPH_BATCH"); return !(e && e[0] == '0'); }();
    bool same_field = count >= 2 && log_rows <= 30;
    for (uint32_t i = 0; same_field && i < count; i++) {
        const dehalo_graph* g = graphs[i];
        same_field = g && d_outs[i] && g->field == graphs[0]->field && in->num_fixed >= g->max_fixed && in->num_advice >= g->max_advice && in->num_instance >= g->max_instance &&
                     in->num_challenges >= g->max_challenge;
    }
    if (same_field && ((in->num_fixed && !in->fixed) || (in->num_advice && !in->advice) || (in->num_instance && !in->instance) || (in->num_challenges && !in->challenges))) same_field = false;
    if (!batched || !same_field) {      // (the single-program entry point reports what is wrong with an argument)
        for (uint32_t i = 0; i < count; i++)
            TRY(dehalo_graph_evaluate_device(ctx, graphs[i], in, log_rows, rot_scale, nullptr, d_outs[i], stream));
        return 0;
    }
    (void)hipSetDevice(ctx->device);
    hipStream_t s = pick_stream(ctx, stream);
#define CALL(N) graph_evaluate_batch_##N(ctx, graphs, count, in, log_rows, rot_scale, (fe* const*)d_outs, s)
    FIELD_SWITCH(ctx, graphs[0]->field, CALL)
#undef CALL
}

int dehalo_permutation_h_device(dehalo_ctx* ctx, int field, const dehalo_perm_inputs* in, uint32_t log_rows, uint32_t rot_scale, uint64_t* d_values,
                                void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!in || !d_values || !in->l0 || !in->l_last || !in->l_active_row || !in->beta || !in->gamma || !in->y || !in->delta || !in->beta_zeta || !in->extended_omega)
        return dh_fail(ctx, DEHALO_ERR_INVALID, "permutation_h: null argument");
    if ((in->num_sets && !in->z) || (in->num_columns && (!in->columns || !in->sigma))) return dh_fail(ctx, DEHALO_ERR_INVALID, "permutation_h: null column table");
    if (log_rows == 0 || log_rows > 30 || in->chunk_len == 0 || (uint64_t)in->num_sets * in->chunk_len < in->num_columns)
        return dh_fail(ctx, DEHALO_ERR_INVALID, "permutation_h: sets * chunk_len must cover the columns");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_perm_h(ctx, field, in, log_rows, rot_scale, (fe*)d_values, pick_stream(ctx, stream));
}

int dehalo_lookup_h_device(dehalo_ctx* ctx, int field, const dehalo_lookup_inputs* in, uint32_t log_rows, uint32_t rot_scale, uint64_t* d_values,
                           void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!in || !d_values || !in->product_coset || !in->permuted_input_coset || !in->permuted_table_coset || !in->table_value || !in->l0 || !in->l_last ||
        !in->l_active_row || !in->beta || !in->gamma || !in->y)
        return dh_fail(ctx, DEHALO_ERR_INVALID, "lookup_h: null argument");
    if (log_rows > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "lookup_h: log_rows > 30");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_lookup_h(ctx, field, in, log_rows, rot_scale, (fe*)d_values, pick_stream(ctx, stream));
}

int dehalo_product_terms_device(dehalo_ctx* ctx, int field, const dehalo_product_inputs* in, size_t n, uint64_t* d_num, uint64_t* d_den, size_t stride_elems, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!in || !d_num || !d_den || !in->beta || !in->gamma) return dh_fail(ctx, DEHALO_ERR_INVALID, "product_terms: null argument");
    if (in->num_columns && (!in->columns || !in->sigma || !in->omega_powers || !in->delta || !in->set_factors || in->chunk_len == 0))
        return dh_fail(ctx, DEHALO_ERR_INVALID, "product_terms: incomplete permutation inputs");
    if (in->num_lookups && (!in->compressed_input || !in->compressed_table || !in->permuted_input || !in->permuted_table))
        return dh_fail(ctx, DEHALO_ERR_INVALID, "product_terms: incomplete lookup inputs");
    if (in->num_columns > 256 || in->num_lookups > 256) return dh_fail(ctx, DEHALO_ERR_INVALID, "product_terms: too many columns");
    if (stride_elems < n) return dh_fail(ctx, DEHALO_ERR_INVALID, "product_terms: stride shorter than the columns");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    hipStream_t s = pick_stream(ctx, stream);
    switch (field) {
        case DEHALO_FIELD_BN254_FR: return product_terms_bn254_fr(ctx, in, n, (fe*)d_num, (fe*)d_den, stride_elems, s);
        case DEHALO_FIELD_BN254_FQ: return product_terms_bn254_fq(ctx, in, n, (fe*)d_num, (fe*)d_den, stride_elems, s);
        case DEHALO_FIELD_PASTA_FP: return product_terms_pasta_fp(ctx, in, n, (fe*)d_num, (fe*)d_den, stride_elems, s);
        case DEHALO_FIELD_PASTA_FQ: return product_terms_pasta_fq(ctx, in, n, (fe*)d_num, (fe*)d_den, stride_elems, s);
        default: return dh_fail(ctx, DEHALO_ERR_INVALID, "unknown field id");
    }
}

int dehalo_lookup_h_batch_device(dehalo_ctx* ctx, int field, const dehalo_lookup_inputs* in, uint32_t count, uint32_t log_rows, uint32_t rot_scale,
                                 uint64_t* d_values, void* stream) {
    if (!ctx) return DEHALO_ERR_INVALID;
    if (!in || !d_values || count == 0) return dh_fail(ctx, DEHALO_ERR_INVALID, "lookup_h_batch: null argument");
    if (count > 8) return dh_fail(ctx, DEHALO_ERR_INVALID, "lookup_h_batch: more than 8 lookups in one call");
    for (uint32_t l = 0; l < count; l++) {
        const dehalo_lookup_inputs& q = in[l];
        if (!q.product_coset || !q.permuted_input_coset || !q.permuted_table_coset || !q.table_value || !q.l0 || !q.l_last || !q.l_active_row || !q.beta ||
            !q.gamma || !q.y)
            return dh_fail(ctx, DEHALO_ERR_INVALID, "lookup_h_batch: null argument");
        if (q.l0 != in[0].l0 || q.l_last != in[0].l_last || q.l_active_row != in[0].l_active_row || q.form_flags != in[0].form_flags ||
            memcmp(q.beta, in[0].beta, 32) || memcmp(q.gamma, in[0].gamma, 32) || memcmp(q.y, in[0].y, 32))
            return dh_fail(ctx, DEHALO_ERR_INVALID, "lookup_h_batch: the lookups of one call share l0 / l_last / l_active_row, the challenges and the form flags");
    }
    if (log_rows > 30) return dh_fail(ctx, DEHALO_ERR_INVALID, "lookup_h: log_rows > 30");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return do_lookup_h_batch(ctx, field, in, count, log_rows, rot_scale, (fe*)d_values, pick_stream(ctx, stream));
}

// ---- measurement ------------------------------------------------------------------------------
int dehalo_timing_enable(dehalo_ctx* ctx, int on) {
    if (!ctx) return DEHALO_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
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
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(timing_collect(ctx));
    for (int i = 0; i < DEHALO_K_COUNT; i++) { ctx->timing_ms[i] = 0; ctx->timing_cnt[i] = 0; }
    return 0;
}

int dehalo_timing_get(dehalo_ctx* ctx, int kernel_id, double* total_ms, uint64_t* count) {
    if (!ctx || kernel_id < 0 || kernel_id >= DEHALO_K_COUNT || !total_ms || !count) return DEHALO_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    TRY(timing_collect(ctx));
    *total_ms = ctx->timing_ms[kernel_id];
    *count = ctx->timing_cnt[kernel_id];
    return 0;
}

int dehalo_msm_last_shape(dehalo_ctx* ctx, uint32_t out[6]) {
    if (!ctx || !out) return DEHALO_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    memset(out, 0, 6 * sizeof(uint32_t));
    if (!ctx->ws_counters.p) return 0;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t raw[10];      // eight merge-class counters (k_msm_merge_classify2: 2 | 3-4 | 5-8 | 9-64 | 65-512 records, parts of heavy buckets, heavy buckets, unused), L0, M
    HIP_TRY(ctx, hipMemcpy(raw, ctx->ws_counters.p, sizeof(raw), hipMemcpyDeviceToHost));
    out[0] = raw[0] + raw[1] + raw[2]; out[1] = raw[3]; out[2] = raw[4]; out[3] = raw[6]; out[4] = raw[8]; out[5] = raw[9];      // light (2-8 records) | 9-64 | 65-512 | > 512
    return 0;
}

}  // extern "C"
