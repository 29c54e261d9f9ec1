// evalh.cuh -- the quotient numerator of create_proof on the device.
//
// Replaces the row loops of [UPSTREAM halo2_proofs/src/plonk/evaluation.rs @ v2023_04_20]:
//   GraphEvaluator::evaluate      custom gates and the lookups' compressed (input, table) product:
//                                 a straight-line program of Calculation{Add, Sub, Mul, Square,
//                                 Double, Negate, Horner, Store} over ValueSource{Constant,
//                                 Intermediate, Fixed, Advice, Instance, Challenge, Beta, Gamma,
//                                 Theta, Y, PreviousValue}, run once per extended-domain row;
//   Evaluator::evaluate_h         the hard-coded permutation and lookup argument terms folded into
//                                 the same accumulator with  value = value * y + term.
// Upstream runs them with `parallelize` over row chunks on the CPU; here one lane owns one row,
// every lane executes the same instruction stream (the program is uniform), intermediates live in
// LDS slots assigned by a liveness pass on the host (spilling to HBM beyond the LDS budget), and
// columns are read with the rotation folded into the index:  (row + rot * rot_scale) mod rows.
//
// Values in a slot: x * 2^261, < 2p, limbs normalized.
#pragma once
#include <vector>

#include "fp29.cuh"
#include "internal.hpp"

#include "evalh_types.hpp"

#define EVH_ARG_COLS 40
struct EvhArgs {
    const DevCalc* calcs;
    const DevSrc* parts;
    const fe* scalars;            // [beta, gamma, theta, y, constants..., challenges...] internal packed
    const fe* const* columns;     // [fixed..., advice..., instance...] device pointers (table in HBM: a table carried in the kernel
                                  // arguments and indexed dynamically is demoted to scratch memory -- measured 40 % slower)
    u32 num_calcs, fixed_base, advice_base, instance_base;
    u32 rows_mask, rot_scale;
    const fe* previous;
    fe* out;
    fe* spill;                    // [hbm slot][row], internal packed
    u64 rows;
    DevSrc result;
    u32 cols_internal, vals_internal;
};

// ---- lazily reduced helpers: every stored value is < 2p with normalized limbs ----------------
template <class F9> FP_DEV f29 evh_reduce_lt4p(const f29& a) { return f29_cond_sub(f29_norm(a), F9::P2); }
template <class F9> FP_DEV f29 evh_add(const f29& a, const f29& b) { return evh_reduce_lt4p<F9>(f29_add(a, b)); }
template <class F9> FP_DEV f29 evh_sub(const f29& a, const f29& b) {                  // a - b + 4p < 6p
    f29 t = f29_norm(f29_sub(a, b, F9::KM));
    return f29_cond_sub(f29_cond_sub(t, F9::P4), F9::P2);
}
template <class F9> FP_DEV f29 evh_neg(const f29& a) { return evh_reduce_lt4p<F9>(f29_sub(f29_zero(), a, F9::KM)); }

struct EvhLds {
    u32* base;
    FP_DEV f29 load(u32 slot) const {
        f29 r;
#pragma unroll
        for (int l = 0; l < 9; l++) r.v[l] = base[(slot * 9 + l) * EVH_THREADS + threadIdx.x];
        return r;
    }
    FP_DEV void store(u32 slot, const f29& v) const {
#pragma unroll
        for (int l = 0; l < 9; l++) base[(slot * 9 + l) * EVH_THREADS + threadIdx.x] = v.v[l];
    }
};

template <class F9>
FP_DEV f29 evh_fetch(const EvhArgs& A, const EvhLds& L, const DevSrc& s, u64 row) {
    switch (s.kind) {
        case EVS_SCALAR: return f29_unpack(f_load(&A.scalars[s.index]));
        case EVS_SLOT_LDS: return L.load(s.index);
        case EVS_SLOT_HBM: return f29_unpack(f_load(&A.spill[(u64)s.index * A.rows + row]));
        case EVS_PREVIOUS:
            if (!A.previous) return f29_zero();
            return A.vals_internal ? f29_unpack(f_load(&A.previous[row])) : f29_from_std<F9>(f_load(&A.previous[row]));
        default: {
            u32 base = s.kind == EVS_FIXED ? A.fixed_base : (s.kind == EVS_ADVICE ? A.advice_base : A.instance_base);
            const fe* col = A.columns[base + s.index];
            u32 r = ((u32)row + (u32)(s.rot * (int32_t)A.rot_scale)) & A.rows_mask;    // rem_euclid for a power of two
            return A.cols_internal ? f29_unpack(f_load(&col[r])) : f29_from_std<F9>(f_load(&col[r]));
        }
    }
}

template <class F>
FP_DEV void graph_eval_body(const EvhArgs& A) {
    typedef typename f29_of<F>::type F9;
    extern __shared__ u32 evh_lds[];
    EvhLds L{evh_lds};
    const u64 row = (u64)blockIdx.x * EVH_THREADS + threadIdx.x;
    if (row >= A.rows) return;                  // no barriers below: a lane only touches its own LDS column
    for (u32 ci = 0; ci < A.num_calcs; ci++) {
        const DevCalc c = A.calcs[ci];
        f29 a = evh_fetch<F9>(A, L, c.a, row);
        f29 r;
        switch (c.op) {
            case DEHALO_CALC_ADD: r = evh_add<F9>(a, evh_fetch<F9>(A, L, c.b, row)); break;
            case DEHALO_CALC_SUB: r = evh_sub<F9>(a, evh_fetch<F9>(A, L, c.b, row)); break;
            case DEHALO_CALC_MUL: r = f29_mul<F9>(a, evh_fetch<F9>(A, L, c.b, row)); break;
            case DEHALO_CALC_SQUARE: r = f29_sqr<F9>(a); break;
            case DEHALO_CALC_DOUBLE: r = evh_add<F9>(a, a); break;
            case DEHALO_CALC_NEGATE: r = evh_neg<F9>(a); break;
            case DEHALO_CALC_HORNER: {
                const f29 factor = evh_fetch<F9>(A, L, c.b, row);
                r = a;
                for (u32 k = 0; k < c.parts_len; k++)
                    r = evh_add<F9>(f29_mul<F9>(r, factor), evh_fetch<F9>(A, L, A.parts[c.parts_begin + k], row));
            } break;
            default: r = a; break;              // STORE
        }
        if (c.target_kind == EVS_SLOT_LDS) L.store(c.target_slot, r);
        else f_store(&A.spill[(u64)c.target_slot * A.rows + row], f29_to_packed_canon<F9>(r));
    }
    f29 res = A.num_calcs ? evh_fetch<F9>(A, L, A.result, row) : f29_zero();
    f_store(&A.out[row], A.vals_internal ? f29_to_packed_canon<F9>(res) : f29_to_std<F9>(res));
}
template <class F>
__global__ __launch_bounds__(EVH_THREADS) void k_graph_eval(EvhArgs A) { graph_eval_body<F>(A); }
// several programs over the same rows and columns in ONE launch (grid.y = program): the lookups' compressed input / table columns of a proof were a
// staging kernel + an evaluation kernel EACH, a dozen launches queued behind the side context's NTTs (0.6 ms of the lookups' phase at k = 17 for
// microseconds of arithmetic).  The argument blocks live in device memory (written by k_evh_stage_batch): a block reads its own with scalar loads.
template <class F>
__global__ __launch_bounds__(EVH_THREADS) void k_graph_eval_batch(const EvhArgs* args) {
    const EvhArgs A = args[blockIdx.y];
    graph_eval_body<F>(A);
}

// standard-form scalars -> the internal packed table.  Per-call values (challenges, beta, ...)
// travel as kernel arguments, eight at a time: no pageable host buffer has to outlive the call.
template <class F>
__global__ void k_evh_scalars(const fe* std_in, fe* out, u32 n) {
    typedef typename f29_of<F>::type F9;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f_store(&out[i], f29_to_packed_canon<F9>(f29_from_std<F9>(f_load(&std_in[i]))));
}
template <class F>
__global__ void k_convert_form(const fe* in, fe* out, u64 n, int to_internal) {
    typedef typename f29_of<F>::type F9;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (to_internal) f_store(&out[i], f29_to_packed_canon<F9>(f29_from_std<F9>(f_load(&in[i]))));
    else f_store(&out[i], f29_to_std<F9>(f29_unpack(f_load(&in[i]))));
}
struct FeBatch { fe v[8]; };
template <class F>
__global__ void k_evh_scalars_val(FeBatch b, fe* out, u32 n) {
    typedef typename f29_of<F>::type F9;
    u32 i = threadIdx.x;
    if (i < n) f_store(&out[i], f29_to_packed_canon<F9>(f29_from_std<F9>(b.v[i])));
}
// the whole scalar table of one graph evaluation in ONE launch: [beta, gamma, theta, y | constants (already internal) | challenges]
// and the column pointer table
struct EvhStage { fe v[16]; const fe* constants; u32 nconst, nchal; const void* cols[EVH_ARG_COLS]; u32 ncols; };
template <class F>
__global__ void k_evh_stage(EvhStage st, fe* table, const void** ptr_table) {
    typedef typename f29_of<F>::type F9;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, total = 4 + st.nconst + st.nchal;
    if (i < st.ncols) {                                       // (static unrolled select: no dynamic indexing of the argument array)
        const void* p = nullptr;
#pragma unroll
        for (u32 j = 0; j < EVH_ARG_COLS; j++) if (j == i) p = st.cols[j];
        ptr_table[i] = p;
    }
    if (i >= total) return;
    if (i < 4) f_store(&table[i], f29_to_packed_canon<F9>(f29_from_std<F9>(st.v[i])));
    else if (i < 4 + st.nconst) f_store(&table[i], f_load(&st.constants[i - 4]));
    else f_store(&table[i], f29_to_packed_canon<F9>(f29_from_std<F9>(st.v[4 + (i - 4 - st.nconst)])));
}
#define EVH_GRAPH_BATCH 8
struct EvhStageBatch {
    fe v[16]; u32 nchal;                                   // beta, gamma, theta, y, challenges (standard form)
    const void* cols[EVH_ARG_COLS]; u32 ncols;
    const fe* constants[EVH_GRAPH_BATCH]; u32 nconst[EVH_GRAPH_BATCH];
    EvhArgs args[EVH_GRAPH_BATCH]; u32 count, tstride;     // program g's scalar table at table + g * tstride
};
template <class F>
__global__ void k_evh_stage_batch(EvhStageBatch st, fe* table, const void** ptr_table, EvhArgs* args_out) {
    typedef typename f29_of<F>::type F9;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    u32 nconst = 0; const fe* consts = nullptr;
#pragma unroll
    for (u32 j = 0; j < EVH_GRAPH_BATCH; j++) if (j == g) { nconst = st.nconst[j]; consts = st.constants[j]; }
    if (g == 0 && i < st.ncols) {
        const void* p = nullptr;
#pragma unroll
        for (u32 j = 0; j < EVH_ARG_COLS; j++) if (j == i) p = st.cols[j];
        ptr_table[i] = p;
    }
    if (i == 0) {
#pragma unroll
        for (u32 j = 0; j < EVH_GRAPH_BATCH; j++) if (j == g) args_out[j] = st.args[j];
    }
    const u32 total = 4 + nconst + st.nchal;
    if (i >= total) return;
    fe* t = table + (size_t)g * st.tstride;
    if (i < 4 || i >= 4 + nconst) {
        fe x{};
        const u32 k = i < 4 ? i : 4 + (i - 4 - nconst);
#pragma unroll
        for (u32 j = 0; j < 16; j++) if (j == k) x = st.v[j];
        f_store(&t[i], f29_to_packed_canon<F9>(f29_from_std<F9>(x)));
    } else f_store(&t[i], f_load(&consts[i - 4]));
}
struct PtrBatch { const void* p[32]; };
static __global__ void k_evh_ptrs(PtrBatch b, const void** out, u32 n) {
    u32 i = threadIdx.x;
    if (i < n) out[i] = b.p[i];
}

// (every  v = v y + term * l  below is ONE reduction of two products, f29_mul2: the reduction is the larger half of a bn256::Fr multiplication --
// tools/ubench_mfma_price.hip: product 81 multiply-adds 2.4 ms, reduction 2.9 ms of the same run)
// ---- permutation argument terms (Evaluator::evaluate_h, "Permutation constraints") ------------
// sets: the grand-product cosets z_0 .. z_{S-1}; cols: the permuted columns' cosets in order,
// chunk_len per set; sigma: pk.permutation.cosets in the same order.  Per row (X = zeta w^row):
//   v = v y + l0 (1 - z_0)
//   v = v y + l_last (z_last^2 - z_last)
//   for s >= 1:  v = v y + l0 (z_s - z_{s-1}(w^last X))
//   for each set: v = v y + l_active ( z_s(wX) prod_j (col_j + beta sigma_j + gamma)
//                                   - z_s(X)  prod_j (col_j + delta^j beta X + gamma) )      delta^j running across sets
struct PermArgs {
    const fe* const* z;        // S set cosets
    const fe* const* cols;     // ncols column cosets
    const fe* const* sigma;    // ncols permutation cosets
    const fe* l0; const fe* l_last; const fe* l_active;
    const fe* scalars;         // internal packed: [beta, gamma, y, delta, beta*zeta]
    const fe* tw;              // omega_ext^j * 2^261 (canonical, packed), j < rows / 2: the NTT's table
    u32 nsets, ncols, chunk_len;
    u32 rows_mask, rot_scale;
    int32_t last_rotation;     // -(blinding_factors + 1)
    fe* values;                // in/out
    u64 rows;
    u32 cols_internal, vals_internal;
};

template <class F>
__global__ __launch_bounds__(EVH_THREADS) void k_perm_h(PermArgs A) {
    typedef typename f29_of<F>::type F9;
    const u64 row = (u64)blockIdx.x * EVH_THREADS + threadIdx.x;
    if (row >= A.rows) return;
    const f29 beta = f29_unpack(f_load(&A.scalars[0])), gamma = f29_unpack(f_load(&A.scalars[1])), y = f29_unpack(f_load(&A.scalars[2]));
    const f29 delta = f29_unpack(f_load(&A.scalars[3])), beta_zeta = f29_unpack(f_load(&A.scalars[4]));
    const f29 one = f29_one<F9>();
    const u32 r_next = ((u32)row + A.rot_scale) & A.rows_mask;
    const u32 r_last = ((u32)row + (u32)(A.last_rotation * (int32_t)A.rot_scale)) & A.rows_mask;
    auto ld = [&](const fe* p, u64 i) __attribute__((always_inline)) { return A.cols_internal ? f29_unpack(f_load(&p[i])) : f29_from_std<F9>(f_load(&p[i])); };
    f29 v = A.vals_internal ? f29_unpack(f_load(&A.values[row])) : f29_from_std<F9>(f_load(&A.values[row]));
    const f29 l0 = ld(A.l0, row), l_last = ld(A.l_last, row), l_active = ld(A.l_active, row);
    if (A.nsets) {
        f29 z0 = ld(A.z[0], row);
        v = f29_mul2<F9>(v, y, evh_sub<F9>(one, z0), l0);
        f29 zl = ld(A.z[A.nsets - 1], row);
        v = f29_mul2<F9>(v, y, evh_sub<F9>(f29_sqr<F9>(zl), zl), l_last);
        for (u32 s = 1; s < A.nsets; s++) {
            f29 t = evh_sub<F9>(ld(A.z[s], row), ld(A.z[s - 1], r_last));
            v = f29_mul2<F9>(v, y, t, l0);
        }
        // beta * zeta * omega_ext^row, omega_ext^row from the NTT twiddle table (omega^(rows/2) = -1)
        const u64 half = A.rows >> 1;
        f29 w = f29_unpack(f_load(&A.tw[row >= half ? row - half : row]));
        if (row >= half) w = evh_neg<F9>(w);
        f29 current_delta = f29_mul<F9>(beta_zeta, w);
        for (u32 s = 0; s < A.nsets; s++) {
            const u32 c0 = s * A.chunk_len, c1 = min(c0 + A.chunk_len, A.ncols);
            f29 left = ld(A.z[s], r_next), right = ld(A.z[s], row);
            for (u32 j = c0; j < c1; j++) {
                f29 col = ld(A.cols[j], row);
                f29 t = evh_add<F9>(evh_add<F9>(col, f29_mul<F9>(beta, ld(A.sigma[j], row))), gamma);
                left = f29_mul<F9>(left, t);
                f29 u = evh_add<F9>(evh_add<F9>(col, current_delta), gamma);
                right = f29_mul<F9>(right, u);
                current_delta = f29_mul<F9>(current_delta, delta);
            }
            v = f29_mul2<F9>(v, y, evh_sub<F9>(left, right), l_active);
        }
    }
    f_store(&A.values[row], A.vals_internal ? f29_to_packed_canon<F9>(v) : f29_to_std<F9>(v));
}

// ---- lookup argument terms (Evaluator::evaluate_h, "Lookup constraints") ----------------------
// table_value[row] = (compressed input + beta)(compressed table + gamma) comes from the lookup's
// GraphEvaluator (k_graph_eval into a scratch column).  Per row:
//   v = v y + l0 (1 - z)
//   v = v y + l_last (z^2 - z)
//   v = v y + l_active ( z(wX) (a' + beta)(s' + gamma) - z(X) table_value )
//   v = v y + l0 (a' - s')
//   v = v y + l_active (a' - s')(a' - a'(w^-1 X))
struct LookupArgs {
    const fe* z; const fe* a_perm; const fe* s_perm; const fe* table_value;
    const fe* l0; const fe* l_last; const fe* l_active;
    const fe* scalars;         // internal packed: [beta, gamma, y]
    u32 rows_mask, rot_scale;
    fe* values;
    u64 rows;
    u32 cols_internal, vals_internal;
};

template <class F>
__global__ __launch_bounds__(EVH_THREADS) void k_lookup_h(LookupArgs A) {
    typedef typename f29_of<F>::type F9;
    const u64 row = (u64)blockIdx.x * EVH_THREADS + threadIdx.x;
    if (row >= A.rows) return;
    const f29 beta = f29_unpack(f_load(&A.scalars[0])), gamma = f29_unpack(f_load(&A.scalars[1])), y = f29_unpack(f_load(&A.scalars[2]));
    const f29 one = f29_one<F9>();
    const u32 r_next = ((u32)row + A.rot_scale) & A.rows_mask;
    const u32 r_prev = ((u32)row - A.rot_scale) & A.rows_mask;
    auto ld = [&](const fe* p, u64 i) __attribute__((always_inline)) { return A.cols_internal ? f29_unpack(f_load(&p[i])) : f29_from_std<F9>(f_load(&p[i])); };
    auto ldv = [&](const fe* p, u64 i) __attribute__((always_inline)) { return A.vals_internal ? f29_unpack(f_load(&p[i])) : f29_from_std<F9>(f_load(&p[i])); };
    f29 v = ldv(A.values, row);
    const f29 l0 = ld(A.l0, row), l_last = ld(A.l_last, row), l_active = ld(A.l_active, row);
    const f29 z = ld(A.z, row), a = ld(A.a_perm, row), s = ld(A.s_perm, row);
    const f29 a_minus_s = evh_sub<F9>(a, s);
    v = f29_mul2<F9>(v, y, evh_sub<F9>(one, z), l0);
    v = f29_mul2<F9>(v, y, evh_sub<F9>(f29_sqr<F9>(z), z), l_last);
    f29 lhs = f29_mul<F9>(f29_mul<F9>(ld(A.z, r_next), evh_add<F9>(a, beta)), evh_add<F9>(s, gamma));
    f29 rhs = f29_mul<F9>(z, ldv(A.table_value, row));
    v = f29_mul2<F9>(v, y, evh_sub<F9>(lhs, rhs), l_active);
    v = f29_mul2<F9>(v, y, a_minus_s, l0);
    f29 t = f29_mul<F9>(a_minus_s, evh_sub<F9>(a, ld(A.a_perm, r_prev)));
    v = f29_mul2<F9>(v, y, t, l_active);
    f_store(&A.values[row], A.vals_internal ? f29_to_packed_canon<F9>(v) : f29_to_std<F9>(v));
}

// The lookups of a circuit in ONE pass over the rows: the running value and the three Lagrange columns are read once instead of
// once per lookup and four launches go (five lookups over 4n rows at k = 17: 0.40 -> 0.39 ms -- the pass is bound by its 75
// multiplications per row, not by the re-reads).  Terms and their order as in k_lookup_h, lookup after lookup.
#define EVH_LOOKUP_BATCH 8
struct LookupBatchArgs {
    const fe* z[EVH_LOOKUP_BATCH]; const fe* a_perm[EVH_LOOKUP_BATCH]; const fe* s_perm[EVH_LOOKUP_BATCH]; const fe* table_value[EVH_LOOKUP_BATCH];
    u32 count;
    const fe* l0; const fe* l_last; const fe* l_active;
    const fe* scalars;         // internal packed: [beta, gamma, y]
    u32 rows_mask, rot_scale;
    fe* values;
    u64 rows;
    u32 cols_internal, vals_internal;
};
template <class F>
__global__ __launch_bounds__(EVH_THREADS) void k_lookup_h_batch(LookupBatchArgs A) {
    typedef typename f29_of<F>::type F9;
    const u64 row = (u64)blockIdx.x * EVH_THREADS + threadIdx.x;
    if (row >= A.rows) return;
    const f29 beta = f29_unpack(f_load(&A.scalars[0])), gamma = f29_unpack(f_load(&A.scalars[1])), y = f29_unpack(f_load(&A.scalars[2]));
    const f29 one = f29_one<F9>();
    const u32 r_next = ((u32)row + A.rot_scale) & A.rows_mask;
    const u32 r_prev = ((u32)row - A.rot_scale) & A.rows_mask;
    auto ld = [&](const fe* p, u64 i) __attribute__((always_inline)) { return A.cols_internal ? f29_unpack(f_load(&p[i])) : f29_from_std<F9>(f_load(&p[i])); };
    auto ldv = [&](const fe* p, u64 i) __attribute__((always_inline)) { return A.vals_internal ? f29_unpack(f_load(&p[i])) : f29_from_std<F9>(f_load(&p[i])); };
    f29 v = ldv(A.values, row);
    const f29 l0 = ld(A.l0, row), l_last = ld(A.l_last, row), l_active = ld(A.l_active, row);
    for (u32 l = 0; l < A.count; l++) {
        const fe* zc = A.z[l]; const fe* ac = A.a_perm[l]; const fe* sc = A.s_perm[l];
        const f29 z = ld(zc, row), a = ld(ac, row), s = ld(sc, row);
        const f29 z_next = ld(zc, r_next), a_prev = ld(ac, r_prev), tv = ldv(A.table_value[l], row);
        const f29 a_minus_s = evh_sub<F9>(a, s);
        v = f29_mul2<F9>(v, y, evh_sub<F9>(one, z), l0);
        v = f29_mul2<F9>(v, y, evh_sub<F9>(f29_sqr<F9>(z), z), l_last);
        f29 lhs = f29_mul<F9>(f29_mul<F9>(z_next, evh_add<F9>(a, beta)), evh_add<F9>(s, gamma));
        f29 rhs = f29_mul<F9>(z, tv);
        v = f29_mul2<F9>(v, y, evh_sub<F9>(lhs, rhs), l_active);
        v = f29_mul2<F9>(v, y, a_minus_s, l0);
        f29 t = f29_mul<F9>(a_minus_s, evh_sub<F9>(a, a_prev));
        v = f29_mul2<F9>(v, y, t, l_active);
    }
    f_store(&A.values[row], A.vals_internal ? f29_to_packed_canon<F9>(v) : f29_to_std<F9>(v));
}


// ---- the grand products' per-row factors (permutation::Argument::commit, lookup::Permuted::commit_product) --------------------
// [UPSTREAM halo2_proofs/src/plonk/permutation/prover.rs: "modified_values" -- per set of columns the denominator prod_j (value_j + beta
// sigma_j + gamma) and the numerator prod_j (value_j + delta^j beta omega^i + gamma); plonk/lookup/prover.rs commit_product: denominator
// (a' + beta)(s' + gamma), numerator (A + beta)(S + gamma)] over the n rows of the ORIGINAL domain, every product of a proof in ONE
// launch (grid.y = product: permutation sets, then lookups).  Standard-form columns in, standard-form num / den out (the inputs of
// dehalo_grand_product_batch_device).  ptrs: [cols (ncols) | sigma (ncols) | per lookup: A, S, a', s'].
struct ProductArgs {
    const fe* const* ptrs;
    const fe* omega;           // omega^i, standard form
    const fe* scalars;         // internal packed: [beta, gamma, delta, beta delta^(chunk_len s) for s < nsets]
    u32 nsets, ncols, chunk_len, nlookups;
    fe* num; fe* den;
    u64 stride, n;
};
template <class F>
__global__ __launch_bounds__(EVH_THREADS) void k_product_terms(ProductArgs A) {
    typedef typename f29_of<F>::type F9;
    const u64 row = (u64)blockIdx.x * EVH_THREADS + threadIdx.x;
    const u32 p = blockIdx.y;
    if (row >= A.n) return;
    const f29 beta = f29_unpack(f_load(&A.scalars[0])), gamma = f29_unpack(f_load(&A.scalars[1]));
    auto ld = [&](const fe* c) __attribute__((always_inline)) { return f29_from_std<F9>(f_load(&c[row])); };
    f29 num, den;
    if (p < A.nsets) {
        const f29 delta = f29_unpack(f_load(&A.scalars[2]));
        f29 cur = f29_mul<F9>(f29_unpack(f_load(&A.scalars[3 + p])), ld(A.omega));      // beta delta^(first column of the set) omega^row
        const u32 c0 = p * A.chunk_len, c1 = min(c0 + A.chunk_len, A.ncols);
        for (u32 j = c0; j < c1; j++) {
            const f29 col = ld(A.ptrs[j]);
            const f29 t = evh_add<F9>(evh_add<F9>(col, f29_mul<F9>(beta, ld(A.ptrs[A.ncols + j]))), gamma);
            const f29 u = evh_add<F9>(evh_add<F9>(col, cur), gamma);
            den = j == c0 ? t : f29_mul<F9>(den, t);
            num = j == c0 ? u : f29_mul<F9>(num, u);
            cur = f29_mul<F9>(cur, delta);
        }
        if (c0 >= c1) num = den = f29_one<F9>();
    } else {
        const fe* const* q = A.ptrs + 2 * A.ncols + 4 * (p - A.nsets);
        num = f29_mul<F9>(evh_add<F9>(ld(q[0]), beta), evh_add<F9>(ld(q[1]), gamma));
        den = f29_mul<F9>(evh_add<F9>(ld(q[2]), beta), evh_add<F9>(ld(q[3]), gamma));
    }
    f_store(&A.num[(u64)p * A.stride + row], f29_to_std<F9>(num));
    f_store(&A.den[(u64)p * A.stride + row], f29_to_std<F9>(den));
}

// ==========================================================================================
// host drivers
// ==========================================================================================
template <class F>
int graph_upload_t(dehalo_ctx* ctx, dehalo_graph* g, const uint64_t* constants, hipStream_t s) {
    // constants: standard form on the host -> internal packed on the device, once
    if (g->num_constants) {
        fe* tmp = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&tmp, (size_t)g->num_constants * sizeof(fe)));
        hipError_t e = dh_h2d(ctx, tmp, constants, (size_t)g->num_constants * sizeof(fe), s) == 0 ? hipSuccess : hipErrorUnknown;
        if (e == hipSuccess) {
            k_evh_scalars<F><<<(g->num_constants + 127) / 128, 128, 0, s>>>(tmp, g->d_constants, g->num_constants);
            e = hipStreamSynchronize(s);
        }
        (void)hipFree(tmp);
        HIP_TRY(ctx, e);
    }
    return 0;
}

template <class F>
int evh_stage_scalars(dehalo_ctx* ctx, const std::vector<fe>& host_std, fe* d_table, hipStream_t s) {
    for (size_t o = 0; o < host_std.size(); o += 8) {
        FeBatch b;
        u32 n = (u32)std::min<size_t>(8, host_std.size() - o);
        for (u32 i = 0; i < n; i++) b.v[i] = host_std[o + i];
        k_evh_scalars_val<F><<<1, 8, 0, s>>>(b, d_table + o, n);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
inline int evh_stage_ptrs(dehalo_ctx* ctx, const std::vector<const fe*>& ptrs, const void** d_table, hipStream_t s) {
    for (size_t o = 0; o < ptrs.size(); o += 32) {
        PtrBatch b;
        u32 n = (u32)std::min<size_t>(32, ptrs.size() - o);
        for (u32 i = 0; i < n; i++) b.p[i] = ptrs[o + i];
        k_evh_ptrs<<<1, 32, 0, s>>>(b, d_table + o, n);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class F>
int graph_evaluate_t(dehalo_ctx* ctx, const dehalo_graph* g, const dehalo_eval_inputs* in, uint32_t log_rows, uint32_t rot_scale, const fe* d_previous,
                     fe* d_out, hipStream_t s) {
    const u64 rows = 1ull << log_rows;
    ScopedTimer timer(ctx, s, DEHALO_K_EVAL_H);
    // scalar table: [beta, gamma, theta, y, constants..., challenges...]
    const u32 nsc = 4 + g->num_constants + in->num_challenges;
    TRY(dh_ensure(ctx, ctx->ws_evh[0], (size_t)(nsc + 8) * sizeof(fe)));
    fe* table = (fe*)ctx->ws_evh[0].p;
    const uint64_t* four[4] = {in->beta, in->gamma, in->theta, in->y};
    const u32 ncol = in->num_fixed + in->num_advice + in->num_instance;
    std::vector<const fe*> cols(ncol, nullptr);
    for (u32 i = 0; i < in->num_fixed; i++) cols[i] = (const fe*)in->fixed[i];
    for (u32 i = 0; i < in->num_advice; i++) cols[in->num_fixed + i] = (const fe*)in->advice[i];
    for (u32 i = 0; i < in->num_instance; i++) cols[in->num_fixed + in->num_advice + i] = (const fe*)in->instance[i];
    TRY(dh_ensure(ctx, ctx->ws_evh[2], (cols.size() + 1) * sizeof(void*)));
    if (in->num_challenges <= 12 && ncol <= EVH_ARG_COLS) {   // the common case: scalars, constants and column pointers staged by ONE launch
        EvhStage st{};
        for (int i = 0; i < 4; i++) st.v[i] = four[i] ? fe_from_u64(four[i]) : fe{};
        for (u32 i = 0; i < in->num_challenges; i++) st.v[4 + i] = fe_from_u64(in->challenges + 4 * (size_t)i);
        st.constants = g->d_constants; st.nconst = g->num_constants; st.nchal = in->num_challenges;
        st.ncols = ncol;
        for (u32 i = 0; i < ncol; i++) st.cols[i] = cols[i];
        const u32 work = std::max<u32>(nsc, ncol);
        k_evh_stage<F><<<(work + 63) / 64, 64, 0, s>>>(st, table, (const void**)ctx->ws_evh[2].p);
    } else {
        std::vector<fe> head(4), chal(in->num_challenges);
        for (int i = 0; i < 4; i++) head[i] = four[i] ? fe_from_u64(four[i]) : fe{};
        for (u32 i = 0; i < in->num_challenges; i++) chal[i] = fe_from_u64(in->challenges + 4 * (size_t)i);
        TRY(evh_stage_scalars<F>(ctx, head, table, s));
        if (g->num_constants) HIP_TRY(ctx, hipMemcpyAsync(table + 4, g->d_constants, (size_t)g->num_constants * sizeof(fe), hipMemcpyDeviceToDevice, s));
        TRY(evh_stage_scalars<F>(ctx, chal, table + 4 + g->num_constants, s));
        TRY(evh_stage_ptrs(ctx, cols, (const void**)ctx->ws_evh[2].p, s));
    }
    EvhArgs A{};
    A.columns = (const fe* const*)ctx->ws_evh[2].p;
    if (g->hbm_slots) TRY(dh_ensure(ctx, ctx->ws_evh[3], (size_t)g->hbm_slots * rows * sizeof(fe)));
    A.calcs = g->d_calcs; A.parts = g->d_parts; A.scalars = table;
    A.num_calcs = g->num_calcs; A.fixed_base = 0; A.advice_base = in->num_fixed; A.instance_base = in->num_fixed + in->num_advice;
    A.rows_mask = (u32)(rows - 1); A.rot_scale = rot_scale;
    A.previous = d_previous; A.out = d_out; A.spill = (fe*)ctx->ws_evh[3].p; A.rows = rows; A.result = g->result;
    A.cols_internal = in->form_flags & DEHALO_EVAL_COLUMNS_INTERNAL; A.vals_internal = in->form_flags & DEHALO_EVAL_VALUES_INTERNAL;
    const size_t lds = (size_t)std::max<u32>(1, g->lds_slots) * EVH_SLOT_BYTES;
    if (lds > 48 * 1024)   // per call: the attribute belongs to the device the context is bound to
        HIP_TRY(ctx, dh_func_lds(ctx, (const void*)k_graph_eval<F>, EVH_LDS_BYTES));
    k_graph_eval<F><<<(u32)((rows + EVH_THREADS - 1) / EVH_THREADS), EVH_THREADS, lds, s>>>(A);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// `count` programs over the same inputs, one staging launch + one evaluation launch; falls back to one pair per program when a program spills to HBM,
// or the inputs do not fit the argument blocks
template <class F>
int graph_evaluate_batch_t(dehalo_ctx* ctx, const dehalo_graph* const* graphs, uint32_t count, const dehalo_eval_inputs* in, uint32_t log_rows, uint32_t rot_scale,
                           fe* const* d_outs, hipStream_t s) {
    const u32 ncol = in->num_fixed + in->num_advice + in->num_instance;
    bool ok = count >= 2 && in->num_challenges <= 12 && ncol <= EVH_ARG_COLS;
    u32 max_consts = 0, max_lds = 1;
    for (u32 i = 0; ok && i < count; i++) {
        ok = graphs[i] && graphs[i]->hbm_slots == 0;
        if (ok) { max_consts = std::max(max_consts, graphs[i]->num_constants); max_lds = std::max(max_lds, graphs[i]->lds_slots); }
    }
    if (!ok) {
        for (u32 i = 0; i < count; i++) TRY(graph_evaluate_t<F>(ctx, graphs[i], in, log_rows, rot_scale, nullptr, d_outs[i], s));
        return 0;
    }
    const u64 rows = 1ull << log_rows;
    ScopedTimer timer(ctx, s, DEHALO_K_EVAL_H);
    const u32 tstride = 4 + max_consts + in->num_challenges + 4;
    const uint64_t* four[4] = {in->beta, in->gamma, in->theta, in->y};
    const size_t lds = (size_t)max_lds * EVH_SLOT_BYTES;
    if (lds > 48 * 1024) HIP_TRY(ctx, dh_func_lds(ctx, (const void*)k_graph_eval_batch<F>, EVH_LDS_BYTES));
    for (u32 first = 0; first < count; first += EVH_GRAPH_BATCH) {
        // (the tables of consecutive groups must not overlap in time: a group of its own region each)
        const size_t region = (size_t)(first / EVH_GRAPH_BATCH);
        TRY(dh_ensure(ctx, ctx->ws_evh[0], (region + 1) * EVH_GRAPH_BATCH * (size_t)tstride * sizeof(fe)));
        TRY(dh_ensure(ctx, ctx->ws_evh[2], (ncol + 1) * sizeof(void*) + (region + 1) * EVH_GRAPH_BATCH * sizeof(EvhArgs) + 64));
    }
    for (u32 first = 0; first < count; first += EVH_GRAPH_BATCH) {
        const u32 cnt = std::min<u32>(EVH_GRAPH_BATCH, count - first);
        const size_t region = (size_t)(first / EVH_GRAPH_BATCH);
        fe* table = (fe*)ctx->ws_evh[0].p + region * EVH_GRAPH_BATCH * (size_t)tstride;
        const void** ptrs = (const void**)ctx->ws_evh[2].p;
        EvhArgs* dargs = reinterpret_cast<EvhArgs*>(((uintptr_t)((char*)ctx->ws_evh[2].p + (ncol + 1) * sizeof(void*)) + 63) & ~(uintptr_t)63) + region * EVH_GRAPH_BATCH;
        EvhStageBatch st{};
        for (int i = 0; i < 4; i++) st.v[i] = four[i] ? fe_from_u64(four[i]) : fe{};
        for (u32 i = 0; i < in->num_challenges; i++) st.v[4 + i] = fe_from_u64(in->challenges + 4 * (size_t)i);
        st.nchal = in->num_challenges; st.ncols = ncol; st.count = cnt; st.tstride = tstride;
        for (u32 i = 0; i < in->num_fixed; i++) st.cols[i] = in->fixed[i];
        for (u32 i = 0; i < in->num_advice; i++) st.cols[in->num_fixed + i] = in->advice[i];
        for (u32 i = 0; i < in->num_instance; i++) st.cols[in->num_fixed + in->num_advice + i] = in->instance[i];
        for (u32 j = 0; j < cnt; j++) {
            const dehalo_graph* g = graphs[first + j];
            st.constants[j] = g->d_constants; st.nconst[j] = g->num_constants;
            EvhArgs& A = st.args[j];
            A.columns = (const fe* const*)ptrs;
            A.calcs = g->d_calcs; A.parts = g->d_parts; A.scalars = table + (size_t)j * tstride;
            A.num_calcs = g->num_calcs; A.fixed_base = 0; A.advice_base = in->num_fixed; A.instance_base = in->num_fixed + in->num_advice;
            A.rows_mask = (u32)(rows - 1); A.rot_scale = rot_scale;
            A.previous = nullptr; A.out = d_outs[first + j]; A.spill = nullptr; A.rows = rows; A.result = g->result;
            A.cols_internal = in->form_flags & DEHALO_EVAL_COLUMNS_INTERNAL; A.vals_internal = in->form_flags & DEHALO_EVAL_VALUES_INTERNAL;
        }
        const u32 work = std::max<u32>(tstride, ncol);
        k_evh_stage_batch<F><<<dim3((work + 63) / 64, cnt), 64, 0, s>>>(st, table, ptrs, dargs);
        k_graph_eval_batch<F><<<dim3((u32)((rows + EVH_THREADS - 1) / EVH_THREADS), cnt), EVH_THREADS, lds, s>>>(dargs);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class F>
int perm_h_t(dehalo_ctx* ctx, const dehalo_perm_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* d_values, hipStream_t s) {
    if (log_rows > (uint32_t)F::TWO_ADICITY) return dh_fail(ctx, DEHALO_ERR_UNSUPPORTED, "permutation_h: log_rows exceeds the field's two-adicity");
    const u64 rows = 1ull << log_rows;
    ScopedTimer timer(ctx, s, DEHALO_K_EVAL_H);
    std::vector<fe> sc = {fe_from_u64(in->beta), fe_from_u64(in->gamma), fe_from_u64(in->y), fe_from_u64(in->delta), fe_from_u64(in->beta_zeta)};
    TRY(dh_ensure(ctx, ctx->ws_evh[0], 16 * sizeof(fe)));
    TRY(evh_stage_scalars<F>(ctx, sc, (fe*)ctx->ws_evh[0].p, s));
    std::vector<const fe*> ptrs;
    for (u32 i = 0; i < in->num_sets; i++) ptrs.push_back((const fe*)in->z[i]);
    for (u32 i = 0; i < in->num_columns; i++) ptrs.push_back((const fe*)in->columns[i]);
    for (u32 i = 0; i < in->num_columns; i++) ptrs.push_back((const fe*)in->sigma[i]);
    TRY(dh_ensure(ctx, ctx->ws_evh[2], (ptrs.size() + 1) * sizeof(void*)));
    TRY(evh_stage_ptrs(ctx, ptrs, (const void**)ctx->ws_evh[2].p, s));
    const fe* tw = nullptr;                                   // omega_ext^j * 2^261, j < rows / 2 (shared with the NTT)
    TRY(get_twiddles<F>(ctx, log_rows, in->extended_omega, s, &tw));
    PermArgs A;
    const fe* const* base = (const fe* const*)ctx->ws_evh[2].p;
    A.z = base; A.cols = base + in->num_sets; A.sigma = base + in->num_sets + in->num_columns;
    A.l0 = (const fe*)in->l0; A.l_last = (const fe*)in->l_last; A.l_active = (const fe*)in->l_active_row;
    A.scalars = (const fe*)ctx->ws_evh[0].p;
    A.nsets = in->num_sets; A.ncols = in->num_columns; A.chunk_len = in->chunk_len;
    A.rows_mask = (u32)(rows - 1); A.rot_scale = rot_scale; A.last_rotation = in->last_rotation;
    A.values = d_values; A.rows = rows; A.tw = tw;
    A.cols_internal = in->form_flags & DEHALO_EVAL_COLUMNS_INTERNAL; A.vals_internal = in->form_flags & DEHALO_EVAL_VALUES_INTERNAL;
    k_perm_h<F><<<(u32)((rows + EVH_THREADS - 1) / EVH_THREADS), EVH_THREADS, 0, s>>>(A);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class F>
int lookup_h_t(dehalo_ctx* ctx, const dehalo_lookup_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* d_values, hipStream_t s) {
    const u64 rows = 1ull << log_rows;
    ScopedTimer timer(ctx, s, DEHALO_K_EVAL_H);
    std::vector<fe> sc = {fe_from_u64(in->beta), fe_from_u64(in->gamma), fe_from_u64(in->y)};
    TRY(dh_ensure(ctx, ctx->ws_evh[0], 16 * sizeof(fe)));
    TRY(evh_stage_scalars<F>(ctx, sc, (fe*)ctx->ws_evh[0].p, s));
    LookupArgs A;
    A.z = (const fe*)in->product_coset; A.a_perm = (const fe*)in->permuted_input_coset; A.s_perm = (const fe*)in->permuted_table_coset;
    A.table_value = (const fe*)in->table_value;
    A.l0 = (const fe*)in->l0; A.l_last = (const fe*)in->l_last; A.l_active = (const fe*)in->l_active_row;
    A.scalars = (const fe*)ctx->ws_evh[0].p;
    A.rows_mask = (u32)(rows - 1); A.rot_scale = rot_scale; A.values = d_values; A.rows = rows;
    A.cols_internal = in->form_flags & DEHALO_EVAL_COLUMNS_INTERNAL; A.vals_internal = in->form_flags & DEHALO_EVAL_VALUES_INTERNAL;
    k_lookup_h<F><<<(u32)((rows + EVH_THREADS - 1) / EVH_THREADS), EVH_THREADS, 0, s>>>(A);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <class F>
int lookup_h_batch_t(dehalo_ctx* ctx, const dehalo_lookup_inputs* in, uint32_t count, uint32_t log_rows, uint32_t rot_scale, fe* d_values, hipStream_t s) {
    const u64 rows = 1ull << log_rows;
    ScopedTimer timer(ctx, s, DEHALO_K_EVAL_H);
    std::vector<fe> sc = {fe_from_u64(in[0].beta), fe_from_u64(in[0].gamma), fe_from_u64(in[0].y)};
    TRY(dh_ensure(ctx, ctx->ws_evh[0], 16 * sizeof(fe)));
    TRY(evh_stage_scalars<F>(ctx, sc, (fe*)ctx->ws_evh[0].p, s));
    LookupBatchArgs A{};
    A.count = count;
    for (uint32_t l = 0; l < count; l++) {
        A.z[l] = (const fe*)in[l].product_coset; A.a_perm[l] = (const fe*)in[l].permuted_input_coset; A.s_perm[l] = (const fe*)in[l].permuted_table_coset;
        A.table_value[l] = (const fe*)in[l].table_value;
    }
    A.l0 = (const fe*)in[0].l0; A.l_last = (const fe*)in[0].l_last; A.l_active = (const fe*)in[0].l_active_row;
    A.scalars = (const fe*)ctx->ws_evh[0].p;
    A.rows_mask = (u32)(rows - 1); A.rot_scale = rot_scale; A.values = d_values; A.rows = rows;
    A.cols_internal = in[0].form_flags & DEHALO_EVAL_COLUMNS_INTERNAL; A.vals_internal = in[0].form_flags & DEHALO_EVAL_VALUES_INTERNAL;
    k_lookup_h_batch<F><<<(u32)((rows + EVH_THREADS - 1) / EVH_THREADS), EVH_THREADS, 0, s>>>(A);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}


template <class F>
int product_terms_t(dehalo_ctx* ctx, const dehalo_product_inputs* in, uint64_t n, fe* d_num, fe* d_den, uint64_t stride, hipStream_t s) {
    ScopedTimer timer(ctx, s, DEHALO_K_EVAL_H);
    const u32 nsets = in->num_columns ? (in->num_columns + in->chunk_len - 1) / in->chunk_len : 0;
    std::vector<fe> sc = {fe_from_u64(in->beta), fe_from_u64(in->gamma), in->delta ? fe_from_u64(in->delta) : fe{}};
    for (u32 i = 0; i < nsets; i++) sc.push_back(fe_from_u64(in->set_factors + 4 * (size_t)i));
    TRY(dh_ensure(ctx, ctx->ws_evh[0], (sc.size() + 8) * sizeof(fe)));
    TRY(evh_stage_scalars<F>(ctx, sc, (fe*)ctx->ws_evh[0].p, s));
    std::vector<const fe*> ptrs;
    for (u32 i = 0; i < in->num_columns; i++) ptrs.push_back((const fe*)in->columns[i]);
    for (u32 i = 0; i < in->num_columns; i++) ptrs.push_back((const fe*)in->sigma[i]);
    for (u32 l = 0; l < in->num_lookups; l++) {
        ptrs.push_back((const fe*)in->compressed_input[l]);
        ptrs.push_back((const fe*)in->compressed_table[l]);
        ptrs.push_back((const fe*)in->permuted_input[l]);
        ptrs.push_back((const fe*)in->permuted_table[l]);
    }
    TRY(dh_ensure(ctx, ctx->ws_evh[2], (ptrs.size() + 1) * sizeof(void*)));
    TRY(evh_stage_ptrs(ctx, ptrs, (const void**)ctx->ws_evh[2].p, s));
    ProductArgs A{};
    A.ptrs = (const fe* const*)ctx->ws_evh[2].p;
    A.omega = (const fe*)in->omega_powers;
    A.scalars = (const fe*)ctx->ws_evh[0].p;
    A.nsets = nsets; A.ncols = in->num_columns; A.chunk_len = in->chunk_len; A.nlookups = in->num_lookups;
    A.num = d_num; A.den = d_den; A.stride = stride; A.n = n;
    if (nsets + in->num_lookups)
        k_product_terms<F><<<dim3((u32)((n + EVH_THREADS - 1) / EVH_THREADS), nsets + in->num_lookups), EVH_THREADS, 0, s>>>(A);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

#define DEFINE_EVALH_ENTRY(NAME, F)                                                                                                              \
    int convert_form_##NAME(dehalo_ctx* ctx, const fe* in, fe* out, uint64_t n, int to_internal, hipStream_t s) {                                \
        if (n) k_convert_form<F><<<(u32)((n + 255) / 256), 256, 0, s>>>(in, out, n, to_internal);                                                  \
        HIP_TRY(ctx, hipGetLastError());                                                                                                         \
        return 0; }                                                                                                                              \
    int graph_upload_##NAME(dehalo_ctx* ctx, dehalo_graph* g, const uint64_t* constants, hipStream_t s) { return graph_upload_t<F>(ctx, g, constants, s); } \
    int graph_evaluate_##NAME(dehalo_ctx* ctx, const dehalo_graph* g, const dehalo_eval_inputs* in, uint32_t log_rows, uint32_t rot_scale,      \
                              const fe* prev, fe* out, hipStream_t s) { return graph_evaluate_t<F>(ctx, g, in, log_rows, rot_scale, prev, out, s); } \
    int graph_evaluate_batch_##NAME(dehalo_ctx* ctx, const dehalo_graph* const* graphs, uint32_t count, const dehalo_eval_inputs* in, uint32_t log_rows,     \
                                    uint32_t rot_scale, fe* const* outs, hipStream_t s) { return graph_evaluate_batch_t<F>(ctx, graphs, count, in, log_rows, rot_scale, outs, s); } \
    int perm_h_##NAME(dehalo_ctx* ctx, const dehalo_perm_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s) {             \
        return perm_h_t<F>(ctx, in, log_rows, rot_scale, v, s); }                                                                                \
    int lookup_h_##NAME(dehalo_ctx* ctx, const dehalo_lookup_inputs* in, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s) {         \
        return lookup_h_t<F>(ctx, in, log_rows, rot_scale, v, s); }                                                                              \
    int lookup_h_batch_##NAME(dehalo_ctx* ctx, const dehalo_lookup_inputs* in, uint32_t count, uint32_t log_rows, uint32_t rot_scale, fe* v, hipStream_t s) { \
        return lookup_h_batch_t<F>(ctx, in, count, log_rows, rot_scale, v, s); }                                                                 \
    int product_terms_##NAME(dehalo_ctx* ctx, const dehalo_product_inputs* in, uint64_t n, fe* num, fe* den, uint64_t stride, hipStream_t s) {    \
        return product_terms_t<F>(ctx, in, n, num, den, stride, s); }
