// witness.hip -- Circuit::synthesize of the reference's three circuits as VALUES, in C++: what fills the advice columns (and, at keygen,
// the fixed columns and the permutation) before create_proof commits them -- the part of the reference's timed call that precedes the
// first commitment [src/lib.rs:164-318 DelayEncryptCircuit::synthesize; benches/mod_pow.rs:79-120 RSACircuit; src/encryption/chip.rs:
// 131-204 PoseidonEncCircuit].  Restated from:
//   big_pow_mod (its value: read off the pow_mod rows)   src/big_integer/utils.rs:2-17
//   BigIntChip                                    src/big_integer/chip.rs: assign_integer :64-85, assign_constant :1255-1285, max_value :141-157, add :250-300,
//                                                 sub :313-376, mul :389-422, mul_mod :545-632, pow_mod :667-699, is_equal_fresh :778-803, is_equal_muled :825-898,
//                                                 is_less_than :911-923, assert_in_field :1153-1161, sub_unchecked :1290-1322, div_mod_main_gate :1327-1353
//   RSAChip::{assign_public_key, modpow_public_key}   src/rsa/chip.rs:61-73, 102-117
//   Grain LFSR, Cauchy MDS, optimised constants and sparse matrices, the optimised permutation   src/poseidon/grain.rs:12-157, spec.rs:170-180, 325-397, permutation.rs:7-46
//   PoseidonChip rows                             src/poseidon/chip.rs:199-419
//   sponge (RATE 4) and cipher                    src/hash/chip.rs:63-85, src/encryption/chip.rs:72-110, src/encryption/poseidon_enc.rs:66-133, src/lib.rs:216-316
// The cell layout follows halo2wrong's MainGate / RangeChip instruction by instruction ([UPSTREAM] maingate v2023_04_20, not in the reference tree: one `apply` row
// per instruction, compose / decompose four terms a row, is_equal four rows, is_zero three, assert_equal one), which reproduces the row counts the reference
// publishes (benches/README.md:56-99; tests/test_witness.py): the same rows, in the same order, as dehalo2_amd/witness.py, which the tests compare it with
// bit for bit.  Host code only (no device work): big-integer arithmetic on 64-bit limbs, field arithmetic through hostfield.hpp.
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <stdexcept>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/dehalo.h"
#include "hostfield.hpp"

namespace {

typedef unsigned __int128 u128;
constexpr int LIMB_WIDTH = 64;

// ---- big integers: little-endian 64-bit limbs ----
typedef std::vector<uint64_t> Big;

void big_trim(Big& a) { while (!a.empty() && a.back() == 0) a.pop_back(); }
int big_cmp(const Big& a, const Big& b) {
    size_t na = a.size(), nb = b.size();
    while (na && a[na - 1] == 0) na--;
    while (nb && b[nb - 1] == 0) nb--;
    if (na != nb) return na < nb ? -1 : 1;
    for (size_t i = na; i-- > 0;)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
Big big_mul(const Big& a, const Big& b) {
    Big r(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); i++) {
        u128 c = 0;
        for (size_t j = 0; j < b.size(); j++) {
            c += (u128)a[i] * b[j] + r[i + j];
            r[i + j] = (uint64_t)c;
            c >>= 64;
        }
        r[i + b.size()] = (uint64_t)c;
    }
    return r;
}
// Knuth algorithm D: (q, r) = divmod(u, v), v != 0
void big_divmod(const Big& u_in, const Big& v_in, Big& q, Big& r) {
    Big u = u_in, v = v_in;
    big_trim(u);
    big_trim(v);
    if (big_cmp(u, v) < 0) {
        q.clear();
        r = u;
        return;
    }
    const size_t n = v.size(), m = u.size() - n;
    if (n == 1) {
        q.assign(u.size(), 0);
        u128 rem = 0;
        for (size_t i = u.size(); i-- > 0;) {
            const u128 cur = (rem << 64) | u[i];
            q[i] = (uint64_t)(cur / v[0]);
            rem = cur % v[0];
        }
        r.assign(1, (uint64_t)rem);
        big_trim(q);
        big_trim(r);
        return;
    }
    const int s = __builtin_clzll(v[n - 1]);
    Big vn(n), un(u.size() + 1);
    for (size_t i = n - 1; i > 0; i--) vn[i] = s ? (v[i] << s) | (v[i - 1] >> (64 - s)) : v[i];
    vn[0] = v[0] << s;
    un[u.size()] = s ? u[u.size() - 1] >> (64 - s) : 0;
    for (size_t i = u.size() - 1; i > 0; i--) un[i] = s ? (u[i] << s) | (u[i - 1] >> (64 - s)) : u[i];
    un[0] = u[0] << s;
    q.assign(m + 1, 0);
    for (size_t j = m + 1; j-- > 0;) {
        const u128 num = ((u128)un[j + n] << 64) | un[j + n - 1];
        u128 qhat = num / vn[n - 1], rhat = num % vn[n - 1];
        while ((qhat >> 64) || (uint64_t)qhat * (u128)vn[n - 2] > ((rhat << 64) | un[j + n - 2])) {
            qhat--;
            rhat += vn[n - 1];
            if (rhat >> 64) break;
        }
        // multiply and subtract
        u128 borrow = 0, carry = 0;
        for (size_t i = 0; i < n; i++) {
            const u128 p = (u128)(uint64_t)qhat * vn[i] + carry;
            carry = p >> 64;
            const u128 t = (u128)un[i + j] - (uint64_t)p - borrow;
            un[i + j] = (uint64_t)t;
            borrow = (t >> 64) & 1;
        }
        const u128 t = (u128)un[j + n] - carry - borrow;
        un[j + n] = (uint64_t)t;
        if ((t >> 64) & 1) {      // qhat was one too large: add back
            qhat--;
            u128 c = 0;
            for (size_t i = 0; i < n; i++) {
                c += (u128)un[i + j] + vn[i];
                un[i + j] = (uint64_t)c;
                c >>= 64;
            }
            un[j + n] += (uint64_t)c;
        }
        q[j] = (uint64_t)qhat;
    }
    r.assign(n, 0);
    for (size_t i = 0; i < n; i++) r[i] = s ? (un[i] >> s) | (un[i + 1] << (64 - s)) : un[i];
    big_trim(q);
    big_trim(r);
}
Big big_limbs(const Big& a, size_t n) {
    Big r(n, 0);
    for (size_t i = 0; i < std::min(n, a.size()); i++) r[i] = a[i];
    return r;
}

// ---- field values: canonical 4 x u64 ----
struct Fld {
    const HostField* f;
    Fe zero() const { return Fe{{0, 0, 0, 0}}; }
    Fe u(uint64_t x) const { return Fe{{x, 0, 0, 0}}; }
    Fe add(const Fe& a, const Fe& b) const { return f->add(a, b); }
    Fe sub(const Fe& a, const Fe& b) const { return f->sub(a, b); }
    Fe neg(const Fe& a) const { return f->neg(a); }
    Fe mul(const Fe& a, const Fe& b) const {
        // most products of these circuits are limb x limb or limb x word (64 x 64, 128 x 64 bits): below 2^192 < p, the integer product IS the field product
        const int la = a.v[3] ? 4 : a.v[2] ? 3 : a.v[1] ? 2 : 1, lb = b.v[3] ? 4 : b.v[2] ? 3 : b.v[1] ? 2 : 1;
        if (la + lb <= 3) {
            const Fe& x = la >= lb ? a : b;      // x: up to two limbs, y: one
            const uint64_t y = la >= lb ? b.v[0] : a.v[0];
            const u128 p0 = (u128)x.v[0] * y, p1 = (u128)x.v[1] * y + (uint64_t)(p0 >> 64);
            return Fe{{(uint64_t)p0, (uint64_t)p1, (uint64_t)(p1 >> 64), 0}};
        }
        return f->mul(f->mul(a, b), f->r2);      // (a b / R) R^2 / R = a b
    }
    Fe inv(const Fe& a) const {      // canonical inverse: to Montgomery, invert, back
        if (a.is_zero()) return a;
        return f->to_canonical(f->invert(f->from_canonical(a)));
    }
    Fe pow2(unsigned k) const {      // 2^k mod p, k < 256
        Fe r{{0, 0, 0, 0}};
        r.v[k >> 6] = (uint64_t)1 << (k & 63);
        while (HostField::geq(r.v, f->p)) HostField::sub_limbs(r.v, r.v, f->p);
        return r;
    }
    Fe from_u128(u128 x) const { return Fe{{(uint64_t)x, (uint64_t)(x >> 64), 0, 0}}; }
};

// ---- Poseidon (native) ----
struct Grain {
    const HostField* f;
    std::vector<uint8_t> bits;      // 80-bit state as a sliding window over a growing vector
    size_t head = 0;
    Grain(const HostField* field, uint32_t t, uint32_t r_f, uint32_t r_p) : f(field) {
        auto put = [&](unsigned width, uint32_t v) { for (int i = (int)width - 1; i >= 0; i--) bits.push_back((v >> i) & 1); };
        put(2, 1); put(4, 0); put(12, f->bits); put(12, t); put(10, r_f); put(10, r_p); put(30, (1u << 30) - 1);
        for (int i = 0; i < 160; i++) new_bit();
    }
    int new_bit() {
        const uint8_t* b = bits.data() + head;
        const uint8_t nb = b[0] ^ b[62] ^ b[51] ^ b[38] ^ b[23] ^ b[13];
        head++;
        bits.push_back(nb);
        if (head > (1u << 16)) {      // compact
            bits.erase(bits.begin(), bits.begin() + (long)head);
            head = 0;
        }
        return nb;
    }
    int bit() {
        while (!new_bit()) new_bit();
        return new_bit();
    }
    Fe draw() {      // nbits bits, MSB first
        Fe v{{0, 0, 0, 0}};
        for (uint32_t i = 0; i < f->bits; i++) {
            for (int j = 3; j > 0; j--) v.v[j] = (v.v[j] << 1) | (v.v[j - 1] >> 63);
            v.v[0] = (v.v[0] << 1) | (uint64_t)bit();
        }
        return v;
    }
    Fe field_element() {      // with rejection: round constants
        for (;;) {
            const Fe v = draw();
            if (!HostField::geq(v.v, f->p)) return v;
        }
    }
    Fe field_element_mod() {      // without rejection: the MDS x, y (value < 2^bits < 2 p: one subtraction)
        Fe v = draw();
        while (HostField::geq(v.v, f->p)) HostField::sub_limbs(v.v, v.v, f->p);
        return v;
    }
};

// Spec::new(r_f, r_p) (src/poseidon/spec.rs:310-397): Grain's round constants and the Cauchy MDS, then the optimised form the chip's rows follow --
// start / partial / end constants, the pre-sparse matrix, one sparse matrix (first row, first column) per partial round.
typedef std::vector<std::vector<Fe>> Mat;
struct PoseidonSpec {
    Fld F;
    uint32_t t, r_f, r_p;
    Mat constants, mds;                       // Grain's (canonical)
    Mat start, end, pre_sparse;               // optimised constants; the transition matrix
    std::vector<Fe> partial;
    std::vector<std::vector<Fe>> sparse_row, sparse_col;      // per partial round: the sparse matrix's first row (t) and first column below it (t - 1)
    Mat mds_m, pre_sparse_m;                  // the same matrices times R (Montgomery form): ONE Montgomery product gives m * x for a canonical x
    std::vector<std::vector<Fe>> sparse_row_m, sparse_col_m;
    static Mat mat_mul(const Fld& F, const Mat& a, const Mat& b) {
        const size_t n = a.size();
        Mat r(n, std::vector<Fe>(n, F.zero()));
        for (size_t i = 0; i < n; i++) for (size_t j = 0; j < n; j++) for (size_t k = 0; k < n; k++) r[i][j] = F.add(r[i][j], F.mul(a[i][k], b[k][j]));
        return r;
    }
    static std::vector<Fe> mat_vec(const Fld& F, const Mat& a, const std::vector<Fe>& v) {
        std::vector<Fe> r(a.size(), F.zero());
        for (size_t i = 0; i < a.size(); i++) for (size_t j = 0; j < v.size(); j++) r[i] = F.add(r[i], F.mul(a[i][j], v[j]));
        return r;
    }
    static Mat transpose(const Mat& a) {
        Mat r(a.size(), std::vector<Fe>(a.size()));
        for (size_t i = 0; i < a.size(); i++) for (size_t j = 0; j < a.size(); j++) r[j][i] = a[i][j];
        return r;
    }
    static Mat mat_inv(const Fld& F, const Mat& a) {      // Gauss-Jordan (src/poseidon/matrix.rs:84-121 computes the same inverse)
        const size_t n = a.size();
        Mat m(n, std::vector<Fe>(2 * n, F.zero()));
        for (size_t i = 0; i < n; i++) {
            for (size_t j = 0; j < n; j++) m[i][j] = a[i][j];
            m[i][n + i] = F.u(1);
        }
        for (size_t i = 0; i < n; i++) {
            size_t piv = i;
            while (piv < n && m[piv][i].is_zero()) piv++;
            if (piv == n) throw std::runtime_error("singular matrix");
            std::swap(m[i], m[piv]);
            const Fe inv = F.inv(m[i][i]);
            for (auto& x : m[i]) x = F.mul(x, inv);
            for (size_t r = 0; r < n; r++) {
                if (r == i || m[r][i].is_zero()) continue;
                const Fe f = m[r][i];
                for (size_t c = 0; c < 2 * n; c++) m[r][c] = F.sub(m[r][c], F.mul(f, m[i][c]));
            }
        }
        Mat r(n, std::vector<Fe>(n));
        for (size_t i = 0; i < n; i++) for (size_t j = 0; j < n; j++) r[i][j] = m[i][n + j];
        return r;
    }
    Mat to_mont(const Mat& a) const {
        Mat r = a;
        for (auto& row : r) for (auto& m : row) m = F.f->from_canonical(m);
        return r;
    }
    PoseidonSpec(const HostField* f, uint32_t t_, uint32_t rf, uint32_t rp) : F{f}, t(t_), r_f(rf), r_p(rp) {
        Grain g(f, t, r_f, r_p);
        constants.assign(r_f + r_p, std::vector<Fe>(t));
        for (auto& row : constants) for (auto& c : row) c = g.field_element();
        std::vector<Fe> xs(t), ys(t);
        for (auto& x : xs) x = g.field_element_mod();
        for (auto& y : ys) y = g.field_element_mod();
        mds.assign(t, std::vector<Fe>(t));
        for (uint32_t i = 0; i < t; i++) for (uint32_t j = 0; j < t; j++) mds[i][j] = F.inv(F.add(xs[i], ys[j]));
        // calculate_optimized_constants, spec.rs:325-378
        const uint32_t half = r_f / 2;
        const Mat inv = mat_inv(F, mds);
        start.push_back(constants[0]);
        for (uint32_t i = 1; i < half; i++) start.push_back(mat_vec(F, inv, constants[i]));
        std::vector<Fe> acc = constants[half + r_p];
        partial.assign(r_p, F.zero());
        for (uint32_t i = r_p; i-- > 0;) {      // constants[half .. half + r_p) walked backwards
            std::vector<Fe> tmp = mat_vec(F, inv, acc);
            partial[i] = tmp[0];
            tmp[0] = F.zero();
            for (uint32_t j = 0; j < t; j++) acc[j] = F.add(tmp[j], constants[half + i][j]);
        }
        start.push_back(mat_vec(F, inv, acc));
        for (uint32_t i = half + r_p + 1; i < r_f + r_p; i++) end.push_back(mat_vec(F, inv, constants[i]));
        // calculate_sparse_matrices, spec.rs:380-397 with factorise :203-241
        const Mat mds_t = transpose(mds);
        Mat acc_m = mds_t;
        for (uint32_t r = 0; r < r_p; r++) {
            std::vector<Fe> w(t - 1);
            Mat hat(t - 1, std::vector<Fe>(t - 1));
            for (uint32_t i = 1; i < t; i++) {
                w[i - 1] = acc_m[i][0];
                for (uint32_t j = 1; j < t; j++) hat[i - 1][j - 1] = acc_m[i][j];
            }
            const std::vector<Fe> w_hat = mat_vec(F, mat_inv(F, hat), w);
            Mat prime(t, std::vector<Fe>(t, F.zero()));
            prime[0][0] = F.u(1);
            for (uint32_t i = 1; i < t; i++) for (uint32_t j = 1; j < t; j++) prime[i][j] = hat[i - 1][j - 1];
            std::vector<Fe> row = {acc_m[0][0]};
            row.insert(row.end(), w_hat.begin(), w_hat.end());
            sparse_row.push_back(row);
            sparse_col.push_back(std::vector<Fe>(acc_m[0].begin() + 1, acc_m[0].end()));
            acc_m = mat_mul(F, mds_t, prime);
        }
        std::reverse(sparse_row.begin(), sparse_row.end());
        std::reverse(sparse_col.begin(), sparse_col.end());
        pre_sparse = transpose(acc_m);
        mds_m = to_mont(mds);
        pre_sparse_m = to_mont(pre_sparse);
        sparse_row_m = to_mont(sparse_row);
        sparse_col_m = to_mont(sparse_col);
    }
    Fe pow5(const Fe& x) const {
        const Fe x2 = F.mul(x, x);
        return F.mul(F.mul(x2, x2), x);
    }
    std::vector<Fe> apply_m(const Mat& m_mont, const std::vector<Fe>& st) const {
        std::vector<Fe> nx(t, F.zero());
        for (uint32_t i = 0; i < t; i++) for (uint32_t j = 0; j < t; j++) nx[i] = F.add(nx[i], F.f->mul(m_mont[i][j], st[j]));
        return nx;
    }
    std::vector<Fe> permute(std::vector<Fe> st) const {      // src/poseidon/permutation.rs:7-46 (the optimised permutation)
        const uint32_t half = r_f / 2;
        for (uint32_t i = 0; i < t; i++) st[i] = F.add(st[i], start[0][i]);
        for (uint32_t r = 1; r < half; r++) {
            for (uint32_t i = 0; i < t; i++) st[i] = F.add(pow5(st[i]), start[r][i]);
            st = apply_m(mds_m, st);
        }
        for (uint32_t i = 0; i < t; i++) st[i] = F.add(pow5(st[i]), start[half][i]);
        st = apply_m(pre_sparse_m, st);
        for (uint32_t r = 0; r < r_p; r++) {
            st[0] = F.add(pow5(st[0]), partial[r]);
            std::vector<Fe> nx(t, F.zero());
            for (uint32_t j = 0; j < t; j++) nx[0] = F.add(nx[0], F.f->mul(sparse_row_m[r][j], st[j]));
            for (uint32_t i = 1; i < t; i++) nx[i] = F.add(F.f->mul(sparse_col_m[r][i - 1], st[0]), st[i]);
            st = nx;
        }
        for (uint32_t r = 0; r < end.size(); r++) {
            for (uint32_t i = 0; i < t; i++) st[i] = F.add(pow5(st[i]), end[r][i]);
            st = apply_m(mds_m, st);
        }
        for (uint32_t i = 0; i < t; i++) st[i] = pow5(st[i]);
        return apply_m(mds_m, st);
    }
};

// The parameters depend on (T, R_F, R_P) only -- 325 Grain draws, 58 matrix inversions at the reference's parameters, a few milliseconds, many times a
// PoseidonEnc witness -- so a process keeps each parameter set it has used (the reference rebuilds the Spec in every synthesize: src/poseidon/spec.rs).
std::shared_ptr<const PoseidonSpec> poseidon_spec(const HostField* f, uint32_t t, uint32_t r_f, uint32_t r_p) {
    static std::mutex mu;
    static std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::shared_ptr<const PoseidonSpec>> cache;
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= 16) cache.clear();      // (callers hold their own reference)
    auto& slot = cache[std::make_tuple(t, r_f, r_p)];
    if (!slot) slot = std::make_shared<const PoseidonSpec>(f, t, r_f, r_p);
    return slot;
}

// ---- the MainGate / RangeChip layouter (dehalo2_amd/witness.py Layouter, row for row): every MainGateInstructions / RangeInstructions call the
// reference makes, laid out the way [UPSTREAM] halo2wrong maingate's `apply` lays it out (term i in column i) ----
enum { MG_SA = 0, MG_SB, MG_SC, MG_SD, MG_SE, MG_MUL_AB, MG_MUL_CD, MG_NEXT, MG_CONST, RC_T_TAG, RC_T_VALUE, RC_TAG_COMPOSITION, RC_TAG_OVERFLOW, RC_S_COMPOSITION, RC_S_OVERFLOW, NUM_FIX };
// RangeChip::configure(composition_bit_lens = [8, 1, 8, 4], overflow_bit_lens = [0, 0, 6]) (src/lib.rs:144-149): the distinct non-zero lengths in
// ascending order carry the tags 1..4
// -- {1, 4, 6, 8} for the reference's 2048-bit modulus.  Only the carries' overflow length depends on the modulus length (compute_range_lens(num_limbs):
// 2 x (num_limbs (2^64 - 1)^2 + 2^64 - 1) has 134 bits at 32 limbs -- 70-bit carries, a 6-bit overflow limb --, 133 at 16 limbs -- 5 bits).
struct RangeLens {
    unsigned bits[5] = {0, 0, 0, 0, 0};
    unsigned count = 0;
    explicit RangeLens(size_t num_limbs) {
        // bit length of 2 * word_max = 1 + bit length of num_limbs * (2^64 - 1)^2 + (2^64 - 1), which is 128 + bit length of num_limbs, less one when num_limbs is
        // a power of two (num_limbs * (2^64 - 1)^2 + 2^64 - 1 < num_limbs * 2^128 for every num_limbs >= 1)
        unsigned bl = 0;
        for (size_t v = num_limbs; v; v >>= 1) bl++;
        const bool pow2 = num_limbs && !(num_limbs & (num_limbs - 1));
        const unsigned wm_bits = 1 + 128 + bl - (pow2 ? 1 : 0);
        const unsigned carry_bits = wm_bits - 64, comp = std::max(1u, carry_bits / 8), over = carry_bits % comp;
        unsigned cand[5] = {1, 4, 8, comp, over};
        std::sort(cand, cand + 5);
        for (unsigned c : cand)
            if (c && (count == 0 || bits[count - 1] != c)) bits[count++] = c;
    }
    uint64_t tag(unsigned b) const {
        for (unsigned i = 0; i < count; i++)
            if (bits[i] == b) return i + 1;
        return 0;
    }
};
constexpr unsigned NUM_LOOKUP_LIMBS = 8;      // src/big_integer/chip.rs:1167
inline unsigned sublimb_bit_len(unsigned bits) { return std::max(1u, bits / NUM_LOOKUP_LIMBS); }

struct NotSatisfied : std::runtime_error { using std::runtime_error::runtime_error; };      // a constraint of the reference's circuit fails for these inputs

struct Cell { int col = -1; uint32_t row = 0; Fe val{}; bool is_cell() const { return col >= 0; } };
struct Arg {      // a cell (copied), a value, or nothing (0)
    Cell c;
    Arg() { c.val = Fe{{0, 0, 0, 0}}; }
    Arg(const Cell& cell) : c(cell) {}
    Arg(const Fe& v) { c.val = v; }
};
struct Sel { int col; Fe val; };
struct Copy { uint32_t c0, r0, c1, r1; };

struct Layouter {
    Fld F;
    bool want_fixed;                 // fixed columns and copies are only needed at keygen
    // the advice columns are written where they are wanted: straight into the caller's 2^k-row columns (or into `own` when the caller only asks for
    // the summary); rows past the capacity are counted, not stored -- the caller's "not enough rows" check then refuses the circuit
    Fe* adv[5];
    size_t cap;
    uint32_t nrows = 0;
    // Proving only: what follows the exponentiation in the circuit (the expected value's rows, the hash and cipher regions) reads nothing of it but its
    // VALUE, which the integer chain gives before any row is written -- so pow_mod may run it as one more task beside the multiplication regions, through a
    // cursor at the row where the exponentiation will end.  Set by dehalo_synthesize; `after_pow_done` tells the sequential code behind pow_mod to skip it.
    std::function<void(Layouter&, const std::vector<Cell>&)> after_pow;
    bool after_pow_done = false;
    std::vector<Fe> own;
    std::vector<Fe> fix[NUM_FIX];
    std::vector<Copy> copies;
    Fe one_, m1_;
    RangeLens range{32};             // the RangeChip table this circuit configures (dehalo_synthesize sets it from the modulus length)
    uint64_t range_tag(unsigned bits) const { return range.tag(bits); }
    Layouter(const HostField* f, bool fixed_too, uint64_t* advice, size_t n) : F{f}, want_fixed(fixed_too), cap(n) {
        if (!advice) own.resize(5 * n);
        Fe* base = advice ? reinterpret_cast<Fe*>(advice) : own.data();
        for (int i = 0; i < 5; i++) adv[i] = base + (size_t)i * n;
        one_ = F.u(1); m1_ = F.neg(one_);
    }
    // a second cursor over the same columns, starting at row `start` (proving only: no fixed columns, no copies): the rows of independent regions whose
    // positions are known in advance are written by several host threads at once (BigIntChip::pow_mod)
    Layouter(const Layouter& parent, uint32_t start) : F(parent.F), want_fixed(false), cap(parent.cap), nrows(start), one_(parent.one_), m1_(parent.m1_), range(parent.range) {
        for (int i = 0; i < 5; i++) adv[i] = parent.adv[i];
    }
    uint32_t rows() const { return nrows; }

    void row(const Arg* cells, int ncells, const Sel* sel, int nsel, Cell out[5]) {
        const uint32_t r = nrows++;
        const bool store = r < cap;
        for (int i = 0; i < 5; i++) {
            Fe v{{0, 0, 0, 0}};
            if (i < ncells) {
                v = cells[i].c.val;
                if (cells[i].c.is_cell() && want_fixed) copies.push_back(Copy{(uint32_t)cells[i].c.col, cells[i].c.row, (uint32_t)i, r});
            }
            if (store) adv[i][r] = v;
            out[i].col = i; out[i].row = r; out[i].val = v;
        }
        if (want_fixed) {
            for (auto& col : fix) col.push_back(Fe{{0, 0, 0, 0}});
            for (int i = 0; i < nsel; i++) fix[sel[i].col][r] = sel[i].val;
        }
    }
    const Fe& one() const { return one_; }
    const Fe& m1() const { return m1_; }
    [[noreturn]] void fail(const char* what) const { throw NotSatisfied(std::string(what) + " at row " + std::to_string(nrows)); }

    // --- MainGateInstructions, one row each unless said otherwise
    Cell assign_value(const Fe& v) { Cell o[5]; Arg a[1] = {Arg(v)}; row(a, 1, nullptr, 0, o); return o[0]; }
    Cell assign_constant(const Fe& c) { Cell o[5]; Arg a[1] = {Arg(c)}; Sel s[2] = {{MG_SA, m1()}, {MG_CONST, c}}; row(a, 1, s, 2, o); return o[0]; }      // -a + c = 0
    Cell assign_bit(uint64_t b) {      // a b - c = 0 with a = b = c
        Cell o[5];
        const Fe v = F.u(b);
        Arg x[3] = {Arg(v), Arg(v), Arg(v)};
        Sel s[2] = {{MG_MUL_AB, one()}, {MG_SC, m1()}};
        row(x, 3, s, 2, o);
        if (want_fixed) { copies.push_back(Copy{0, o[0].row, 1, o[0].row}); copies.push_back(Copy{1, o[0].row, 2, o[0].row}); }
        return o[2];
    }
    void assert_equal(const Cell& a, const Cell& b) {      // a - b = 0
        if (!(a.val == b.val)) fail("assert_equal on different values");
        Cell o[5];
        Arg x[2] = {Arg(a), Arg(b)};
        Sel s[2] = {{MG_SA, one()}, {MG_SB, m1()}};
        row(x, 2, s, 2, o);
    }
    void assert_one(const Cell& a) {
        if (!(a.val == one())) fail("assert_one fails");
        Cell o[5];
        Arg x[1] = {Arg(a)};
        Sel s[2] = {{MG_SA, one()}, {MG_CONST, m1()}};
        row(x, 1, s, 2, o);
    }
    Cell add_with_constant(const Cell& a, const Cell& b, const Fe& constant) {
        Cell o[5];
        Arg x[3] = {Arg(a), Arg(b), Arg(F.add(F.add(a.val, b.val), constant))};
        Sel s[4] = {{MG_SA, one()}, {MG_SB, one()}, {MG_SC, m1()}, {MG_CONST, constant}};
        row(x, 3, s, 4, o);
        return o[2];
    }
    Cell add(const Cell& a, const Cell& b) { return add_with_constant(a, b, F.zero()); }
    Cell sub(const Cell& a, const Cell& b) {
        Cell o[5];
        Arg x[3] = {Arg(a), Arg(b), Arg(F.sub(a.val, b.val))};
        Sel s[3] = {{MG_SA, one()}, {MG_SB, m1()}, {MG_SC, m1()}};
        row(x, 3, s, 3, o);
        return o[2];
    }
    Cell add_constant(const Cell& a, const Fe& constant) {
        Cell o[5];
        Arg x[2] = {Arg(a), Arg(F.add(a.val, constant))};
        Sel s[3] = {{MG_SA, one()}, {MG_SB, m1()}, {MG_CONST, constant}};
        row(x, 2, s, 3, o);
        return o[1];
    }
    Cell mul(const Arg& a, const Arg& b) {
        Cell o[5];
        Arg x[3] = {a, b, Arg(F.mul(a.c.val, b.c.val))};
        Sel s[2] = {{MG_MUL_AB, one()}, {MG_SC, m1()}};
        row(x, 3, s, 2, o);
        return o[2];
    }
    Cell mul_add(const Arg& a, const Arg& b, const Arg& c) {
        Cell o[5];
        Arg x[4] = {a, b, c, Arg(F.add(F.mul(a.c.val, b.c.val), c.c.val))};
        Sel s[3] = {{MG_MUL_AB, one()}, {MG_SC, one()}, {MG_SD, m1()}};
        row(x, 4, s, 3, o);
        return o[3];
    }
    Cell mul_add_constant(const Cell& a, const Cell& b, const Fe& constant) {
        Cell o[5];
        Arg x[3] = {Arg(a), Arg(b), Arg(F.add(F.mul(a.val, b.val), constant))};
        Sel s[3] = {{MG_MUL_AB, one()}, {MG_SC, m1()}, {MG_CONST, constant}};
        row(x, 3, s, 3, o);
        return o[2];
    }
    Cell and_(const Cell& a, const Cell& b) { return mul(Arg(a), Arg(b)); }
    Cell not_(const Cell& c) {      // c + not_c - 1 = 0
        Cell o[5];
        Arg x[2] = {Arg(c), Arg(F.sub(one(), c.val))};
        Sel s[3] = {{MG_SA, one()}, {MG_SB, one()}, {MG_CONST, m1()}};
        row(x, 2, s, 3, o);
        return o[1];
    }
    Cell select(const Cell& a, const Cell& b, const Cell& cond) {      // cond a - cond b + b - res = 0, columns | cond | a | cond | b | res |
        Cell o[5];
        Arg x[5] = {Arg(cond), Arg(a), Arg(cond), Arg(b), Arg(cond.val.is_zero() ? b.val : a.val)};
        Sel s[4] = {{MG_MUL_AB, one()}, {MG_MUL_CD, m1()}, {MG_SD, one()}, {MG_SE, m1()}};
        row(x, 5, s, 4, o);
        return o[4];
    }
    Cell is_equal(const Cell& a, const Cell& b) {      // four rows: r (a bit), dif = a - b, u = r - r x + x, dif u + r - 1 = 0  (x = 1 / dif, or 1 when dif = 0)
        const Fe dv = F.sub(a.val, b.val);
        const bool eq = dv.is_zero();
        const Fe x = eq ? one() : F.inv(dv);
        const Cell r = assign_bit(eq ? 1 : 0);
        const Cell dif = sub(a, b);
        Cell o[5], o2[5];
        Arg t[5] = {Arg(r), Arg(x), Arg(r), Arg(x), Arg(x)};      // u = r - r x + x = x in both cases (r = 0: x; r = 1: x = 1)
        Sel s[4] = {{MG_MUL_AB, one()}, {MG_SC, m1()}, {MG_SD, m1()}, {MG_SE, one()}};
        row(t, 5, s, 4, o);
        Arg t2[3] = {Arg(dif), Arg(o[4]), Arg(r)};
        Sel s2[3] = {{MG_MUL_AB, one()}, {MG_SC, one()}, {MG_CONST, m1()}};
        row(t2, 3, s2, 3, o2);
        return r;
    }
    Cell is_zero(const Cell& a) {      // MainGate::invert's flag, three rows: r (a bit), a a' + r - 1 = 0, r a' - r = 0
        const bool z = a.val.is_zero();
        const Fe a_inv = z ? one() : F.inv(a.val);
        const Cell r = assign_bit(z ? 1 : 0);
        Cell o[5], o2[5];
        Arg t[3] = {Arg(a), Arg(a_inv), Arg(r)};
        Sel s[3] = {{MG_MUL_AB, one()}, {MG_SC, one()}, {MG_CONST, m1()}};
        row(t, 3, s, 3, o);
        Arg t2[3] = {Arg(r), Arg(o[1]), Arg(r)};
        Sel s2[2] = {{MG_MUL_AB, one()}, {MG_SC, m1()}};
        row(t2, 3, s2, 2, o2);
        return r;
    }
    // compose: four terms a row, column e holds what is still to be added (the total in the first row).  prod[i] = coeff[i] * cells[i].val (the caller's
    // cheapest way to it); coeff is only read at keygen
    Cell compose(const Cell* cells, const Fe* coeff, const Fe* prod, int n, const Fe& constant) {
        Fe remaining = constant;
        for (int i = 0; i < n; i++) remaining = F.add(remaining, prod[i]);
        Cell result, o[5];
        for (int g = 0; g < n; g += 4) {
            const int cnt = std::min(4, n - g);
            const bool last = g + 4 >= n;
            Arg a[5];
            Sel s[8];
            int ns = 0;
            for (int i = 0; i < cnt; i++) {
                a[i] = Arg(cells[g + i]);
                if (want_fixed) s[ns++] = {MG_SA + i, coeff[g + i]};
            }
            a[4] = Arg(remaining);
            s[ns++] = {MG_SE, m1()};
            if (!last) s[ns++] = {MG_NEXT, one()};
            if (g == 0) s[ns++] = {MG_CONST, constant};
            row(a, 5, s, ns, o);
            if (g == 0) { result = o[4]; remaining = F.sub(remaining, constant); }
            for (int i = 0; i < cnt; i++) remaining = F.sub(remaining, prod[g + i]);
        }
        return result;
    }
    std::vector<Cell> to_bits(const Cell& v, unsigned nbits) {
        if (nbits < 64 ? (v.val.v[0] >> nbits) != 0 || v.val.v[1] || v.val.v[2] || v.val.v[3] : (v.val.v[1] || v.val.v[2] || v.val.v[3])) fail("to_bits: the value does not fit");
        std::vector<Cell> bits;
        std::vector<Fe> coeff, prod;
        for (unsigned i = 0; i < nbits; i++) {
            bits.push_back(assign_bit((v.val.v[0] >> i) & 1));
            coeff.push_back(F.pow2(i));
            prod.push_back(bits.back().val.is_zero() ? F.zero() : coeff.back());
        }
        assert_equal(compose(bits.data(), coeff.data(), prod.data(), (int)nbits, F.zero()), v);
        return bits;
    }
    // RangeInstructions::assign: `limb_bits`-bit limbs four to a row with the composition lookup on a..d, the (bit_len mod limb_bits)-bit overflow limb
    // last, alone in column a of its row, with the overflow lookup.  value < 2^bit_len <= 2^128; the first row's e is the value, the later rows' what is left.
    Cell range_assign(u128 value, unsigned limb_bits, unsigned bit_len) {
        if (bit_len < 128 && (value >> bit_len)) fail("range_assign: the value does not fit");
        const unsigned full = bit_len / limb_bits, over = bit_len % limb_bits, nl = full + (over ? 1 : 0);
        if (over && full % 4) throw std::runtime_error("the overflow limb must open a row");
        if (!range_tag(limb_bits) || (over && !range_tag(over))) fail("range_assign: the range table has no rows for this bit length");
        const u128 mask = ((u128)1 << limb_bits) - 1;
        Cell result, o[5];
        for (unsigned g = 0; g < nl; g += 4) {
            const unsigned cnt = std::min(4u, nl - g);
            const bool last = g + 4 >= nl;
            Arg a[5];
            Sel s[10];
            int ns = 0;
            for (unsigned i = 0; i < cnt; i++) {
                a[i] = Arg(F.u((uint64_t)((value >> (limb_bits * (g + i))) & mask)));
                if (want_fixed) s[ns++] = {MG_SA + (int)i, F.pow2(limb_bits * (g + i))};
            }
            const unsigned sh = limb_bits * g;
            a[4] = Arg(F.from_u128(sh ? (value >> sh) << sh : value));
            if (want_fixed) {
                s[ns++] = {MG_SE, m1()};
                if (!last) s[ns++] = {MG_NEXT, one()};
                s[ns++] = {RC_S_COMPOSITION, one()};
                s[ns++] = {RC_TAG_COMPOSITION, F.u(range_tag(limb_bits))};
                if (last && over) { s[ns++] = {RC_S_OVERFLOW, one()}; s[ns++] = {RC_TAG_OVERFLOW, F.u(range_tag(over))}; }
            }
            row(a, 5, s, ns, o);
            if (g == 0) result = o[4];
        }
        return result;
    }
};

unsigned synth_threads();

// ---- BigIntChip (src/big_integer/chip.rs) ----
struct BigIntChip {
    Layouter& lay;
    size_t num_limbs;
    Big to_big(const std::vector<Cell>& limbs) const {
        Big r;
        for (auto& c : limbs) r.push_back(c.val.v[0]);
        return r;
    }
    Cell range_limb(uint64_t v) { return lay.range_assign(v, sublimb_bit_len(LIMB_WIDTH), LIMB_WIDTH); }
    std::vector<Cell> assign_integer(const Big& x, size_t n) {      // :64-85
        std::vector<Cell> out;
        for (uint64_t v : big_limbs(x, n)) out.push_back(range_limb(v));
        return out;
    }
    // :1255-1285: one row per limb the integer HAS, then one zero row whose cell pads the rest
    std::vector<Cell> assign_constant(Big x, size_t max_num_limbs) {
        big_trim(x);
        if (x.size() > max_num_limbs) throw std::runtime_error("assign_constant: too many limbs");
        std::vector<Cell> out;
        for (uint64_t v : x) out.push_back(lay.assign_constant(lay.F.u(v)));
        const Cell zero = lay.assign_constant(lay.F.zero());
        while (out.size() < max_num_limbs) out.push_back(zero);
        return out;
    }
    std::vector<Cell> max_value(size_t n) {
        std::vector<Cell> out;
        for (size_t i = 0; i < n; i++) out.push_back(lay.assign_constant(lay.F.u(~(uint64_t)0)));
        return out;
    }
    // :250-300: limb sums with range-checked (c, carry) pairs; max(n1, n2) + 1 limbs
    std::vector<Cell> add(std::vector<Cell> a, std::vector<Cell> b) {
        const size_t max_n = std::max(a.size(), b.size());
        const Cell zero = lay.assign_constant(lay.F.zero());
        a.resize(max_n, zero);
        b.resize(max_n, zero);
        Cell carry = zero;
        const Cell limb_max = lay.assign_constant(lay.F.pow2(LIMB_WIDTH));
        std::vector<Cell> out;
        for (size_t i = 0; i < max_n; i++) {
            const Cell s = lay.add(lay.add(a[i], b[i]), carry);
            const Cell c = range_limb(s.val.v[0]);
            if (s.val.v[2] | s.val.v[3]) lay.fail("add: a limb sum left 128 bits");
            carry = range_limb(s.val.v[1]);
            lay.assert_equal(s, lay.mul_add(Arg(carry), Arg(limb_max), Arg(c)));
            out.push_back(c);
        }
        out.push_back(carry);
        return out;
    }
    static Big big_sub(const Big& a, const Big& b, bool& negative) {      // a - b over max(len) limbs
        const size_t n = std::max(a.size(), b.size());
        Big r(n, 0);
        u128 borrow = 0;
        for (size_t i = 0; i < n; i++) {
            const u128 d = (u128)(i < a.size() ? a[i] : 0) - (i < b.size() ? b[i] : 0) - borrow;
            r[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
        negative = borrow != 0;
        return r;
    }
    // :1290-1322: c = a - b as fresh limbs, then a = b + c
    std::vector<Cell> sub_unchecked(const std::vector<Cell>& a, const std::vector<Cell>& b) {
        bool neg = false;
        const Big c_big = big_sub(to_big(a), to_big(b), neg);
        if (neg) lay.fail("sub_unchecked: a < b");
        std::vector<Cell> c;
        for (uint64_t v : big_limbs(c_big, a.size())) c.push_back(range_limb(v));
        assert_equal_fresh(a, add(b, c));
        return c;
    }
    // :313-376: (|a - b|, is_overflowed) through a + max - b
    std::vector<Cell> sub(const std::vector<Cell>& a, const std::vector<Cell>& b, Cell& is_overflowed) {
        const size_t n2 = b.size();
        const std::vector<Cell> max_int = max_value(n2);
        const std::vector<Cell> inflated_subed = sub_unchecked(add(a, max_int), b);
        const Cell one = lay.assign_bit(1);
        const Cell is_not_overflowed = lay.is_equal(inflated_subed[n2], one);
        is_overflowed = lay.not_(is_not_overflowed);
        const size_t num_l = inflated_subed.size(), num_r = std::max(a.size(), n2);
        const Cell zero = lay.assign_constant(lay.F.zero());
        std::vector<Cell> sel_l, sel_r;
        for (size_t i = 0; i < num_l; i++) sel_l.push_back(lay.select(inflated_subed[i], i >= n2 ? zero : b[i], is_not_overflowed));
        for (size_t i = 0; i < num_r; i++) {
            if (i >= a.size()) sel_r.push_back(lay.select(max_int[i], zero, is_not_overflowed));
            else if (i >= n2) sel_r.push_back(lay.select(zero, a[i], is_not_overflowed));
            else sel_r.push_back(lay.select(max_int[i], a[i], is_not_overflowed));
        }
        return sub_unchecked(sel_l, sel_r);
    }
    // :389-422: limb i of the product = sum_{j + k = i} a_j b_k by a chain of mul_add rows
    std::vector<Cell> mul(const std::vector<Cell>& a, const std::vector<Cell>& b) {
        const size_t d0 = a.size(), d1 = b.size();
        std::vector<Cell> out;
        // Proving (no fixed columns, no copies wanted) with one-word limbs -- every call of these circuits: the same rows written directly.  A product
        // limb is a sum of <= min(d0, d1) products of two 64-bit words: three words hold it, no field arithmetic is needed (half of all rows of the
        // delay-encryption circuit are these: 1087 per multiplication, 60 multiplications at a 15-bit exponent).
        bool words = !lay.want_fixed && std::min(d0, d1) <= ((size_t)1 << 60);
        for (size_t i = 0; words && i < d0; i++) words = !(a[i].val.v[1] | a[i].val.v[2] | a[i].val.v[3]);
        for (size_t i = 0; words && i < d1; i++) words = !(b[i].val.v[1] | b[i].val.v[2] | b[i].val.v[3]);
        if (words) {
            out.reserve(d0 + d1 - 1);
            const Fe zero{{0, 0, 0, 0}};
            for (size_t i = 0; i + 1 < d0 + d1; i++) {
                Cell acc = lay.assign_constant(zero);
                uint64_t w0 = 0, w1 = 0, w2 = 0;
                for (size_t j = d1 >= i + 1 ? 0 : i + 1 - d1; j < d0 && j <= i; j++) {
                    const uint64_t x = a[j].val.v[0], y = b[i - j].val.v[0];
                    const u128 p = (u128)x * y;
                    const u128 s0 = (u128)w0 + (uint64_t)p;
                    const u128 s1 = (u128)w1 + (uint64_t)(p >> 64) + (uint64_t)(s0 >> 64);
                    const Fe prev{{w0, w1, w2, 0}};
                    w0 = (uint64_t)s0; w1 = (uint64_t)s1; w2 += (uint64_t)(s1 >> 64);
                    const uint32_t r = lay.nrows++;
                    acc.col = 3; acc.row = r; acc.val = Fe{{w0, w1, w2, 0}};
                    if (r < lay.cap) {
                        lay.adv[0][r] = Fe{{x, 0, 0, 0}}; lay.adv[1][r] = Fe{{y, 0, 0, 0}}; lay.adv[2][r] = prev; lay.adv[3][r] = acc.val; lay.adv[4][r] = zero;
                    }
                }
                out.push_back(acc);
            }
            return out;
        }
        for (size_t i = 0; i + 1 < d0 + d1; i++) {
            Cell acc = lay.assign_constant(lay.F.zero());
            for (size_t j = d1 >= i + 1 ? 0 : i + 1 - d1; j < d0 && j <= i; j++) acc = lay.mul_add(Arg(a[j]), Arg(b[i - j]), Arg(acc));
            out.push_back(acc);
        }
        return out;
    }
    // :1327-1353 with n = 2^64 (its only use), five rows: q, a mod n assigned; n q; a - n q; equal to a mod n.  a is far below p.
    void div_mod_limb(const Cell& a, const Cell& limb_max, Cell& q, Cell& r) {
        const Fe qv{{a.val.v[1], a.val.v[2], a.val.v[3], 0}}, rv{{a.val.v[0], 0, 0, 0}};
        q = lay.assign_value(qv);
        r = lay.assign_value(rv);
        lay.assert_equal(r, lay.sub(a, lay.mul(Arg(limb_max), Arg(q))));
    }
    Cell is_equal_fresh(const std::vector<Cell>& a, const std::vector<Cell>& b) {      // :778-803
        const size_t n1 = a.size(), n2 = b.size();
        Cell eq = lay.assign_bit(1);
        for (size_t i = 0; i < std::max(n1, n2); i++) {
            Cell flag;
            if (n1 > n2 && i >= n2) flag = lay.is_zero(a[i]);
            else if (n1 <= n2 && i >= n1) flag = lay.is_zero(b[i]);
            else flag = lay.is_equal(a[i], b[i]);
            eq = lay.and_(eq, flag);
        }
        return eq;
    }
    void assert_equal_fresh(const std::vector<Cell>& a, const std::vector<Cell>& b) { lay.assert_one(is_equal_fresh(a, b)); }
    // :825-898 + assert_one (:1056-1066): a - b + word_max carried limb by limb, carries range-checked
    void assert_equal_muled(const std::vector<Cell>& a, const std::vector<Cell>& b, size_t n1, size_t n2) {
        Fld& F = lay.F;
        const size_t min_n = std::min(n1, n2);
        // word_max = min_n * limb_max^2 + limb_max (compute_mul_word_max): below 2^134 for 32 limbs
        const u128 limb_max_v = ~(uint64_t)0;
        Fe word_max;
        {
            const Fe lm2 = F.mul(F.from_u128(limb_max_v), F.from_u128(limb_max_v));
            word_max = F.add(F.mul(F.u(min_n), lm2), F.from_u128(limb_max_v));
        }
        unsigned wm_bits = 256;      // bit length of 2 * word_max
        {
            Fe two_wm = F.add(word_max, word_max);
            while (wm_bits && !((two_wm.v[(wm_bits - 1) >> 6] >> ((wm_bits - 1) & 63)) & 1)) wm_bits--;
        }
        const unsigned carry_bits = wm_bits - LIMB_WIDTH;
        const Cell limb_max = lay.assign_constant(F.pow2(LIMB_WIDTH));
        Cell accumulated_extra = lay.assign_constant(F.zero());
        Cell carry = lay.assign_constant(F.zero());
        Cell eq_bit = lay.assign_bit(1);
        const size_t num = n1 + n2 - 1;
        for (size_t i = 0; i < num; i++) {
            const Cell s = lay.add_with_constant(lay.sub(a[i], b[i]), carry, word_max);
            if (s.val.v[3]) lay.fail("is_equal_muled: the carried sum left the integers");
            Cell new_carry, c, q_acc, mod_acc;
            div_mod_limb(s, limb_max, new_carry, c);
            accumulated_extra = lay.add_constant(accumulated_extra, word_max);
            div_mod_limb(accumulated_extra, limb_max, q_acc, mod_acc);
            eq_bit = lay.and_(eq_bit, lay.is_equal(c, mod_acc));
            accumulated_extra = q_acc;
            if (i + 1 < num) {
                if (new_carry.val.v[2] | new_carry.val.v[3]) lay.fail("is_equal_muled: a carry left 128 bits");
                const u128 nc = ((u128)new_carry.val.v[1] << 64) | new_carry.val.v[0];
                const Cell ranged = lay.range_assign(nc, sublimb_bit_len(carry_bits), carry_bits);
                eq_bit = lay.and_(eq_bit, lay.is_equal(new_carry, ranged));
            } else {
                eq_bit = lay.and_(eq_bit, lay.is_equal(new_carry, accumulated_extra));
            }
            carry = new_carry;
        }
        lay.assert_one(eq_bit);
    }
    // :911-923, :1153-1161: (a <= b through sub's overflow flag) and not (a == b), asserted
    void assert_in_field(const std::vector<Cell>& a, const std::vector<Cell>& n) {
        Cell is_overflowed;
        sub(a, n, is_overflowed);
        lay.assert_one(lay.and_(is_overflowed, lay.not_(is_equal_fresh(a, n))));
    }
    std::vector<Cell> pow_mod_threads(std::vector<Cell> acc, std::vector<Cell> squared, const std::vector<Cell>& e_bits, const std::vector<Cell>& n, const Big& n_big);
    // :545-632
    std::vector<Cell> mul_mod(const std::vector<Cell>& a, const std::vector<Cell>& b, const std::vector<Cell>& n, const Big& n_big) {
        Big q_big, r_big;
        big_divmod(big_mul(to_big(a), to_big(b)), n_big, q_big, r_big);
        return mul_mod_rows(a, b, n, q_big, r_big);
    }
    // the rows of a * b = q * n + r for a quotient and remainder already known
    std::vector<Cell> mul_mod_rows(const std::vector<Cell>& a, const std::vector<Cell>& b, const std::vector<Cell>& n, const Big& q_big, const Big& r_big) {
        const size_t n1 = a.size(), n2 = b.size();
        std::vector<Cell> q, r;
        for (uint64_t v : big_limbs(q_big, n2)) q.push_back(range_limb(v));
        for (uint64_t v : big_limbs(r_big, n1)) r.push_back(range_limb(v));
        const std::vector<Cell> ab = mul(a, b), qn = mul(q, n);
        std::vector<Cell> eq_b;
        for (size_t i = 0; i + 1 < n1 + n2; i++) eq_b.push_back(i < n1 ? lay.add(qn[i], r[i]) : qn[i]);
        assert_equal_muled(ab, eq_b, n1, n2);
        return r;
    }
    // :667-699: the exponent's limbs into bits, then per bit (LSB first) acc * squared, select, squared^2
    std::vector<Cell> pow_mod(const std::vector<Cell>& a, const std::vector<Cell>& e, const std::vector<Cell>& n, const Big& n_big, unsigned exp_limb_bits) {
        std::vector<Cell> e_bits;
        for (auto& limb : e)
            for (auto& b : lay.to_bits(limb, exp_limb_bits)) e_bits.push_back(b);
        std::vector<Cell> acc = assign_constant(Big{1}, num_limbs);      // assign_constant_fresh(1)
        std::vector<Cell> squared = a;
        if (!lay.want_fixed && e_bits.size() >= 3 && synth_threads() > 1) return pow_mod_threads(acc, squared, e_bits, n, n_big);
        for (auto& bit : e_bits) {
            const std::vector<Cell> muled = mul_mod(acc, squared, n, n_big);
            for (size_t j = 0; j < acc.size(); j++) acc[j] = lay.select(muled[j], acc[j], bit);
            squared = mul_mod(squared, squared, n, n_big);
        }
        return acc;
    }
};

// Proving (values only): the two multiplications of every exponent bit are regions of a fixed number of rows whose operands follow from the
// integers alone, so the chain x^(2^i), acc_i is computed first (two 2048-bit products and divisions per bit) and the regions are then written
// by several host threads, each through a cursor of its own at the row where the sequential order puts it -- the same rows, bit for bit
// (tests/test_witness.py compares with the one-thread keygen path).  The first bit runs on the caller's cursor and gives the regions' lengths.
unsigned synth_threads() {
    static const unsigned n = [] {
        const char* e = getenv("DEHALO_SYNTH_THREADS");
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        return e ? (unsigned)std::max(1, atoi(e)) : std::min(8u, hw);
    }();
    return n;
}

// Worker threads kept for the life of the process (starting seven threads costs more than the regions they would write: 25 us each against 34 us per
// region): they sleep on a condition variable between calls.  One job at a time -- a second caller that finds the pool busy writes its regions itself.
struct SynthPool {
    std::mutex job_mu, mu;
    std::condition_variable cv, done_cv;
    const std::function<void()>* fn = nullptr;
    uint64_t gen = 0;
    unsigned pending = 0, workers = 0;
    explicit SynthPool(unsigned n) : workers(n) {
        for (unsigned i = 0; i < n; i++)
            std::thread([this] {
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return gen != seen; });
                    seen = gen;
                    const std::function<void()>* f = fn;
                    lk.unlock();
                    (*f)();
                    lk.lock();
                    if (--pending == 0) done_cv.notify_one();
                }
            }).detach();
    }
    void run(const std::function<void()>& f) {      // f on every worker and on the caller; returns when all are back
        { std::lock_guard<std::mutex> lk(mu); fn = &f; pending = workers; gen++; }
        cv.notify_all();
        f();
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
};
// (never destroyed: its threads sleep until the process ends.  A forked child has none of them: it starts with no pool and makes its own.)
std::atomic<SynthPool*> g_synth_pool{nullptr};
SynthPool* synth_pool() {
    static const int registered = pthread_atfork(nullptr, nullptr, [] { g_synth_pool.store(nullptr); });
    (void)registered;
    SynthPool* p = g_synth_pool.load();
    if (!p) {
        SynthPool* fresh = new SynthPool(synth_threads() - 1);
        if (g_synth_pool.compare_exchange_strong(p, fresh)) p = fresh;      // (lost the race: `fresh` stays behind, asleep)
    }
    return p;
}

std::vector<Cell> BigIntChip::pow_mod_threads(std::vector<Cell> acc, std::vector<Cell> squared, const std::vector<Cell>& e_bits, const std::vector<Cell>& n, const Big& n_big) {
    const size_t nb = e_bits.size();
    auto cells_of = [&](const Big& x) {
        std::vector<Cell> out(num_limbs);
        const Big l = big_limbs(x, num_limbs);
        for (size_t i = 0; i < num_limbs; i++) out[i].val = Fe{{l[i], 0, 0, 0}};      // values only: nothing of the proving path reads where a cell sits
        return out;
    };
    auto T0 = std::chrono::steady_clock::now();
    struct Step { Big acc, sq, q_mul, r_mul, q_sq, r_sq; };
    std::vector<Step> st(nb);
    Big a_cur = to_big(acc), s_cur = to_big(squared);
    for (size_t i = 0; i < nb; i++) {
        st[i].acc = a_cur; st[i].sq = s_cur;
        big_divmod(big_mul(a_cur, s_cur), n_big, st[i].q_mul, st[i].r_mul);
        big_divmod(big_mul(s_cur, s_cur), n_big, st[i].q_sq, st[i].r_sq);
        if (!e_bits[i].val.is_zero()) a_cur = st[i].r_mul;
        s_cur = st[i].r_sq;
    }
    auto T1 = std::chrono::steady_clock::now();
    // bit 0 on the caller's cursor: the lengths of the two regions
    const uint32_t r0 = lay.nrows;
    {
        const std::vector<Cell> muled = mul_mod_rows(acc, squared, n, st[0].q_mul, st[0].r_mul);
        for (size_t j = 0; j < acc.size(); j++) acc[j] = lay.select(muled[j], acc[j], e_bits[0]);
    }
    const uint32_t len_a = lay.nrows - r0;
    squared = mul_mod_rows(squared, squared, n, st[0].q_sq, st[0].r_sq);
    const uint32_t len_b = lay.nrows - r0 - len_a;
    const uint32_t base = lay.nrows;
    auto T2 = std::chrono::steady_clock::now();
    // regions 2 (i - 1) and 2 (i - 1) + 1 of bit i >= 1
    const bool with_tail = (bool)lay.after_pow;
    const size_t tasks = 2 * (nb - 1) + (with_tail ? 1 : 0);
    const uint32_t end_row = base + (uint32_t)(nb - 1) * (len_a + len_b);
    uint32_t tail_end = end_row;
    const std::vector<Cell> powed = cells_of(a_cur);
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true}, unsat{false};
    auto work = [&]() {
        for (;;) try {
            size_t t = next.fetch_add(1);
            if (t >= tasks) return;
            if (with_tail) {      // the longest task first
                if (t == 0) {
                    Layouter sub(lay, end_row);
                    lay.after_pow(sub, powed);
                    tail_end = sub.nrows;
                    continue;
                }
                t--;
            }
            const size_t i = 1 + t / 2;
            const uint32_t start = base + (uint32_t)(i - 1) * (len_a + len_b) + (t & 1 ? len_a : 0);
            Layouter sub(lay, start);
            BigIntChip chip{sub, num_limbs};
            if (t & 1) {
                const std::vector<Cell> sq = cells_of(st[i].sq);
                chip.mul_mod_rows(sq, sq, n, st[i].q_sq, st[i].r_sq);
                if (sub.nrows != start + len_b) ok = false;
            } else {
                const std::vector<Cell> ac = cells_of(st[i].acc), sq = cells_of(st[i].sq);
                const std::vector<Cell> muled = chip.mul_mod_rows(ac, sq, n, st[i].q_mul, st[i].r_mul);
                for (size_t j = 0; j < ac.size(); j++) sub.select(muled[j], ac[j], e_bits[i]);
                if (sub.nrows != start + len_a) ok = false;
            }
        } catch (const NotSatisfied&) { unsat = true; ok = false; return;
        } catch (...) { ok = false; return; }
    };
    const unsigned nthreads = synth_threads();
    {
        SynthPool* pool = synth_pool();
        const std::function<void()> job = work;
        if (pool->job_mu.try_lock()) {
            pool->run(job);
            pool->job_mu.unlock();
        } else work();
    }
    auto T3 = std::chrono::steady_clock::now();
    if (getenv("DEHALO_SYNTH_TRACE")) fprintf(stderr, "pow_mod: chain %.3f ms, bit 0 %.3f ms, %zu regions on %u threads %.3f ms\n", std::chrono::duration<double, std::milli>(T1 - T0).count(), std::chrono::duration<double, std::milli>(T2 - T1).count(), tasks, nthreads, std::chrono::duration<double, std::milli>(T3 - T2).count());
    if (unsat) throw NotSatisfied("a constraint behind the first exponent bit fails");
    if (!ok) throw std::runtime_error("pow_mod: a region's length depends on its values");
    lay.nrows = with_tail ? tail_end : end_row;
    lay.after_pow_done = with_tail;
    return powed;
}

// ---- PoseidonChip (src/poseidon/chip.rs): the optimised permutation as MainGate rows -- 5 (absorb) + 8 x (15 + 10) + 57 x (3 + 6) = 718 rows for T = 5 ----
struct PoseidonChip {
    Layouter& lay;
    const PoseidonSpec& sp;
    std::vector<Cell> state;
    void sbox(Cell& w, const Fe& constant) {      // :199-222: w^2, w^4, w^4 w + constant
        Cell t = lay.mul(Arg(w), Arg(w));
        t = lay.mul(Arg(t), Arg(t));
        w = lay.mul_add_constant(t, w, constant);
    }
    void sbox_full(const std::vector<Fe>& constants) { for (uint32_t i = 0; i < sp.t; i++) sbox(state[i], constants[i]); }
    // :225-262: word 0 + c0; the inputs added to the words behind it with their constants; the rest + constant (+ 1 on the first of them when hashing)
    void absorb_with_pre_constants(const std::vector<Cell>& inputs, const std::vector<Fe>& pre, bool h_flag) {
        const size_t offset = inputs.size() + 1;
        state[0] = lay.add_constant(state[0], pre[0]);
        for (size_t i = 0; i < inputs.size(); i++) state[1 + i] = lay.add_with_constant(state[1 + i], inputs[i], pre[1 + i]);
        for (size_t i = offset; i < sp.t; i++) state[i] = lay.add_constant(state[i], h_flag && i == offset ? lay.F.add(pre[i], lay.one()) : pre[i]);
    }
    Cell linear(const Cell* cells, const Fe* coeff, const Fe* coeff_m, int n) {
        Fe prod[8];
        for (int i = 0; i < n; i++) prod[i] = lay.F.f->mul(coeff_m[i], cells[i].val);
        return lay.compose(cells, coeff, prod, n, lay.F.zero());
    }
    void apply_mds(const Mat& m, const Mat& m_mont) {      // :264-288: every output word a compose over the five state words (two rows)
        std::vector<Cell> nx;
        for (uint32_t i = 0; i < sp.t; i++) nx.push_back(linear(state.data(), m[i].data(), m_mont[i].data(), (int)sp.t));
        state = nx;
    }
    void apply_sparse_mds(uint32_t r) {      // :291-330: word 0 a compose over the state, word i a compose of (word 0, col_hat) and (word i, 1)
        std::vector<Cell> nx = {linear(state.data(), sp.sparse_row[r].data(), sp.sparse_row_m[r].data(), (int)sp.t)};
        const Fe one_m = lay.F.f->one;
        for (uint32_t i = 1; i < sp.t; i++) {
            const Cell c[2] = {state[0], state[i]};
            const Fe k[2] = {sp.sparse_col[r][i - 1], lay.one()}, km[2] = {sp.sparse_col_m[r][i - 1], one_m};
            nx.push_back(linear(c, k, km, 2));
        }
        state = nx;
    }
    void permutation(const std::vector<Cell>& inputs, bool h_flag) {      // :333-419
        const uint32_t half = sp.r_f / 2;
        absorb_with_pre_constants(inputs, sp.start[0], h_flag);
        for (uint32_t r = 1; r < half; r++) {
            sbox_full(sp.start[r]);
            apply_mds(sp.mds, sp.mds_m);
        }
        sbox_full(sp.start[half]);
        apply_mds(sp.pre_sparse, sp.pre_sparse_m);
        for (uint32_t r = 0; r < sp.r_p; r++) {
            sbox(state[0], sp.partial[r]);
            apply_sparse_mds(r);
        }
        for (auto& c : sp.end) {
            sbox_full(c);
            apply_mds(sp.mds, sp.mds_m);
        }
        sbox_full(std::vector<Fe>(sp.t, lay.F.zero()));
        apply_mds(sp.mds, sp.mds_m);
    }
};

// PoseidonCipher::{initial_state, encrypt} (src/encryption/poseidon_enc.rs:66-133), T = 5, RATE = 4, MESSAGE_CAPACITY = message.size(): the message is added
// to a COPY of the state (`state.words()`, :108-119); a full chunk is then absorbed and permuted (update), a short one permutes the state as it is (squeeze(0)
// with an empty absorbing line)
std::vector<Fe> native_encrypt(const PoseidonSpec& sp, const Fe key[2], const std::vector<Fe>& message, uint64_t nonce, uint32_t rate) {
    const Fld& F = sp.F;
    std::vector<Fe> st = sp.permute({F.zero(), F.zero(), key[0], key[1], F.u(nonce)});
    std::vector<Fe> cipher;
    for (size_t c0 = 0; c0 < message.size(); c0 += rate) {
        const size_t cnt = std::min<size_t>(rate, message.size() - c0);
        for (size_t j = 0; j < cnt; j++) cipher.push_back(F.add(st[1 + j], message[c0 + j]));
        if (cnt == rate) for (size_t j = 0; j < cnt; j++) st[1 + j] = F.add(st[1 + j], message[c0 + j]);
        st = sp.permute(st);
    }
    cipher.push_back(st[1]);
    return cipher;
}

// the expected value x^e mod n as constants, asserted equal to the exponentiation's result (src/lib.rs:205-215)
std::vector<Cell> rsa_expected_rows(Layouter& lay, const std::vector<Cell>& powed) {
    BigIntChip chip{lay, powed.size()};
    const std::vector<Cell> valid = chip.assign_constant(chip.to_big(powed), powed.size());
    chip.assert_equal_fresh(powed, valid);
    return valid;
}

// src/lib.rs:179-215 / benches/mod_pow.rs:91-116: assign (n, e), x; x < n; x^e mod n in-circuit; equal to the native big_pow_mod
std::vector<Cell> rsa_region(Layouter& lay, const Big& n_big, uint64_t e, const Big& x, unsigned exp_bits, size_t num_limbs, Big& want) {
    BigIntChip chip{lay, num_limbs};
    const std::vector<Cell> n_limbs = chip.assign_integer(n_big, num_limbs);      // assign_public_key: n, then the one-limb exponent
    const std::vector<Cell> e_limbs = chip.assign_integer(Big{e}, 1);
    const std::vector<Cell> x_limbs = chip.assign_integer(x, num_limbs);
    chip.assert_in_field(x_limbs, n_limbs);                                        // modpow_public_key, src/rsa/chip.rs:109
    const std::vector<Cell> powed = chip.pow_mod(x_limbs, e_limbs, n_limbs, n_big, exp_bits);
    // the native big_pow_mod(x, e, n) the reference assigns as the expected value: the rows above hold exactly its square-and-multiply chain (every
    // mul_mod's remainder came from the same big-integer division), so its value is read off them instead of being computed a second time
    want = chip.to_big(powed);
    big_trim(want);
    if (lay.after_pow_done) return powed;      // (the rows behind the exponentiation were written beside it)
    return rsa_expected_rows(lay, powed);
}

// src/lib.rs:261-316 / src/encryption/chip.rs:150-200: the Poseidon cipher in-circuit, constrained equal to the native one.  pose_enc (no key cells): the
// initial state is five constants (new_enc); delay_enc: five witnesses (new_enc_de), words 2 and 3 equal to the digest.
std::vector<Cell> cipher_region(Layouter& lay, const PoseidonSpec& spec, const Fe key_vals[2], const std::vector<Fe>& message, const Cell* key_cells, uint32_t rate) {
    std::vector<Cell> expected;
    for (auto& v : native_encrypt(spec, key_vals, message, 1, rate)) expected.push_back(lay.assign_value(v));
    const Fe init[5] = {lay.F.zero(), lay.F.zero(), key_vals[0], key_vals[1], lay.F.u(1)};
    PoseidonChip chip{lay, spec, {}};
    for (auto& v : init) chip.state.push_back(key_cells ? lay.assign_value(v) : lay.assign_constant(v));
    if (key_cells) {
        lay.assert_equal(chip.state[2], key_cells[0]);
        lay.assert_equal(chip.state[3], key_cells[1]);
    }
    chip.permutation({}, false);
    std::vector<Cell> msg, cipher;
    for (auto& m : message) msg.push_back(lay.assign_value(m));
    for (size_t c0 = 0; c0 < msg.size(); c0 += rate) {      // absorb_and_relese, src/encryption/chip.rs:72-110
        const std::vector<Cell> chunk(msg.begin() + (long)c0, msg.begin() + (long)std::min<size_t>(msg.size(), c0 + rate));
        for (size_t j = 0; j < chunk.size(); j++) {
            chip.state[1 + j] = lay.add(chip.state[1 + j], chunk[j]);
            cipher.push_back(chip.state[1 + j]);
        }
        chip.permutation(chunk, false);
    }
    cipher.push_back(chip.state[1]);
    for (size_t i = 0; i < cipher.size(); i++) lay.assert_equal(cipher[i], expected[i]);
    return cipher;
}

// src/lib.rs:222-259: limbs packed three to a field element, HasherChip::hash (src/hash/chip.rs:63-85) with perm_hash's padding; returns the two key words
void hash_region(Layouter& L, const PoseidonSpec& spec, const std::vector<Cell>& rsa_out, uint32_t rate, Cell key_cells[2]) {
    Fld& F = L.F;
    PoseidonChip chip{L, spec, {}};
    chip.state.push_back(L.assign_constant(F.pow2(64)));      // State::default: capacity word 2^64
    for (uint32_t i = 1; i < spec.t; i++) chip.state.push_back(L.assign_constant(F.zero()));
    const Cell base1 = L.assign_constant(F.pow2(LIMB_WIDTH));
    const Cell base2 = L.mul(Arg(base1), Arg(base1));
    std::vector<Cell> inputs;
    for (size_t i = 0; i < rsa_out.size() / 3; i++) {
        const Cell a = L.mul_add(Arg(rsa_out[3 * i + 1]), Arg(base1), Arg(rsa_out[3 * i]));
        inputs.push_back(L.mul_add(Arg(rsa_out[3 * i + 2]), Arg(base2), Arg(a)));
    }
    if (rsa_out.size() % 3 == 2) inputs.push_back(L.mul_add(Arg(rsa_out[rsa_out.size() - 1]), Arg(base1), Arg(rsa_out[rsa_out.size() - 2])));      // limbs 30, 31 of the reference's 32
    else if (rsa_out.size() % 3 == 1) inputs.push_back(rsa_out.back());
    size_t padding_offset = 0;
    for (size_t c0 = 0; c0 < inputs.size(); c0 += rate) {
        const std::vector<Cell> chunk(inputs.begin() + (long)c0, inputs.begin() + (long)std::min<size_t>(inputs.size(), c0 + rate));
        padding_offset = rate - chunk.size();
        chip.permutation(chunk, true);
    }
    if (padding_offset == 0) chip.permutation({}, true);
    key_cells[0] = chip.state[1];
    key_cells[1] = chip.state[2];
}

// permutation::keygen::Assembly::copy [UPSTREAM plonk/permutation/keygen.rs]: cycles over cells, the smaller cycle relabelled
struct Assembly {
    size_t n;
    std::vector<uint64_t> mapping, aux, sizes;
    Assembly(size_t cols, size_t n_) : n(n_), mapping(cols * n_), aux(cols * n_), sizes(cols * n_, 1) {
        for (size_t i = 0; i < mapping.size(); i++) mapping[i] = aux[i] = i;
    }
    void copy(const Copy& c) {
        uint64_t left = c.c0 * n + c.r0, right = c.c1 * n + c.r1;
        if (aux[left] == aux[right]) return;
        if (sizes[aux[left]] < sizes[aux[right]]) std::swap(left, right);
        const uint64_t la = aux[left];
        sizes[la] += sizes[aux[right]];
        uint64_t i = right;
        do {
            aux[i] = la;
            i = mapping[i];
        } while (i != right);
        std::swap(mapping[left], mapping[right]);
    }
};

Fe fe_from(const uint64_t* p) {
    Fe r;
    memcpy(r.v, p, 32);
    return r;
}

}   // namespace

extern "C" int dehalo_synthesize(const dehalo_circuit_inputs* in, uint64_t* advice, uint64_t* fixed, uint64_t* mapping, uint8_t* const* selectors,
                                 dehalo_synthesis_info* info) try {
    if (!in || (!advice && !fixed && !mapping && !info)) return DEHALO_ERR_INVALID;
    const HostField* f = host_field(DEHALO_FIELD_BN254_FR);
    const bool keygen_outputs = fixed || mapping || selectors;
    const bool range_lookups = in->circuit != DEHALO_CIRCUIT_POSE_ENC;
    if (in->circuit > DEHALO_CIRCUIT_POSE_ENC || in->k < 4 || in->k > 24) return DEHALO_ERR_INVALID;
    const uint32_t t = in->t ? in->t : 5, rate = in->rate ? in->rate : 4, r_f = in->r_f ? in->r_f : 8, r_p = in->r_p ? in->r_p : 57;
    if (t != 5 || rate != 4) return DEHALO_ERR_UNSUPPORTED;      // the row layout (linear()) is the T = 5 one the reference instantiates (src/lib.rs:120-121)
    if (in->message_len > 2 || (in->message_len && !in->message)) return DEHALO_ERR_INVALID;
    const size_t n = (size_t)1 << in->k;
    const auto t_begin = std::chrono::steady_clock::now();
    auto trace = [&](const char* what) {
        if (getenv("DEHALO_SYNTH_TRACE")) fprintf(stderr, "synthesize: %-28s +%.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    Layouter lay(f, keygen_outputs, advice, n);
    std::vector<Fe> message;
    for (uint32_t i = 0; i < in->message_len; i++) {
        Fe m = fe_from(in->message + 4 * i);
        if (HostField::geq(m.v, f->p)) return DEHALO_ERR_INVALID;
        message.push_back(m);
    }
    dehalo_synthesis_info inf{};
    std::vector<Fe> cipher_vals;
    if (in->circuit == DEHALO_CIRCUIT_POSE_ENC) {
        if (!in->key) return DEHALO_ERR_INVALID;
        const std::shared_ptr<const PoseidonSpec> spec_ref = poseidon_spec(f, t, r_f, r_p);
        const PoseidonSpec& spec = *spec_ref;
        const Fe key[2] = {fe_from(in->key), fe_from(in->key + 4)};
        for (auto& c : cipher_region(lay, spec, key, message, nullptr, rate)) cipher_vals.push_back(c.val);
    } else {
        if (!in->n || !in->x || in->bits_len % LIMB_WIDTH || in->bits_len == 0 || in->bits_len > 8192 || in->exp_bits == 0 || in->exp_bits > 64) return DEHALO_ERR_INVALID;
        const size_t num_limbs = in->bits_len / LIMB_WIDTH;
        lay.range = RangeLens(num_limbs);
        Big n_big(in->n, in->n + num_limbs), x(in->x, in->x + num_limbs);
        { Big t0 = n_big; big_trim(t0); if (t0.empty()) return DEHALO_ERR_INVALID; }
        if (in->exp_bits < 64 && (in->e >> in->exp_bits)) return DEHALO_ERR_INVALID;
        // hash region, then the cipher keyed by the digest's words 1 and 2 (src/lib.rs:216-316)
        auto hash_and_cipher = [&](Layouter& L, const std::vector<Cell>& rsa_out) {
            const std::shared_ptr<const PoseidonSpec> spec_ref = poseidon_spec(f, t, r_f, r_p);
            const PoseidonSpec& spec = *spec_ref;
            Cell key_cells[2];
            hash_region(L, spec, rsa_out, rate, key_cells);
            const Fe key_vals[2] = {key_cells[0].val, key_cells[1].val};
            for (auto& c : cipher_region(L, spec, key_vals, message, key_cells, rate)) cipher_vals.push_back(c.val);
        };
        const bool with_hash = in->circuit == DEHALO_CIRCUIT_DELAY_ENC;
        uint32_t rsa_rows_beside = 0;
        if (!keygen_outputs)      // proving: everything behind the exponentiation may be written beside it (BigIntChip::pow_mod_threads)
            lay.after_pow = [&](Layouter& sub, const std::vector<Cell>& powed) {
                const std::vector<Cell> valid = rsa_expected_rows(sub, powed);
                rsa_rows_beside = sub.rows();
                if (with_hash) hash_and_cipher(sub, valid);
            };
        Big want;
        const std::vector<Cell> rsa_out = rsa_region(lay, n_big, in->e, x, in->exp_bits, num_limbs, want);
        inf.rsa_rows = lay.after_pow_done ? rsa_rows_beside : lay.rows();
        trace(lay.after_pow_done ? "rsa, hash and cipher regions" : "rsa region");
        const Big wl = big_limbs(want, std::min<size_t>(num_limbs, 128));
        for (size_t i = 0; i < wl.size() && i < 128; i++) inf.rsa_result[i] = wl[i];
        if (with_hash && !lay.after_pow_done) hash_and_cipher(lay, rsa_out);
    }
    trace("hash and cipher regions");
    inf.total_rows = lay.rows();
    if (in->circuit == DEHALO_CIRCUIT_POSE_ENC) inf.rsa_rows = 0;
    for (size_t i = 0; i < cipher_vals.size() && i < 3; i++) memcpy(inf.cipher + 4 * i, cipher_vals[i].v, 32);
    inf.cipher_len = (uint32_t)std::min<size_t>(cipher_vals.size(), 3);
    if (info) *info = inf;
    // ---- into 2^k-row columns
    const uint32_t bf = 5;      // blinding_factors of both constraint systems: max(3, 2 queries of advice column e) + 2
    const size_t u = n - (bf + 1);
    if ((size_t)lay.rows() + 1 > u) return DEHALO_ERR_INVALID;      // "not enough rows available" (upstream: Error::NotEnoughRowsAvailable)
    const size_t rows = lay.rows();
    if (advice)      // (the used rows are already in place)
        for (int c = 0; c < 5; c++) memset(advice + ((size_t)c * n + rows) * 4, 0, (n - rows) * 32);
    trace("unused rows zeroed");
    const uint32_t num_fixed = range_lookups ? 15 : 9;
    if (fixed) {
        memset(fixed, 0, (size_t)num_fixed * n * 32);
        for (uint32_t c = 0; c < num_fixed; c++) memcpy(fixed + (size_t)c * n * 4, lay.fix[c].data(), rows * 32);
        if (range_lookups) {      // RangeChip::load_table rows (tag, value): the disabled row (0, 0), then every value of every bit length
            size_t r = 1;
            for (unsigned ti = 0; ti < lay.range.count; ti++)
                for (uint64_t v = 0; v < ((uint64_t)1 << lay.range.bits[ti]); v++, r++) {
                    if (r >= u) return DEHALO_ERR_INVALID;
                    fixed[((size_t)RC_T_TAG * n + r) * 4] = ti + 1;
                    fixed[((size_t)RC_T_VALUE * n + r) * 4] = v;
                }
        }
    }
    if (!range_lookups && keygen_outputs)
        for (int c = 9; c < NUM_FIX; c++)
            for (auto& v : lay.fix[c])
                if (!v.is_zero()) return DEHALO_ERR_INVALID;      // range rows in a MainGate-only circuit
    if (mapping) {
        Assembly asm_(6, n);
        for (auto& c : lay.copies) asm_.copy(c);
        memcpy(mapping, asm_.mapping.data(), 6 * n * 8);
    }
    if (selectors && range_lookups) {
        for (int s = 0; s < 2; s++) {
            if (!selectors[s]) continue;
            memset(selectors[s], 0, n);
            const auto& col = lay.fix[s == 0 ? RC_S_COMPOSITION : RC_S_OVERFLOW];
            for (size_t r = 0; r < rows; r++) selectors[s][r] = col[r].is_zero() ? 0 : 1;
        }
    }
    return 0;
} catch (const NotSatisfied&) {
    return DEHALO_ERR_INVALID;      // a constraint of the reference's circuit does not hold for these inputs (x >= n, an exponent wider than exp_bits, a non-zero message)
} catch (...) { return DEHALO_ERR_OOM; }
