// witness.hip -- Circuit::synthesize of the reference's three circuits as VALUES, in C++: what fills the advice columns (and, at keygen,
// the fixed columns and the permutation) before create_proof commits them -- the part of the reference's timed call that precedes the
// first commitment [src/lib.rs:164-318 DelayEncryptCircuit::synthesize; benches/mod_pow.rs:63-110 RSACircuit; src/encryption/chip.rs:
// 114-204 PoseidonEncCircuit].  Restated from:
//   big_pow_mod (its value: read off the pow_mod rows)   src/big_integer/utils.rs:2-17
//   BigIntChip::{mul, mul_mod, pow_mod, is_equal_muled}   src/big_integer/chip.rs:389-422, 545-632, 667-699, 825-898
//   Grain LFSR, Cauchy MDS, the permutation       src/poseidon/grain.rs:12-157, spec.rs:170-180, permutation.rs:60-80
//   sponge (RATE 4) and cipher                    src/hash/chip.rs:63-85, src/encryption/poseidon_enc.rs:66-133, src/lib.rs:222-316
// The cell layout is this repository's own small layouter over the reference's gate (MainGate + RangeChip; halo2wrong's region code is
// upstream and not in the reference tree): the same rows, in the same order, as dehalo2_amd/witness.py, which the tests compare it with
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

struct PoseidonSpec {
    Fld F;
    uint32_t t, r_f, r_p;
    std::vector<std::vector<Fe>> constants, mds, mds_m;      // mds_m: the same matrix times R (Montgomery form): ONE Montgomery product gives m * x for a canonical x
    Fe mds_mul(uint32_t i, uint32_t j, const Fe& x) const { return F.f->mul(mds_m[i][j], x); }
    PoseidonSpec(const HostField* f, uint32_t t_, uint32_t rf, uint32_t rp) : F{f}, t(t_), r_f(rf), r_p(rp) {
        Grain g(f, t, r_f, r_p);
        constants.assign(r_f + r_p, std::vector<Fe>(t));
        for (auto& row : constants) for (auto& c : row) c = g.field_element();
        std::vector<Fe> xs(t), ys(t);
        for (auto& x : xs) x = g.field_element_mod();
        for (auto& y : ys) y = g.field_element_mod();
        mds.assign(t, std::vector<Fe>(t));
        for (uint32_t i = 0; i < t; i++) for (uint32_t j = 0; j < t; j++) mds[i][j] = F.inv(F.add(xs[i], ys[j]));
        mds_m = mds;
        for (auto& row : mds_m) for (auto& m : row) m = f->from_canonical(m);
    }
    Fe pow5(const Fe& x) const {
        const Fe x2 = F.mul(x, x);
        return F.mul(F.mul(x2, x2), x);
    }
    std::vector<Fe> permute(std::vector<Fe> st) const {      // src/poseidon/permutation.rs:60-80
        const uint32_t half = r_f / 2;
        for (uint32_t r = 0; r < constants.size(); r++) {
            for (uint32_t i = 0; i < t; i++) st[i] = F.add(st[i], constants[r][i]);
            if (r < half || r >= half + r_p) for (auto& e : st) e = pow5(e);
            else st[0] = pow5(st[0]);
            std::vector<Fe> nx(t, F.zero());
            for (uint32_t i = 0; i < t; i++) for (uint32_t j = 0; j < t; j++) nx[i] = F.add(nx[i], mds_mul(i, j, st[j]));
            st = nx;
        }
        return st;
    }
};

// The round constants and the MDS matrix depend on (T, R_F, R_P) only -- 325 Grain draws and 25 field inversions at the reference's parameters, 1.5 ms, most
// of a PoseidonEnc witness -- so a process keeps each parameter set it has used (the reference rebuilds the Spec in every synthesize: src/poseidon/spec.rs).
std::shared_ptr<const PoseidonSpec> poseidon_spec(const HostField* f, uint32_t t, uint32_t r_f, uint32_t r_p) {
    static std::mutex mu;
    static std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::shared_ptr<const PoseidonSpec>> cache;
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= 16) cache.clear();      // (callers hold their own reference)
    auto& slot = cache[std::make_tuple(t, r_f, r_p)];
    if (!slot) slot = std::make_shared<const PoseidonSpec>(f, t, r_f, r_p);
    return slot;
}

// ---- the MainGate / RangeChip layouter (dehalo2_amd/witness.py Layouter, row for row) ----
enum { MG_SA = 0, MG_SB, MG_SC, MG_SD, MG_SE, MG_MUL_AB, MG_MUL_CD, MG_NEXT, MG_CONST, RC_T_TAG, RC_T_VALUE, RC_TAG_COMPOSITION, RC_TAG_OVERFLOW, RC_S_COMPOSITION, RC_S_OVERFLOW, NUM_FIX };
// RangeChip::configure(composition_bit_lens = (8, 4, 1), overflow_bit_lens = (6,)): tags 1..4 (src/lib.rs:144-149)
constexpr unsigned COMPOSITION_BITS[3] = {8, 4, 1};
constexpr unsigned OVERFLOW_BITS = 6;
inline uint64_t range_tag(unsigned bits) { return bits == 8 ? 1 : bits == 4 ? 2 : bits == 1 ? 3 : bits == 6 ? 4 : 0; }

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
    Layouter(const HostField* f, bool fixed_too, uint64_t* advice, size_t n) : F{f}, want_fixed(fixed_too), cap(n) {
        if (!advice) own.resize(5 * n);
        Fe* base = advice ? reinterpret_cast<Fe*>(advice) : own.data();
        for (int i = 0; i < 5; i++) adv[i] = base + (size_t)i * n;
    }
    // a second cursor over the same columns, starting at row `start` (proving only: no fixed columns, no copies): the rows of independent regions whose
    // positions are known in advance are written by several host threads at once (BigIntChip::pow_mod)
    Layouter(const Layouter& parent, uint32_t start) : F(parent.F), want_fixed(false), cap(parent.cap), nrows(start) {
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
    Fe one() const { return F.u(1); }
    Fe m1() const { return F.neg(F.u(1)); }

    Cell assign_value(const Fe& v) { Cell o[5]; Arg a[1] = {Arg(v)}; row(a, 1, nullptr, 0, o); return o[0]; }
    Cell assign_constant(const Fe& v) { Cell o[5]; Arg a[1] = {Arg(v)}; Sel s[2] = {{MG_SA, one()}, {MG_CONST, F.neg(v)}}; row(a, 1, s, 2, o); return o[0]; }
    Cell mul_add(const Arg& a, const Arg& b, const Arg& c) {
        Cell o[5];
        Arg x[4] = {a, b, c, Arg(F.add(F.mul(a.c.val, b.c.val), c.c.val))};
        Sel s[3] = {{MG_MUL_AB, one()}, {MG_SC, one()}, {MG_SD, m1()}};
        row(x, 4, s, 3, o);
        return o[3];
    }
    Cell mul(const Arg& a, const Arg& b) { return mul_add(a, b, Arg()); }
    Cell add(const Cell& a, const Cell& b, const Fe& constant) {
        Cell o[5];
        Arg x[3] = {Arg(a), Arg(b), Arg(F.add(F.add(a.val, b.val), constant))};
        Sel s[4] = {{MG_SA, one()}, {MG_SB, one()}, {MG_SC, m1()}, {MG_CONST, constant}};
        row(x, 3, s, 4, o);
        return o[2];
    }
    Cell add(const Cell& a, const Cell& b) { return add(a, b, F.zero()); }
    Cell sub(const Cell& a, const Cell& b) {
        Cell o[5];
        Arg x[3] = {Arg(a), Arg(b), Arg(F.sub(a.val, b.val))};
        Sel s[3] = {{MG_SA, one()}, {MG_SB, m1()}, {MG_SC, m1()}};
        row(x, 3, s, 3, o);
        return o[2];
    }
    Cell add_constant(const Cell& a, const Fe& constant) {
        Cell o[5];
        Arg x[3] = {Arg(a), Arg(), Arg(F.add(a.val, constant))};
        Sel s[3] = {{MG_SA, one()}, {MG_SC, m1()}, {MG_CONST, constant}};
        row(x, 3, s, 3, o);
        return o[2];
    }
    void assert_equal(const Cell& a, const Cell& b) { if (want_fixed) copies.push_back(Copy{(uint32_t)a.col, a.row, (uint32_t)b.col, b.row}); }
    Cell assign_bit(uint64_t v) {      // a b - a = 0 with a == b
        Cell o[5];
        Arg x[2] = {Arg(F.u(v)), Arg(F.u(v))};
        Sel s[2] = {{MG_MUL_AB, one()}, {MG_SA, m1()}};
        row(x, 2, s, 2, o);
        if (want_fixed) copies.push_back(Copy{0, o[0].row, 1, o[0].row});
        return o[0];
    }
    Cell select(const Cell& a, const Cell& b, const Cell& cond) {      // a cond - cond b + b - res = 0
        Cell o[5];
        Arg x[5] = {Arg(a), Arg(cond), Arg(cond), Arg(b), Arg(cond.val.is_zero() ? b.val : a.val)};
        Sel s[4] = {{MG_MUL_AB, one()}, {MG_MUL_CD, m1()}, {MG_SD, one()}, {MG_SE, m1()}};
        row(x, 5, s, 4, o);
        return o[4];
    }
    Cell is_equal(const Cell& x, const Cell& y) {
        const Cell d = sub(x, y);
        const Fe bit = F.u(d.val.is_zero() ? 1 : 0), inv = F.inv(d.val);
        Cell o[5], o2[5];
        Arg a[3] = {Arg(d), Arg(inv), Arg(bit)};
        Sel s[3] = {{MG_MUL_AB, one()}, {MG_SC, one()}, {MG_CONST, m1()}};      // d inv + bit - 1 = 0
        row(a, 3, s, 3, o);
        Arg b[2] = {Arg(d), Arg(o[2])};
        Sel s2[1] = {{MG_MUL_AB, one()}};                                       // d bit = 0
        row(b, 2, s2, 1, o2);
        return o[2];
    }
    // s = q 2^width + r over the integers (the value is far below p)
    void div_mod(const Cell& s, unsigned width, Cell& q, Cell& r) {
        Fe qv{{0, 0, 0, 0}}, rv{{0, 0, 0, 0}};
        const unsigned wl = width >> 6, wb = width & 63;      // (width < 256)
        for (unsigned i = 0; i < 4; i++) {
            if (i < wl) rv.v[i] = s.val.v[i];
            else if (i == wl && wb) rv.v[i] = s.val.v[i] & (((uint64_t)1 << wb) - 1);
            if (i + wl < 4) {
                qv.v[i] = s.val.v[i + wl] >> wb;
                if (wb && i + wl + 1 < 4) qv.v[i] |= s.val.v[i + wl + 1] << (64 - wb);
            }
        }
        Cell o[5];
        Arg a[3] = {Arg(qv), Arg(rv), Arg(s)};
        Sel sl[3] = {{MG_SA, F.pow2(width)}, {MG_SB, one()}, {MG_SC, m1()}};
        row(a, 3, sl, 3, o);
        q = o[0];
        r = o[1];
    }
    std::vector<Cell> to_bits(const Cell& v, unsigned nbits) {
        std::vector<Cell> bits;
        for (unsigned i = 0; i < nbits; i++) bits.push_back(assign_bit((v.val.v[i >> 6] >> (i & 63)) & 1));
        Fe acc = F.zero();
        Cell o[5];
        for (unsigned g = 0; g < nbits; g += 4) {      // four bits a row, the running value carried through e / e(next row)
            const unsigned cnt = std::min(4u, nbits - g);
            Arg a[5];
            Sel s[6];
            int ns = 0;
            s[ns++] = {MG_SE, one()};
            s[ns++] = {MG_NEXT, m1()};
            for (unsigned i = 0; i < cnt; i++) {
                a[i] = Arg(bits[g + i]);
                s[ns++] = {MG_SA + (int)i, F.pow2(g + i)};
            }
            a[4] = Arg(acc);
            row(a, 5, s, ns, o);
            for (unsigned i = 0; i < cnt; i++)
                if (!bits[g + i].val.is_zero()) acc = F.add(acc, F.pow2(g + i));
        }
        Arg a[5];
        a[4] = Arg(acc);
        row(a, 5, nullptr, 0, o);
        assert_equal(o[4], v);
        return bits;
    }
    // RangeChip::assign(value, 8-bit sub-limbs, bit_len): sub-limbs four to a row (tagged lookups on a..d), a 6-bit overflow limb on its
    // own row, the running sum carried through e.  value < 2^bit_len <= 2^128.
    Cell range_assign(u128 value, unsigned bit_len) {
        const unsigned nsub = bit_len / 8, rem = bit_len % 8;
        u128 acc = 0;
        Cell o[5];
        for (unsigned g = 0; g < nsub; g += 4) {
            Arg a[5];
            Sel s[8];
            int ns = 0;
            s[ns++] = {MG_SE, one()};
            s[ns++] = {MG_NEXT, m1()};
            s[ns++] = {RC_S_COMPOSITION, one()};
            s[ns++] = {RC_TAG_COMPOSITION, F.u(range_tag(8))};
            u128 part = 0;
            for (unsigned i = 0; i < 4; i++) {
                const uint64_t sub = g + i < nsub ? (uint64_t)((value >> (8 * (g + i))) & 0xFF) : 0;
                a[i] = Arg(F.u(sub));
                s[ns++] = {MG_SA + (int)i, g + i < nsub ? F.pow2(8 * (g + i)) : F.zero()};
                if (g + i < nsub) part += (u128)sub << (8 * (g + i));
            }
            a[4] = Arg(F.from_u128(acc));
            row(a, 5, s, ns, o);
            acc += part;
        }
        if (rem) {
            const uint64_t top = (uint64_t)(value >> (8 * nsub));
            Arg a[5];
            a[0] = Arg(F.u(top));
            a[4] = Arg(F.from_u128(acc));
            Sel s[5] = {{MG_SA, F.pow2(8 * nsub)}, {MG_SE, one()}, {MG_NEXT, m1()}, {RC_S_OVERFLOW, one()}, {RC_TAG_OVERFLOW, F.u(range_tag(rem))}};
            row(a, 5, s, 5, o);
            acc += (u128)top << (8 * nsub);
        }
        Arg a[5];
        a[4] = Arg(F.from_u128(value));
        row(a, 5, nullptr, 0, o);
        return o[4];
    }
};

unsigned synth_threads();

// ---- BigIntChip ----
struct BigIntChip {
    Layouter& lay;
    size_t num_limbs;
    std::vector<Cell> assign_integer(const Big& x) {
        std::vector<Cell> out;
        for (uint64_t v : big_limbs(x, num_limbs)) out.push_back(lay.range_assign(v, LIMB_WIDTH));
        return out;
    }
    std::vector<Cell> assign_constant(const Big& x) {
        std::vector<Cell> out;
        for (uint64_t v : big_limbs(x, num_limbs)) out.push_back(lay.assign_constant(lay.F.u(v)));
        return out;
    }
    // src/big_integer/chip.rs:389-422: limb i of the product = sum_{j + k = i} a_j b_k by a chain of mul_add rows
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
    // src/big_integer/chip.rs:825-898 (is_equal_muled) + the final assertion: a - b + word_max carried limb by limb
    void assert_equal_muled(const std::vector<Cell>& a, const std::vector<Cell>& b, size_t n1, size_t n2) {
        Fld& F = lay.F;
        const size_t min_n = std::min(n1, n2);
        // word_max = min_n * limb_max^2 + limb_max (compute_mul_word_max): below 2^134 for 32 limbs
        const u128 limb_max = ~(uint64_t)0;
        Fe word_max;
        {
            const Fe lm2 = F.mul(F.from_u128(limb_max), F.from_u128(limb_max));
            word_max = F.add(F.mul(F.u(min_n), lm2), F.from_u128(limb_max));
        }
        unsigned wm_bits = 256;      // bit length of 2 * word_max
        {
            Fe two_wm = F.add(word_max, word_max);
            while (wm_bits && !((two_wm.v[(wm_bits - 1) >> 6] >> ((wm_bits - 1) & 63)) & 1)) wm_bits--;
        }
        const unsigned carry_bits = wm_bits - LIMB_WIDTH;
        Cell accumulated_extra = lay.assign_constant(F.zero());
        Cell carry = lay.assign_constant(F.zero());
        Cell eq_bit = lay.assign_bit(1);
        const size_t num = n1 + n2 - 1;
        for (size_t i = 0; i < num; i++) {
            const Cell a_b = lay.sub(a[i], b[i]);
            const Cell s = lay.add(a_b, carry, word_max);
            Cell new_carry, c, q_acc, mod_acc;
            lay.div_mod(s, LIMB_WIDTH, new_carry, c);
            accumulated_extra = lay.add_constant(accumulated_extra, word_max);
            lay.div_mod(accumulated_extra, LIMB_WIDTH, q_acc, mod_acc);
            eq_bit = lay.mul(Arg(eq_bit), Arg(lay.is_equal(c, mod_acc)));
            accumulated_extra = q_acc;
            if (i + 1 < num) {
                const u128 nc = ((u128)new_carry.val.v[1] << 64) | new_carry.val.v[0];
                const Cell ranged = lay.range_assign(nc, carry_bits);
                eq_bit = lay.mul(Arg(eq_bit), Arg(lay.is_equal(new_carry, ranged)));
            } else {
                eq_bit = lay.mul(Arg(eq_bit), Arg(lay.is_equal(new_carry, accumulated_extra)));
            }
            carry = new_carry;
        }
        lay.assert_equal(eq_bit, lay.assign_constant(F.u(1)));
    }
    std::vector<Cell> pow_mod_threads(std::vector<Cell> acc, std::vector<Cell> squared, const std::vector<Cell>& e_bits, const std::vector<Cell>& n, const Big& n_big);
    Big to_big(const std::vector<Cell>& limbs) const {
        Big r;
        for (auto& c : limbs) r.push_back(c.val.v[0]);
        return r;
    }
    // src/big_integer/chip.rs:545-632
    std::vector<Cell> mul_mod(const std::vector<Cell>& a, const std::vector<Cell>& b, const std::vector<Cell>& n, const Big& n_big) {
        Big q_big, r_big;
        big_divmod(big_mul(to_big(a), to_big(b)), n_big, q_big, r_big);
        return mul_mod_rows(a, b, n, q_big, r_big);
    }
    // the rows of a * b = q * n + r for a quotient and remainder already known
    std::vector<Cell> mul_mod_rows(const std::vector<Cell>& a, const std::vector<Cell>& b, const std::vector<Cell>& n, const Big& q_big, const Big& r_big) {
        const size_t n1 = a.size(), n2 = b.size();
        std::vector<Cell> q, r;
        for (uint64_t v : big_limbs(q_big, n2)) q.push_back(lay.range_assign(v, LIMB_WIDTH));
        for (uint64_t v : big_limbs(r_big, n1)) r.push_back(lay.range_assign(v, LIMB_WIDTH));
        const std::vector<Cell> ab = mul(a, b), qn = mul(q, n);
        std::vector<Cell> eq_b;
        for (size_t i = 0; i + 1 < n1 + n2; i++) eq_b.push_back(i < n1 ? lay.add(qn[i], r[i]) : qn[i]);
        assert_equal_muled(ab, eq_b, n1, n2);
        return r;
    }
    // src/big_integer/chip.rs:667-699: per exponent bit (LSB first) acc * squared, select, squared^2
    std::vector<Cell> pow_mod(const std::vector<Cell>& a, const std::vector<Cell>& e_bits, const std::vector<Cell>& n, const Big& n_big) {
        std::vector<Cell> acc;
        for (uint64_t v : big_limbs(Big{1}, num_limbs)) acc.push_back(lay.range_assign(v, LIMB_WIDTH));      // assign_constant_fresh(1)
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
    std::atomic<bool> ok{true};
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
    if (!ok) throw std::runtime_error("pow_mod: a region's length depends on its values");
    lay.nrows = with_tail ? tail_end : end_row;
    lay.after_pow_done = with_tail;
    return powed;
}

// ---- PoseidonChip rows: x^5 as three multiplication rows, every MDS output as two rows of a five-term sum ----
struct PoseidonRows {
    Layouter& lay;
    const PoseidonSpec& sp;
    Cell pow5(const Cell& x) {
        const Cell x2 = lay.mul(Arg(x), Arg(x));
        const Cell x4 = lay.mul(Arg(x2), Arg(x2));
        return lay.mul(Arg(x4), Arg(x));
    }
    Cell linear(const std::vector<Cell>& st, uint32_t mds_row, const Fe& constant) {
        Fld& F = lay.F;
        const std::vector<Fe>& coeffs = sp.mds[mds_row];
        Fe part = F.zero();
        for (int i = 0; i < 4; i++) part = F.add(part, sp.mds_mul(mds_row, i, st[i].val));
        Cell o[5];
        Arg a[5] = {Arg(st[0]), Arg(st[1]), Arg(st[2]), Arg(st[3]), Arg(F.zero())};
        Sel s[6] = {{MG_SA, coeffs[0]}, {MG_SB, coeffs[1]}, {MG_SC, coeffs[2]}, {MG_SD, coeffs[3]}, {MG_SE, lay.one()}, {MG_NEXT, lay.m1()}};
        lay.row(a, 5, s, 6, o);
        const bool five = st.size() > 4;
        const Fe total = F.add(F.add(part, five ? sp.mds_mul(mds_row, 4, st[4].val) : F.zero()), constant);
        Arg b[5] = {five ? Arg(st[4]) : Arg(), Arg(total), Arg(), Arg(), Arg(part)};
        Sel s2[4] = {{MG_SA, five ? coeffs[4] : F.zero()}, {MG_SB, lay.m1()}, {MG_SE, lay.one()}, {MG_CONST, constant}};
        lay.row(b, 5, s2, 4, o);
        return o[1];
    }
    // Round r: add constants, S-box (all words in a full round, word 0 in a partial one), MDS.  The constants of round r + 1 ride on
    // round r's linear layer, so only the first round adds them on rows of their own.
    std::vector<Cell> permutation(std::vector<Cell> st) {
        const uint32_t half = sp.r_f / 2, rounds = (uint32_t)sp.constants.size();
        for (size_t i = 0; i < st.size(); i++) st[i] = lay.add_constant(st[i], sp.constants[0][i]);
        for (uint32_t r = 0; r < rounds; r++) {
            const bool full = r < half || r >= half + sp.r_p;
            std::vector<Cell> s = st;
            if (full) for (auto& x : s) x = pow5(x);
            else s[0] = pow5(st[0]);
            std::vector<Cell> nx;
            for (uint32_t i = 0; i < sp.t; i++) nx.push_back(linear(s, i, r + 1 < rounds ? sp.constants[r + 1][i] : lay.F.zero()));
            st = nx;
        }
        return st;
    }
};

struct NativeCipher {      // PoseidonCipher::{initial_state, encrypt} (src/encryption/poseidon_enc.rs:66-133), MESSAGE_CAPACITY = 2, T = 5
    const PoseidonSpec& sp;
    Fe key[2];
    std::vector<Fe> encrypt(const std::vector<Fe>& message, uint64_t nonce) const {
        const Fld& F = sp.F;
        std::vector<Fe> st = {F.zero(), F.zero(), key[0], key[1], F.u(nonce)};
        st = sp.permute(st);
        std::vector<Fe> cipher;
        for (size_t i = 0; i < message.size(); i++) {
            st[1 + i] = F.add(st[1 + i], message[i]);
            cipher.push_back(st[1 + i]);
        }
        st = sp.permute(st);
        cipher.push_back(st[1]);
        return cipher;
    }
};

// src/lib.rs:179-206 / benches/mod_pow.rs:63-110: assign n, e, x; x^e mod n in-circuit; equal to the native big_pow_mod
// the expected value x^e mod n as constants, constrained equal to the exponentiation's result (src/lib.rs:205-219)
std::vector<Cell> rsa_expected_rows(Layouter& lay, const std::vector<Cell>& powed) {
    BigIntChip chip{lay, powed.size()};
    Big want = chip.to_big(powed);
    big_trim(want);
    const std::vector<Cell> valid = chip.assign_constant(want);
    for (size_t i = 0; i < powed.size(); i++) lay.assert_equal(powed[i], valid[i]);
    return valid;
}

std::vector<Cell> rsa_region(Layouter& lay, const Big& n_big, uint64_t e, const Big& x, unsigned exp_bits, size_t num_limbs, Big& want) {
    BigIntChip chip{lay, num_limbs};
    const std::vector<Cell> n_limbs = chip.assign_integer(n_big);
    const Cell e_cell = exp_bits % 8 == 0 ? lay.range_assign(e, 8 * ((exp_bits + 7) / 8)) : lay.assign_value(lay.F.u(e));
    const std::vector<Cell> e_bits = lay.to_bits(e_cell, exp_bits);
    const std::vector<Cell> x_limbs = chip.assign_integer(x);
    const std::vector<Cell> powed = chip.pow_mod(x_limbs, e_bits, n_limbs, n_big);
    // the native big_pow_mod(x, e, n) the reference assigns as the expected value: the rows above hold exactly its square-and-multiply chain (every
    // mul_mod's remainder came from the same big-integer division), so its value is read off them instead of being computed a second time
    want = chip.to_big(powed);
    big_trim(want);
    if (lay.after_pow_done) return powed;      // (the rows behind the exponentiation were written beside it)
    return rsa_expected_rows(lay, powed);
}

// src/lib.rs:261-316 / src/encryption/chip.rs:72-110: the Poseidon cipher in-circuit, constrained equal to the native one
std::vector<Cell> cipher_region(Layouter& lay, const PoseidonSpec& spec, const Fe key_vals[2], const std::vector<Fe>& message, const Cell* key_cells) {
    PoseidonRows rows{lay, spec};
    NativeCipher native{spec, {key_vals[0], key_vals[1]}};
    std::vector<Cell> expected;
    for (auto& v : native.encrypt(message, 1)) expected.push_back(lay.assign_value(v));
    std::vector<Cell> st = {lay.assign_constant(lay.F.zero()), lay.assign_constant(lay.F.zero()), lay.assign_value(key_vals[0]), lay.assign_value(key_vals[1]),
                            lay.assign_constant(lay.F.u(1))};
    if (key_cells) {
        lay.assert_equal(st[2], key_cells[0]);
        lay.assert_equal(st[3], key_cells[1]);
    }
    st = rows.permutation(st);
    std::vector<Cell> msg;
    for (auto& m : message) msg.push_back(lay.assign_value(m));
    std::vector<Cell> nx = {st[0]};
    for (uint32_t i = 0; i + 1 < spec.t; i++) nx.push_back(i < msg.size() ? lay.add(st[1 + i], msg[i]) : st[1 + i]);
    st = nx;
    std::vector<Cell> cipher(st.begin() + 1, st.begin() + 1 + (long)msg.size());
    st = rows.permutation(st);
    cipher.push_back(st[1]);
    for (size_t i = 0; i < cipher.size(); i++) lay.assert_equal(cipher[i], expected[i]);
    return cipher;
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
    Fld F{f};
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
        for (auto& c : cipher_region(lay, spec, key, message, nullptr)) cipher_vals.push_back(c.val);
    } else {
        if (!in->n || !in->x || in->bits_len % LIMB_WIDTH || in->bits_len == 0 || in->bits_len > 8192 || in->exp_bits == 0 || in->exp_bits > 64) return DEHALO_ERR_INVALID;
        const size_t num_limbs = in->bits_len / LIMB_WIDTH;
        Big n_big(in->n, in->n + num_limbs), x(in->x, in->x + num_limbs);
        { Big t0 = n_big; big_trim(t0); if (t0.empty()) return DEHALO_ERR_INVALID; }
        if (in->exp_bits < 64 && (in->e >> in->exp_bits)) return DEHALO_ERR_INVALID;
        // hash region: limbs packed three to a field element (src/lib.rs:222-249), sponge with RATE 4 (src/hash/chip.rs:63-85); then the cipher keyed by the digest
        auto hash_and_cipher = [&](Layouter& L, const std::vector<Cell>& rsa_out) {
            const std::shared_ptr<const PoseidonSpec> spec_ref = poseidon_spec(f, t, r_f, r_p);
            const PoseidonSpec& spec = *spec_ref;
            PoseidonRows rows{L, spec};
            const Cell base1 = L.assign_constant(F.pow2(LIMB_WIDTH));
            const Cell base2 = L.mul(Arg(base1), Arg(base1));
            std::vector<Cell> inputs;
            for (size_t i = 0; i < rsa_out.size() / 3; i++) {
                const Cell a = L.mul_add(Arg(rsa_out[3 * i + 1]), Arg(base1), Arg(rsa_out[3 * i]));
                inputs.push_back(L.mul_add(Arg(rsa_out[3 * i + 2]), Arg(base2), Arg(a)));
            }
            if (rsa_out.size() % 3 == 2) inputs.push_back(L.mul_add(Arg(rsa_out[rsa_out.size() - 1]), Arg(base1), Arg(rsa_out[rsa_out.size() - 2])));
            std::vector<Cell> state = {L.assign_constant(F.pow2(64))};      // Poseidon::new: capacity word 2^64
            for (uint32_t i = 1; i < t; i++) state.push_back(L.assign_constant(F.zero()));
            for (size_t c0 = 0; c0 < inputs.size(); c0 += rate) {
                const size_t cnt = std::min<size_t>(rate, inputs.size() - c0);
                std::vector<Cell> nxt = {state[0]};
                for (uint32_t i = 0; i < rate; i++) nxt.push_back(i < cnt ? L.add(state[1 + i], inputs[c0 + i]) : state[1 + i]);
                if (cnt < rate) nxt[1 + cnt] = L.add_constant(nxt[1 + cnt], F.u(1));      // padding: + 1 after the last input
                state = rows.permutation(nxt);
            }
            if (inputs.size() % rate == 0) {
                std::vector<Cell> nxt = state;
                nxt[1] = L.add_constant(state[1], F.u(1));
                state = rows.permutation(nxt);
            }
            const Cell key_cells[2] = {state[1], state[2]};
            const Fe key_vals[2] = {state[1].val, state[2].val};
            for (auto& c : cipher_region(L, spec, key_vals, message, key_cells)) cipher_vals.push_back(c.val);
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
            const unsigned lens[4] = {COMPOSITION_BITS[0], COMPOSITION_BITS[1], COMPOSITION_BITS[2], OVERFLOW_BITS};
            for (unsigned ti = 0; ti < 4; ti++)
                for (uint64_t v = 0; v < ((uint64_t)1 << lens[ti]); v++, r++) {
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
} catch (...) { return DEHALO_ERR_OOM; }
