"""Witness generation for the reference's circuits over the MainGate + RangeChip shape: the values
`Circuit::synthesize` writes into the 5 advice columns before `create_proof` commits them -- so that the prover can be fed
the REAL value distribution of a 2048-bit RSA delay-encryption witness instead of a synthetic one.

What is restated, and from where:
  * `big_pow_mod`                       src/big_integer/utils.rs:2-17 (native reference value; pinned by the reference's RSA
                                        vectors src/rsa/chip.rs:706-716 through the signature identity s^65537 mod n)
  * `BigIntChip::{mul, mul_mod, pow_mod, assert_equal_muled}`
                                        src/big_integer/chip.rs:389-422, 545-632, 667-699, 825-898: 64-bit limbs, school-book
                                        `mul_add` rows, witness quotient / remainder with 8-bit range-decomposed limbs, the
                                        carried equality check with `word_max` offsets and 70-bit range-checked carries
  * the top-level flow                  src/lib.rs:164-318: x^e mod n -> 11 packed field elements -> Poseidon sponge (RATE 4)
                                        -> 2-element key -> Poseidon cipher over the message (src/encryption/poseidon_enc.rs:86-133,
                                        src/hash/chip.rs:63-85)
  * Poseidon parameters and permutation src/poseidon/grain.rs:12-157 (Grain LFSR), spec.rs:170-180 (Cauchy MDS),
                                        permutation.rs:60-80 (rounds); pinned by the reference's own known-answer vectors
                                        src/poseidon/permutation.rs:154-158,190-196
The cell layout is NOT halo2wrong's (MainGate's region layout is upstream code that is not in the container): rows are laid
out by the small layouter below over the same gate (plonk.maingate_cs) -- every row satisfies the gate, every range row its
lookup, every reuse of a value is a copy constraint -- so the result is a valid witness + fixed columns + permutation for THIS
constraint system, with the reference's row budget per operation (about 3.3 k rows per mul_mod against the reference's 3.99 k)
and its value classes: 8-bit sub-limbs, 64-bit limbs, <= 134-bit accumulators, 0/1 bits, full-width Poseidon states.
Host-side Python integers: the reference's synthesize is single-threaded BigUint code too.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import plonk
from .circuits import SyntheticCircuit
from .keygen import ints_to_array

LIMB_WIDTH, BITS_LEN = 64, 2048            # src/lib.rs:122-123
NUM_LIMBS = BITS_LEN // LIMB_WIDTH


def big_pow_mod(a: int, b: int, n: int) -> int:
    """src/big_integer/utils.rs:2-17 (recursive square-and-multiply; b = 0 -> 1)."""
    if b == 0:
        return 1
    is_odd = b % 2 == 1
    b = b - 1 if is_odd else b
    x = big_pow_mod(a, b // 2, n)
    x2 = x * x % n
    return a * x2 % n if is_odd else x2


# ---- Poseidon (native) ---------------------------------------------------------------------------------------
class Grain:
    """src/poseidon/grain.rs:12-157: 80-bit LFSR seeded with the parameter description, 160 warm-up bits, output bits taken in
    pairs (a pair whose first bit is 0 is discarded), field elements MSB first."""

    def __init__(self, p: int, t: int, r_f: int, r_p: int):
        self.p, self.nbits = p, p.bit_length()
        bits: List[int] = []
        for width, v in ((2, 1), (4, 0), (12, self.nbits), (12, t), (10, r_f), (10, r_p), (30, (1 << 30) - 1)):
            bits += [(v >> i) & 1 for i in reversed(range(width))]
        self.bits = bits
        for _ in range(160):
            self._new_bit()

    def _new_bit(self) -> int:
        b = self.bits
        nb = b[0] ^ b[62] ^ b[51] ^ b[38] ^ b[23] ^ b[13]
        b.pop(0)
        b.append(nb)
        return nb

    def _bit(self) -> int:
        while not self._new_bit():
            self._new_bit()
        return self._new_bit()

    def _draw(self) -> int:
        v = 0
        for _ in range(self.nbits):
            v = (v << 1) | self._bit()
        return v

    def field_element(self) -> int:                       # with rejection: round constants
        while True:
            v = self._draw()
            if v < self.p:
                return v

    def field_element_mod(self) -> int:                   # without rejection: the MDS x, y
        return self._draw() % self.p


class PoseidonSpec:
    """Spec::new(r_f, r_p) as values: round constants and the Cauchy MDS 1 / (x_i + y_j) (src/poseidon/spec.rs:170-180, 310-324)."""

    def __init__(self, p: int, t: int, r_f: int, r_p: int):
        g = Grain(p, t, r_f, r_p)
        self.p, self.t, self.r_f, self.r_p = p, t, r_f, r_p
        self.constants = [[g.field_element() for _ in range(t)] for _ in range(r_f + r_p)]
        xs = [g.field_element_mod() for _ in range(t)]
        ys = [g.field_element_mod() for _ in range(t)]
        self.mds = [[pow((x + y) % p, -1, p) for y in ys] for x in xs]

    def permute(self, state: Sequence[int]) -> List[int]:
        """src/poseidon/permutation.rs:60-80 (the un-optimised round function; the reference checks its optimised one against it)."""
        p, st, half = self.p, list(state), self.r_f // 2
        for r, rc in enumerate(self.constants):
            st = [(e + c) % p for e, c in zip(st, rc)]
            if r < half or r >= half + self.r_p:
                st = [pow(e, 5, p) for e in st]
            else:
                st[0] = pow(st[0], 5, p)
            st = [sum(m * v for m, v in zip(row, st)) % p for row in self.mds]
        return st


# ---- a MainGate / RangeChip layouter -----------------------------------------------------------------------------
@dataclass
class Cell:
    col: int
    row: int
    val: int


class Layouter:
    """Rows of the gate  a sa + b sb + c sc + d sd + e se + a b s_mul_ab + c d s_mul_cd + e(next row) s_next + s_constant = 0
    plus the RangeChip's tagged lookups; `copies` are the permutation's equalities."""

    def __init__(self, p: int):
        self.p = p
        self.adv: List[List[int]] = [[] for _ in range(5)]
        self.fix: List[List[int]] = [[] for _ in range(15)]
        self.copies: List[Tuple[int, int, int, int]] = []

    @property
    def rows(self) -> int:
        return len(self.adv[0])

    def row(self, cells: Sequence, sel: Optional[dict] = None) -> List[Cell]:
        r, out = self.rows, []
        for i in range(5):
            c = cells[i] if i < len(cells) else None
            if isinstance(c, Cell):
                self.copies.append((c.col, c.row, i, r))
                v = c.val
            else:
                v = 0 if c is None else c % self.p
            self.adv[i].append(v)
            out.append(Cell(i, r, v))
        for col in self.fix:
            col.append(0)
        if sel:
            for k, v in sel.items():
                self.fix[k][r] = v % self.p
        return out

    # MainGate instructions, one row each
    def assign_value(self, v: int) -> Cell:
        return self.row([v])[0]

    def assign_constant(self, v: int) -> Cell:
        return self.row([v], {plonk.MG_SA: 1, plonk.MG_CONST: -v})[0]

    def mul_add(self, a, b, c) -> Cell:
        va, vb, vc = (x.val if isinstance(x, Cell) else (x or 0) for x in (a, b, c))
        return self.row([a, b, c, (va * vb + vc) % self.p], {plonk.MG_MUL_AB: 1, plonk.MG_SC: 1, plonk.MG_SD: -1})[3]

    def mul(self, a, b) -> Cell:
        return self.mul_add(a, b, None)

    def add(self, a: Cell, b: Cell, constant: int = 0) -> Cell:
        return self.row([a, b, (a.val + b.val + constant) % self.p], {plonk.MG_SA: 1, plonk.MG_SB: 1, plonk.MG_SC: -1, plonk.MG_CONST: constant})[2]

    def sub(self, a: Cell, b: Cell) -> Cell:
        return self.row([a, b, (a.val - b.val) % self.p], {plonk.MG_SA: 1, plonk.MG_SB: -1, plonk.MG_SC: -1})[2]

    def add_constant(self, a: Cell, constant: int) -> Cell:
        return self.row([a, None, (a.val + constant) % self.p], {plonk.MG_SA: 1, plonk.MG_SC: -1, plonk.MG_CONST: constant})[2]

    def assert_equal(self, a: Cell, b: Cell):
        assert a.val == b.val, "assert_equal on different values"
        self.copies.append((a.col, a.row, b.col, b.row))

    def assign_bit(self, v: int) -> Cell:
        cells = self.row([v, v], {plonk.MG_MUL_AB: 1, plonk.MG_SA: -1})          # a b - a = 0 with a == b
        self.copies.append((0, cells[0].row, 1, cells[0].row))
        return cells[0]

    def select(self, a: Cell, b: Cell, cond: Cell) -> Cell:
        """cond a + (1 - cond) b  (MainGate::select): a cond - cond b + b - res = 0."""
        res = a.val if cond.val else b.val
        return self.row([a, cond, cond, b, res], {plonk.MG_MUL_AB: 1, plonk.MG_MUL_CD: -1, plonk.MG_SD: 1, plonk.MG_SE: -1})[4]

    def is_equal(self, x: Cell, y: Cell) -> Cell:
        d = self.sub(x, y)
        bit = 1 if d.val == 0 else 0
        inv = pow(d.val, -1, self.p) if d.val else 0
        cells = self.row([d, inv, bit], {plonk.MG_MUL_AB: 1, plonk.MG_SC: 1, plonk.MG_CONST: -1})     # d inv + bit - 1 = 0
        self.row([d, cells[2]], {plonk.MG_MUL_AB: 1})                                                  # d bit = 0
        return cells[2]

    def and_(self, x: Cell, y: Cell) -> Cell:
        return self.mul(x, y)

    def div_mod(self, s: Cell, width: int) -> Tuple[Cell, Cell]:
        q, r = s.val >> width, s.val & ((1 << width) - 1)
        cells = self.row([q, r, s], {plonk.MG_SA: 1 << width, plonk.MG_SB: 1, plonk.MG_SC: -1})
        return cells[0], cells[1]

    def to_bits(self, v: Cell, nbits: int) -> List[Cell]:
        bits = [self.assign_bit((v.val >> i) & 1) for i in range(nbits)]
        acc = 0
        for g in range(0, nbits, 4):                          # four bits a row, the running value carried through e / e(next row)
            chunk = bits[g:g + 4]
            sel = {plonk.MG_SE: 1, plonk.MG_NEXT: -1}
            for i in range(len(chunk)):
                sel[plonk.MG_SA + i] = 1 << (g + i)
            self.row(list(chunk) + [None] * (4 - len(chunk)) + [acc], sel)
            acc += sum(b.val << (g + i) for i, b in enumerate(chunk))
        total = self.row([None, None, None, None, acc])[4]
        self.assert_equal(total, v)
        return bits

    # RangeChip::assign(value, sublimb_bits = 8, bit_len): 8-bit sub-limbs four to a row (tagged lookups on a..d), a
    # 6-bit overflow limb on its own row, the running sum carried through e
    def range_assign(self, value: int, bit_len: int) -> Cell:
        assert 0 <= value < (1 << bit_len)
        nsub, rem = bit_len // 8, bit_len % 8
        assert rem in (0,) + plonk.OVERFLOW_BIT_LENS, "unsupported overflow width"
        acc = 0
        for g in range(0, nsub, 4):
            subs = [(value >> (8 * (g + i))) & 0xFF if g + i < nsub else 0 for i in range(4)]
            sel = {plonk.MG_SE: 1, plonk.MG_NEXT: -1, plonk.RC_S_COMPOSITION: 1, plonk.RC_TAG_COMPOSITION: plonk.range_tag(8)}
            for i in range(4):
                sel[plonk.MG_SA + i] = (1 << (8 * (g + i))) if g + i < nsub else 0
            self.row(subs + [acc], sel)
            acc += sum(s << (8 * (g + i)) for i, s in enumerate(subs))
        if rem:
            top = value >> (8 * nsub)
            self.row([top, None, None, None, acc], {plonk.MG_SA: 1 << (8 * nsub), plonk.MG_SE: 1, plonk.MG_NEXT: -1, plonk.RC_S_OVERFLOW: 1,
                                                    plonk.RC_TAG_OVERFLOW: plonk.range_tag(rem)})
            acc += top << (8 * nsub)
        assert acc == value
        return self.row([None, None, None, None, value])[4]


# ---- BigIntChip ------------------------------------------------------------------------------------------------
def limbs_of(x: int, n: int = NUM_LIMBS) -> List[int]:
    return [(x >> (LIMB_WIDTH * i)) & ((1 << LIMB_WIDTH) - 1) for i in range(n)]


class BigIntChip:
    def __init__(self, lay: Layouter):
        self.lay = lay

    def assign_integer(self, x: int, n: int = NUM_LIMBS) -> List[Cell]:
        return [self.lay.range_assign(v, LIMB_WIDTH) for v in limbs_of(x, n)]

    def assign_constant(self, x: int, n: int = NUM_LIMBS) -> List[Cell]:
        return [self.lay.assign_constant(v) for v in limbs_of(x, n)]

    def mul(self, a: List[Cell], b: List[Cell]) -> List[Cell]:
        """src/big_integer/chip.rs:389-422: limb i of the product = sum_{j + k = i} a_j b_k by a chain of mul_add rows."""
        d0, d1, lay, out = len(a), len(b), self.lay, []
        for i in range(d0 + d1 - 1):
            acc = lay.assign_constant(0)
            j = 0 if d1 >= i + 1 else i + 1 - d1
            while j < d0 and j <= i:
                acc = lay.mul_add(a[j], b[i - j], acc)
                j += 1
            out.append(acc)
        return out

    def assert_equal_muled(self, a: List[Cell], b: List[Cell], n1: int, n2: int):
        """src/big_integer/chip.rs:825-898 (is_equal_muled) + the final assertion: a - b + word_max carried limb by limb."""
        lay, p = self.lay, self.lay.p
        min_n = min(n1, n2)
        limb_max = (1 << LIMB_WIDTH) - 1
        word_max = min_n * limb_max * limb_max + limb_max                          # compute_mul_word_max
        carry_bits = (2 * word_max).bit_length() - LIMB_WIDTH
        accumulated_extra = lay.assign_constant(0)
        carry = lay.assign_constant(0)
        eq_bit = lay.assign_bit(1)
        num_limbs = n1 + n2 - 1
        for i in range(num_limbs):
            a_b = lay.sub(a[i], b[i])
            s = lay.add(a_b, carry, word_max)
            assert s.val < p // 2, "carried sum left the integers"
            new_carry, c = lay.div_mod(s, LIMB_WIDTH)
            accumulated_extra = lay.add_constant(accumulated_extra, word_max)
            q_acc, mod_acc = lay.div_mod(accumulated_extra, LIMB_WIDTH)
            eq_bit = lay.and_(eq_bit, lay.is_equal(c, mod_acc))
            accumulated_extra = q_acc
            if i < num_limbs - 1:
                ranged = lay.range_assign(new_carry.val, carry_bits)
                eq_bit = lay.and_(eq_bit, lay.is_equal(new_carry, ranged))
            else:
                eq_bit = lay.and_(eq_bit, lay.is_equal(new_carry, accumulated_extra))
            carry = new_carry
        lay.assert_equal(eq_bit, lay.assign_constant(1))

    def mul_mod(self, a: List[Cell], b: List[Cell], n: List[Cell], n_big: int) -> List[Cell]:
        """src/big_integer/chip.rs:545-632."""
        lay = self.lay
        to_big = lambda limbs: sum(c.val << (LIMB_WIDTH * i) for i, c in enumerate(limbs))
        full = to_big(a) * to_big(b)
        q_big, r_big = full // n_big, full % n_big
        n1, n2 = len(a), len(b)
        q = [lay.range_assign(v, LIMB_WIDTH) for v in limbs_of(q_big, n2)]
        r = [lay.range_assign(v, LIMB_WIDTH) for v in limbs_of(r_big, n1)]
        ab, qn = self.mul(a, b), self.mul(q, n)
        eq_b = [lay.add(qn[i], r[i]) if i < n1 else qn[i] for i in range(n1 + n2 - 1)]
        self.assert_equal_muled(ab, eq_b, n1, n2)
        return r

    def pow_mod(self, a: List[Cell], e_bits: List[Cell], n: List[Cell], n_big: int) -> List[Cell]:
        """src/big_integer/chip.rs:667-699: per exponent bit (LSB first) acc * squared, select, squared^2."""
        lay = self.lay
        acc = [self.lay.range_assign(v, LIMB_WIDTH) for v in limbs_of(1)]        # assign_constant_fresh(1)
        squared = a
        for bit in e_bits:
            muled = self.mul_mod(acc, squared, n, n_big)
            acc = [lay.select(muled[j], acc[j], bit) for j in range(len(acc))]
            squared = self.mul_mod(squared, squared, n, n_big)
        return acc


# ---- PoseidonChip rows ---------------------------------------------------------------------------------------------
class PoseidonRows:
    """The permutation as MainGate rows (the reference's PoseidonChip, src/poseidon/chip.rs:199-378, builds it from MainGate
    mul / mul_add_constant / compose): x^5 as three multiplication rows, every MDS output as two rows of a five-term sum."""

    def __init__(self, lay: Layouter, spec: PoseidonSpec):
        self.lay, self.spec = lay, spec

    def pow5(self, x: Cell) -> Cell:
        lay = self.lay
        x2 = lay.mul(x, x)
        x4 = lay.mul(x2, x2)
        return lay.mul(x4, x)

    def linear(self, st: List[Cell], coeffs: Sequence[int], constant: int = 0) -> Cell:
        """sum_i coeffs[i] * st[i] + constant over T = 5 cells: four terms + running sum in e, then the fifth term."""
        lay, p = self.lay, self.lay.p
        part = sum(c * s.val for c, s in zip(coeffs[:4], st[:4])) % p
        lay.row(list(st[:4]) + [0], {plonk.MG_SA: coeffs[0], plonk.MG_SB: coeffs[1], plonk.MG_SC: coeffs[2], plonk.MG_SD: coeffs[3], plonk.MG_SE: 1, plonk.MG_NEXT: -1})
        total = (part + (coeffs[4] * st[4].val if len(st) > 4 else 0) + constant) % p
        cells = lay.row([st[4] if len(st) > 4 else None, total, None, None, part], {plonk.MG_SA: coeffs[4] if len(st) > 4 else 0, plonk.MG_SB: -1, plonk.MG_SE: 1, plonk.MG_CONST: constant})
        return cells[1]

    def permutation(self, st: List[Cell]) -> List[Cell]:
        """Round r: add constants, S-box (all words in a full round, word 0 in a partial one), MDS.  The constants of round
        r + 1 ride on round r's linear layer, so only the first round adds them on rows of their own."""
        sp, half = self.spec, self.spec.r_f // 2
        rounds = len(sp.constants)
        st = [self.lay.add_constant(x, c) for x, c in zip(st, sp.constants[0])]
        for r in range(rounds):
            full = r < half or r >= half + sp.r_p
            s = [self.pow5(x) for x in st] if full else [self.pow5(st[0])] + st[1:]
            nxt = sp.constants[r + 1] if r + 1 < rounds else [0] * sp.t
            st = [self.linear(s, row, nxt[i]) for i, row in enumerate(sp.mds)]
        return st


# ---- the circuits ---------------------------------------------------------------------------------------------------
@dataclass
class WitnessInfo:
    rsa_rows: int
    total_rows: int
    rsa_result: int
    cipher: List[int]


def _finish(lay: Layouter, k: int, info: WitnessInfo, range_lookups: bool = True) -> Tuple[SyntheticCircuit, WitnessInfo]:
    cs = plonk.maingate_cs(range_lookups)
    n = 1 << k
    u = n - (cs.blinding_factors() + 1)
    if lay.rows > u - 1:
        raise ValueError("not enough rows available: %d rows need k > %d" % (lay.rows, k))     # upstream: Error::NotEnoughRowsAvailable
    pad = n - lay.rows
    fixed = [col + [0] * pad for col in lay.fix[:cs.num_fixed]]
    if not range_lookups:
        assert not any(any(col) for col in lay.fix[cs.num_fixed:]), "range rows in a MainGate-only circuit"
    for r, (tag, v) in enumerate(plonk.range_table() if range_lookups else []):
        fixed[plonk.RC_T_TAG][r], fixed[plonk.RC_T_VALUE][r] = tag, v
    adv = [col + [0] * pad for col in lay.adv]
    asm = plonk.Assembly(len(cs.permutation_columns), n)
    for c0, r0, c1, r1 in lay.copies:
        asm.copy(c0, r0, c1, r1)
    selectors = [np.array(fixed[plonk.RC_S_COMPOSITION], dtype=bool), np.array(fixed[plonk.RC_S_OVERFLOW], dtype=bool)] if range_lookups else []
    circ = SyntheticCircuit(cs, k, np.stack([ints_to_array(c) for c in fixed]), np.stack([ints_to_array(c) for c in adv]), asm, selectors, lay.rows)
    return circ, info


def rsa_region(lay: Layouter, n_big: int, e: int, x: int, exp_bits: int) -> Tuple[List[Cell], int]:
    """src/lib.rs:179-206 / benches/mod_pow.rs:63-110: assign n, e, x; x^e mod n in-circuit; equal to the native big_pow_mod."""
    chip = BigIntChip(lay)
    n_limbs = chip.assign_integer(n_big)
    e_cell = lay.range_assign(e, 8 * ((exp_bits + 7) // 8)) if exp_bits % 8 == 0 else lay.assign_value(e)
    e_bits = lay.to_bits(e_cell, exp_bits)
    x_limbs = chip.assign_integer(x)
    powed = chip.pow_mod(x_limbs, e_bits, n_limbs, n_big)
    want = big_pow_mod(x, e, n_big)
    valid = chip.assign_constant(want)
    for a, b in zip(powed, valid):
        lay.assert_equal(a, b)
    return valid, want


def mod_pow_witness(p: int, k: int, n_big: int, e: int, x: int, exp_bits: int):
    """benches/mod_pow.rs's RSACircuit (RSA region only): BASELINE configs[2]."""
    lay = Layouter(p)
    _, want = rsa_region(lay, n_big, e, x, exp_bits)
    return _finish(lay, k, WitnessInfo(lay.rows, lay.rows, want, []))


def cipher_region(lay: Layouter, spec: PoseidonSpec, key_vals: Sequence[int], message: Sequence[int], key_cells: Optional[Sequence[Cell]] = None) -> List[Cell]:
    """src/lib.rs:261-316 / src/encryption/chip.rs:72-110: the Poseidon cipher in-circuit, constrained equal to the native one."""
    rows, t = PoseidonRows(lay, spec), spec.t
    native = NativeCipher(spec, list(key_vals))
    expected = [lay.assign_value(v) for v in native.encrypt(list(message), 1)]
    st = [lay.assign_constant(0), lay.assign_constant(0), lay.assign_value(key_vals[0]), lay.assign_value(key_vals[1]), lay.assign_constant(1)]
    if key_cells is not None:
        lay.assert_equal(st[2], key_cells[0])
        lay.assert_equal(st[3], key_cells[1])
    st = rows.permutation(st)
    msg_cells = [lay.assign_value(m) for m in message]
    st = [st[0]] + [lay.add(st[1 + i], msg_cells[i]) if i < len(msg_cells) else st[1 + i] for i in range(t - 1)]
    cipher = st[1:1 + len(msg_cells)]
    st = rows.permutation(st)
    cipher.append(st[1])
    for c, ex in zip(cipher, expected):
        lay.assert_equal(c, ex)
    return cipher


def pose_enc_witness(p: int, k: int, key: Sequence[int], message: Sequence[int], t: int = 5, r_f: int = 8, r_p: int = 57):
    """benches/pose_enc.rs's PoseidonEncCircuit (src/encryption/chip.rs:114-204): MainGate only, the cipher region alone --
    BASELINE configs[0], K = 11."""
    lay = Layouter(p)
    cipher = cipher_region(lay, PoseidonSpec(p, t, r_f, r_p), key, message)
    return _finish(lay, k, WitnessInfo(0, lay.rows, 0, [c.val for c in cipher]), range_lookups=False)


def delay_enc_witness(p: int, k: int, n_big: int, e: int, x: int, exp_bits: int, message: Sequence[int], t: int = 5, rate: int = 4, r_f: int = 8, r_p: int = 57):
    """DelayEncryptCircuit::synthesize (src/lib.rs:164-318): RSA time-lock -> Poseidon hash of the packed result -> the two
    hash outputs key a Poseidon cipher over `message`."""
    lay = Layouter(p)
    rsa_out, want = rsa_region(lay, n_big, e, x, exp_bits)
    rsa_rows = lay.rows
    spec = PoseidonSpec(p, t, r_f, r_p)
    rows = PoseidonRows(lay, spec)
    # hash region: limbs packed three to a field element (src/lib.rs:222-249), sponge with RATE 4 (src/hash/chip.rs:63-85)
    base1 = lay.assign_constant(1 << LIMB_WIDTH)
    base2 = lay.mul(base1, base1)
    inputs = []
    for i in range(len(rsa_out) // 3):
        a = lay.mul_add(rsa_out[3 * i + 1], base1, rsa_out[3 * i])
        inputs.append(lay.mul_add(rsa_out[3 * i + 2], base2, a))
    if len(rsa_out) % 3 == 2:
        inputs.append(lay.mul_add(rsa_out[-1], base1, rsa_out[-2]))
    state = [lay.assign_constant(v) for v in [1 << 64] + [0] * (t - 1)]           # Poseidon::new: capacity word 2^64
    for c0 in range(0, len(inputs), rate):
        chunk = inputs[c0:c0 + rate]
        nxt = [state[0]] + [lay.add(state[1 + i], chunk[i]) if i < len(chunk) else state[1 + i] for i in range(rate)]
        if len(chunk) < rate:                                                      # padding: +1 after the last input
            nxt[1 + len(chunk)] = lay.add_constant(nxt[1 + len(chunk)], 1)
        state = rows.permutation(nxt)
    if len(inputs) % rate == 0:
        state = rows.permutation([state[0], lay.add_constant(state[1], 1)] + state[2:])
    key = [state[1], state[2]]
    cipher = cipher_region(lay, spec, [c.val for c in key], message, key)
    return _finish(lay, k, WitnessInfo(rsa_rows, lay.rows, want, [c.val for c in cipher]))


class NativeCipher:
    """PoseidonCipher::{initial_state, encrypt} (src/encryption/poseidon_enc.rs:66-133) for MESSAGE_CAPACITY = 2, T = 5."""

    def __init__(self, spec: PoseidonSpec, key: Sequence[int]):
        self.spec, self.key = spec, list(key)

    def initial_state(self, nonce: int) -> List[int]:
        return [0, 0, self.key[0], self.key[1], nonce]            # the checked-in state (domain / length words commented out upstream)

    def encrypt(self, message: List[int], nonce: int) -> List[int]:
        p = self.spec.p
        st = self.spec.permute(self.initial_state(nonce))
        cipher = []
        for i, m in enumerate(message):
            st[1 + i] = (st[1 + i] + m) % p
            cipher.append(st[1 + i])
        st = self.spec.permute(st)
        cipher.append(st[1])
        return cipher


def check_rows(circ: SyntheticCircuit, p: int) -> int:
    """Every used row satisfies the gate and (for selected rows) its lookups; returns the number of rows checked.  Python ints."""
    from .keygen import array_to_ints

    fx = [array_to_ints(circ.fixed[i]) for i in range(circ.fixed.shape[0])]
    ad = [array_to_ints(circ.advice[i]) for i in range(5)]
    table = set(plonk.range_table())
    P = plonk
    for r in range(circ.used_rows):
        a, b, c, d, e = (ad[i][r] for i in range(5))
        g = (a * fx[P.MG_SA][r] + b * fx[P.MG_SB][r] + c * fx[P.MG_SC][r] + d * fx[P.MG_SD][r] + e * fx[P.MG_SE][r] + a * b * fx[P.MG_MUL_AB][r] +
             c * d * fx[P.MG_MUL_CD][r] + ad[4][r + 1] * fx[P.MG_NEXT][r] + fx[P.MG_CONST][r]) % p
        if g:
            raise AssertionError("gate not satisfied at row %d" % r)
        if len(fx) <= P.RC_S_OVERFLOW:
            continue                                          # MainGate-only circuit: no lookups
        if fx[P.RC_S_COMPOSITION][r]:
            for v in (a, b, c, d):
                if (fx[P.RC_TAG_COMPOSITION][r], v) not in table:
                    raise AssertionError("composition lookup fails at row %d" % r)
        if fx[P.RC_S_OVERFLOW][r] and (fx[P.RC_TAG_OVERFLOW][r], a) not in table:
            raise AssertionError("overflow lookup fails at row %d" % r)
    m = circ.assembly.mapping
    n = 1 << circ.k
    flat = [v for col in ad for v in col] + [0] * n           # the instance column is empty
    for cell in range(5 * n):
        if m[cell] != cell and flat[cell] != flat[m[cell]]:
            raise AssertionError("copy constraint between different values at cell %d" % cell)
    return circ.used_rows
