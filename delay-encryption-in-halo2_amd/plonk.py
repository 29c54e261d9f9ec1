"""The circuit description `create_proof` is driven by: a host-side mirror of
halo2_proofs::plonk::{Expression, ConstraintSystem} [UPSTREAM halo2_proofs/src/plonk/circuit.rs @ v2023_04_20]
reduced to what the prover reads -- column counts, gate polynomials, lookup arguments, the permutation's
columns, the query lists (their order is the order of the evaluations in the proof) -- plus the two circuit
shapes the reference proves:

  * `maingate_cs(range_lookups=True)`   delay_enc / mod_pow: one MainGate (5 advice, 9 fixed, 1 instance) and one
    RangeChip (2 table columns, 2 tag columns, 2 complex selectors; 5 lookups on a, b, c, d and a) -- degree 5,
    extended domain 4n, 2 permutation sets (reference configure: src/lib.rs:137-162);
  * `maingate_cs(range_lookups=False)`  pose_enc: MainGate only -- degree 3, extended domain 2n, 6 permutation sets
    (benches/pose_enc.rs:79-85 -> src/encryption/chip.rs:114-204).

The circuits themselves (chips, synthesis) are the reference's front-end and out of scope; only their shape
reaches the hot path.  Field values at this level are canonical Python ints.
"""
from __future__ import annotations

from dataclasses import dataclass, field as dc_field
from typing import List, Sequence, Tuple

from . import evaluation as ev

# Expression nodes (nested tuples): ("const", c) ("fixed", i, rot) ("advice", i, rot) ("instance", i, rot)
#                                   ("neg", e) ("sum", a, b) ("product", a, b) ("scaled", e, c)
Expr = tuple
ADVICE, FIXED, INSTANCE = "advice", "fixed", "instance"


def degree(e: Expr) -> int:
    """Expression::degree."""
    k = e[0]
    if k == "const":
        return 0
    if k in (ADVICE, FIXED, INSTANCE):
        return 1
    if k == "neg" or k == "scaled":
        return degree(e[1])
    if k == "sum":
        return max(degree(e[1]), degree(e[2]))
    if k == "product":
        return degree(e[1]) + degree(e[2])
    raise ValueError("unknown expression node %r" % (k,))


def sum_(*terms: Expr) -> Expr:
    acc = terms[0]
    for t in terms[1:]:
        acc = ("sum", acc, t)
    return acc


def mul(a: Expr, b: Expr) -> Expr:
    return ("product", a, b)


@dataclass
class ConstraintSystem:
    """The fields of upstream's ConstraintSystem that the prover and verifier read."""
    num_advice: int = 0
    num_fixed: int = 0
    num_instance: int = 0
    gates: List[Expr] = dc_field(default_factory=list)                           # every gate polynomial, in order
    lookups: List[Tuple[List[Expr], List[Expr]]] = dc_field(default_factory=list)   # (input_expressions, table_expressions)
    permutation_columns: List[Tuple[str, int]] = dc_field(default_factory=list)  # (kind, index), in enable_equality order
    advice_queries: List[Tuple[int, int]] = dc_field(default_factory=list)       # (column, rotation), first-use order
    fixed_queries: List[Tuple[int, int]] = dc_field(default_factory=list)
    instance_queries: List[Tuple[int, int]] = dc_field(default_factory=list)
    minimum_degree: int = 0

    # ---- building (ConstraintSystem::{query_*_index, enable_equality, create_gate, lookup}) ----
    def _query(self, kind: str, index: int, rot: int) -> Expr:
        lst = {ADVICE: self.advice_queries, FIXED: self.fixed_queries, INSTANCE: self.instance_queries}[kind]
        if (index, rot) not in lst:
            lst.append((index, rot))
        return (kind, index, rot)

    def query_advice(self, index: int, rot: int = 0) -> Expr:
        return self._query(ADVICE, index, rot)

    def query_fixed(self, index: int, rot: int = 0) -> Expr:
        return self._query(FIXED, index, rot)

    def query_instance(self, index: int, rot: int = 0) -> Expr:
        return self._query(INSTANCE, index, rot)

    def enable_equality(self, kind: str, index: int):
        self._query(kind, index, 0)
        if (kind, index) not in self.permutation_columns:
            self.permutation_columns.append((kind, index))

    def create_gate(self, polys: Sequence[Expr]):
        self.gates.extend(polys)

    def lookup(self, pairs: Sequence[Tuple[Expr, Expr]]):
        self.lookups.append(([p[0] for p in pairs], [p[1] for p in pairs]))

    def description(self) -> tuple:
        """The constraint system as plain data (what a verifier needs besides the key's commitments); its text is hashed into
        the verifying key's transcript representation."""
        return (self.num_advice, self.num_fixed, self.num_instance, tuple(self.gates), tuple((tuple(i), tuple(t)) for i, t in self.lookups),
                tuple(self.permutation_columns), tuple(self.advice_queries), tuple(self.fixed_queries), tuple(self.instance_queries), self.minimum_degree)

    # ---- derived quantities ----
    def num_advice_queries(self) -> List[int]:
        cnt = [0] * self.num_advice
        for c, _ in self.advice_queries:
            cnt[c] += 1
        return cnt

    def blinding_factors(self) -> int:
        """ConstraintSystem::blinding_factors: max(3, most queries to one advice column) + 2."""
        factors = max(self.num_advice_queries() or [1])
        return max(3, factors) + 2

    def degree(self) -> int:
        """ConstraintSystem::degree: the permutation argument needs 3, a lookup max(4, 2 + input + table), gates their own."""
        d = 3                                                 # permutation::Argument::required_degree() == 3, unconditionally
        for inputs, tables in self.lookups:
            di = max([1] + [degree(e) for e in inputs])
            dt = max([1] + [degree(e) for e in tables])
            d = max(d, max(4, 2 + di + dt))
        for g in self.gates:
            d = max(d, degree(g))
        return max(d, self.minimum_degree)

    def permutation_chunk_len(self) -> int:
        return self.degree() - 2

    def num_permutation_sets(self) -> int:
        c = self.permutation_chunk_len()
        return (len(self.permutation_columns) + c - 1) // c


# ---- Expression -> GraphEvaluator (plonk/evaluation.rs GraphEvaluator::add_expression) ----
_KIND = {FIXED: ev.FIXED, ADVICE: ev.ADVICE, INSTANCE: ev.INSTANCE}
_ZERO, _ONE, _TWO = (ev.CONSTANT, 0, 0), (ev.CONSTANT, 1, 0), (ev.CONSTANT, 2, 0)


def add_expression(g: ev.GraphEvaluator, e: Expr, p: int) -> ev.Source:
    k = e[0]
    if k == "const":
        return g.add_constant(e[1] % p)
    if k in _KIND:
        return g.add_calculation(ev.STORE, g.column(_KIND[k], e[1], e[2]))
    if k == "neg":
        if e[1][0] == "const":
            return g.add_constant(-e[1][1] % p)
        a = add_expression(g, e[1], p)
        return a if a == _ZERO else g.add_calculation(ev.NEGATE, a)
    if k == "sum":
        if e[2][0] == "neg":                                  # a - b
            a, b = add_expression(g, e[1], p), add_expression(g, e[2][1], p)
            if a == _ZERO:
                return g.add_calculation(ev.NEGATE, b)
            return a if b == _ZERO else g.add_calculation(ev.SUB, a, b)
        a, b = add_expression(g, e[1], p), add_expression(g, e[2], p)
        if a == _ZERO:
            return b
        if b == _ZERO:
            return a
        return g.add_calculation(ev.ADD, a, b) if a <= b else g.add_calculation(ev.ADD, b, a)
    if k == "product":
        a, b = add_expression(g, e[1], p), add_expression(g, e[2], p)
        if a == _ZERO or b == _ZERO:
            return _ZERO
        if a == _ONE:
            return b
        if b == _ONE:
            return a
        if a == _TWO:
            return g.add_calculation(ev.DOUBLE, b)
        if b == _TWO:
            return g.add_calculation(ev.DOUBLE, a)
        if a == b:
            return g.add_calculation(ev.SQUARE, a)
        return g.add_calculation(ev.MUL, a, b) if a <= b else g.add_calculation(ev.MUL, b, a)
    if k == "scaled":
        c = e[2] % p
        if c == 0:
            return _ZERO
        if c == 1:
            return add_expression(g, e[1], p)
        cst = g.add_constant(c)
        return g.add_calculation(ev.MUL, add_expression(g, e[1], p), cst)
    raise ValueError("unknown expression node %r" % (k,))


def custom_gates_graph(cs: ConstraintSystem, p: int) -> ev.GraphEvaluator:
    """Evaluator::new, custom gates: value = Horner(previous, gate polynomials, y)."""
    g = ev.GraphEvaluator()
    parts = [add_expression(g, poly, p) for poly in cs.gates]
    g.add_calculation(ev.HORNER, (ev.PREVIOUS, 0, 0), (ev.Y, 0, 0), tuple(parts))
    return g


def lookup_table_value_graph(inputs: Sequence[Expr], tables: Sequence[Expr], p: int) -> ev.GraphEvaluator:
    """Evaluator::new, one lookup: (theta-compressed input + beta) * (theta-compressed table + gamma)."""
    g = ev.GraphEvaluator()
    ci = g.add_calculation(ev.HORNER, _ZERO, (ev.THETA, 0, 0), tuple(add_expression(g, e, p) for e in inputs))
    ct = g.add_calculation(ev.HORNER, _ZERO, (ev.THETA, 0, 0), tuple(add_expression(g, e, p) for e in tables))
    right = g.add_calculation(ev.ADD, ct, (ev.GAMMA, 0, 0))
    left = g.add_calculation(ev.ADD, ci, (ev.BETA, 0, 0))
    g.add_calculation(ev.MUL, left, right)
    return g


def compress_graph(exprs: Sequence[Expr], p: int) -> ev.GraphEvaluator:
    """lookup::Argument::commit_permuted's compress_expressions: fold(acc * theta + expression) over the rows of the
    ORIGINAL domain (rot_scale 1)."""
    g = ev.GraphEvaluator()
    g.add_calculation(ev.HORNER, _ZERO, (ev.THETA, 0, 0), tuple(add_expression(g, e, p) for e in exprs))
    return g


# ---- the reference's two circuit shapes -----------------------------------------------------------------------
# MainGate fixed columns (halo2wrong maingate [UPSTREAM]): sa sb sc sd se, s_mul_ab, s_mul_cd, s_next (e at the next
# row), s_constant; RangeChip: t_tag, t_value (table), tag_composition, tag_overflow, s_composition, s_overflow.
MG_SA, MG_SB, MG_SC, MG_SD, MG_SE, MG_MUL_AB, MG_MUL_CD, MG_NEXT, MG_CONST = range(9)
RC_T_TAG, RC_T_VALUE, RC_TAG_COMPOSITION, RC_TAG_OVERFLOW, RC_S_COMPOSITION, RC_S_OVERFLOW = range(9, 15)
# RangeChip::configure(composition_bit_lens, overflow_bit_lens) of src/lib.rs:144-149 with compute_range_lens
# (src/big_integer/chip.rs:1224-1253, src/rsa/chip.rs:252-257): distinct table bit lengths and their tags
# -- (8, 1, 8, 4) and (0, 0, 6); RangeChip::configure [UPSTREAM maingate/src/range.rs] sorts, dedups and drops the zeros, and numbers
# the union of both lists in ascending bit length (a BTreeMap): 1 -> tag 1, 4 -> 2, 6 -> 3, 8 -> 4
COMPOSITION_BIT_LENS, OVERFLOW_BIT_LENS = (1, 4, 8), (6,)
RANGE_BIT_LENS = tuple(sorted(set(COMPOSITION_BIT_LENS + OVERFLOW_BIT_LENS)))


def range_bit_lens(num_limbs: int = 32, limb_width: int = 64) -> Tuple[int, ...]:
    """The distinct table bit lengths of RSAChip::compute_range_lens(num_limbs) in tag order.  Only the carries' overflow length depends on the
    modulus: 2 x (num_limbs (2^64 - 1)^2 + 2^64 - 1) has 134 bits for the reference's 32 limbs (70-bit carries: eight 8-bit limbs + 6 bits), 133 for 16
    limbs (5 bits), ..."""
    word_max = num_limbs * ((1 << limb_width) - 1) ** 2 + (1 << limb_width) - 1
    carry_bits = (2 * word_max).bit_length() - limb_width
    comp = max(1, carry_bits // 8)
    return tuple(sorted({1, 4, 8, comp} | ({carry_bits % comp} - {0})))


def maingate_cs(range_lookups: bool = True) -> ConstraintSystem:
    cs = ConstraintSystem(num_advice=5, num_fixed=15 if range_lookups else 9, num_instance=1)
    for i in range(5):                                        # MainGate::configure: equality on a..e and the instance column
        cs.enable_equality(ADVICE, i)
    cs.enable_equality(INSTANCE, 0)
    a, b, c, d, e = (cs.query_advice(i) for i in range(5))
    e_next = cs.query_advice(4, 1)
    f = [cs.query_fixed(i) for i in range(9)]
    cs.create_gate([sum_(mul(a, f[MG_SA]), mul(b, f[MG_SB]), mul(c, f[MG_SC]), mul(d, f[MG_SD]), mul(e, f[MG_SE]),
                         mul(mul(a, b), f[MG_MUL_AB]), mul(mul(c, d), f[MG_MUL_CD]), mul(e_next, f[MG_NEXT]), f[MG_CONST])])
    if range_lookups:                                         # RangeChip::configure_lookup_with_column_tag
        for col in range(4):
            s, tag, v = cs.query_fixed(RC_S_COMPOSITION), cs.query_fixed(RC_TAG_COMPOSITION), cs.query_advice(col)
            cs.lookup([(tag, cs.query_fixed(RC_T_TAG)), (mul(s, v), cs.query_fixed(RC_T_VALUE))])
        s, tag, v = cs.query_fixed(RC_S_OVERFLOW), cs.query_fixed(RC_TAG_OVERFLOW), cs.query_advice(0)
        cs.lookup([(tag, cs.query_fixed(RC_T_TAG)), (mul(s, v), cs.query_fixed(RC_T_VALUE))])
    return cs


def range_table(num_limbs: int = 32) -> List[Tuple[int, int]]:
    """RangeChip::load_table rows (tag, value): the disabled row (0, 0), then every value of every bit length, shortest first."""
    rows = [(0, 0)]
    for tag, bits in enumerate(range_bit_lens(num_limbs), start=1):
        rows += [(tag, v) for v in range(1 << bits)]
    return rows


def range_tag(bits: int, num_limbs: int = 32) -> int:
    return 1 + range_bit_lens(num_limbs).index(bits)


# ---- permutation::keygen::Assembly ----------------------------------------------------------------------------
class Assembly:
    """Copy constraints as cycles over cells [UPSTREAM plonk/permutation/keygen.rs Assembly]; a cell is the flat
    index column * n + row."""

    def __init__(self, num_columns: int, n: int):
        import numpy as np

        self.n, self.num_columns = n, num_columns
        self.mapping = np.arange(num_columns * n, dtype=np.int64)
        self.aux = np.arange(num_columns * n, dtype=np.int64)
        self.sizes = np.ones(num_columns * n, dtype=np.int64)

    def copy(self, left_column: int, left_row: int, right_column: int, right_row: int):
        left, right = left_column * self.n + left_row, right_column * self.n + right_row
        m, aux = self.mapping, self.aux
        if aux[left] == aux[right]:
            return
        if self.sizes[aux[left]] < self.sizes[aux[right]]:
            left, right = right, left
        la = aux[left]
        self.sizes[la] += self.sizes[aux[right]]
        i = right
        while True:
            aux[i] = la
            i = m[i]
            if i == right:
                break
        m[left], m[right] = m[right], m[left]
