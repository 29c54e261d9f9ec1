"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's checker leg may import this).

A MockProver for the reference's circuits, on the checker's side: given the constraint-system description the checker wrote down itself
(oracle/shapes.py), the fixed columns, the advice columns and the permutation mapping a synthesis produced, check what [UPSTREAM
halo2_proofs/src/dev.rs MockProver::verify] checks -- every gate polynomial vanishes on every usable row, every lookup's input tuple is a row
of its table, every copy constraint joins equal values -- without calling anything of the product (the product's own `witness.check_rows` is not
used by the tests that matter).  It also restates, from the reference's sources alone, what the RangeChip's table must contain
(src/big_integer/chip.rs:1224-1253 compute_range_lens + src/rsa/chip.rs:252-257: composition lengths [8, 1, 8, 4], overflow lengths [0, 0, 6])
and the row counts the reference publishes (benches/README.md:56-99), as data.

Python integers; columns come in as (n, 4) u64 canonical arrays.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

import shapes

BLINDING_FACTORS = 5          # max(3, queries of one advice column = 2: e and e(next)) + 2 [UPSTREAM ConstraintSystem::blinding_factors]


def ints(arr) -> List[int]:
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in a]


def expected_range_table(num_limbs: int = 32) -> List[Tuple[int, int]]:
    """RangeChip::load_table for the reference's configuration (num_limbs = BITS_LEN / 64 = 32; 16 for a 1024-bit modulus): the row (0, 0), then for each
    distinct non-zero bit length in ascending order (tag = 1 + its rank) every value below 2^bits."""
    limb_width, num_lookup_limbs = 64, 8
    comp = [limb_width // num_lookup_limbs]
    over = [limb_width % comp[0]]
    fresh_carry_bits = (2 * (1 << limb_width)).bit_length() - limb_width
    comp.append(max(1, fresh_carry_bits // num_lookup_limbs))
    over.append(fresh_carry_bits % comp[-1])
    word_max = num_limbs * ((1 << limb_width) - 1) ** 2 + (1 << limb_width) - 1
    mul_carry_bits = (2 * word_max).bit_length() - limb_width
    comp.append(max(1, mul_carry_bits // num_lookup_limbs))
    over.append(mul_carry_bits % comp[-1])
    comp.append(32 // num_lookup_limbs)                       # RSAChip::compute_range_lens
    lens = sorted(set(b for b in comp + over if b))
    assert num_limbs != 32 or lens == [1, 4, 6, 8]
    rows = [(0, 0)]
    for tag, bits in enumerate(lens, start=1):
        rows += [(tag, v) for v in range(1 << bits)]
    return rows


def _evaluate(expr, cols: Dict[str, List[List[int]]], row: int, n: int, p: int) -> int:
    kind = expr[0]
    if kind in ("advice", "fixed", "instance"):
        return cols[kind][expr[1]][(row + expr[2]) % n]
    if kind == "sum":
        return (_evaluate(expr[1], cols, row, n, p) + _evaluate(expr[2], cols, row, n, p)) % p
    if kind == "product":
        return _evaluate(expr[1], cols, row, n, p) * _evaluate(expr[2], cols, row, n, p) % p
    if kind == "const":
        return expr[1] % p
    if kind == "neg":
        return -_evaluate(expr[1], cols, row, n, p) % p
    if kind == "scaled":
        return _evaluate(expr[1], cols, row, n, p) * expr[2] % p
    raise ValueError("unknown expression node %r" % (kind,))


def verify(desc: tuple, k: int, p: int, fixed, advice, mapping, used_rows: int = None) -> Dict[str, int]:
    """Raises AssertionError naming the first failure; returns counts of what was checked."""
    num_advice, num_fixed, num_instance, gates, lookups, perm_cols = desc[:6]
    n = 1 << k
    usable = n - (BLINDING_FACTORS + 1)
    cols = {"fixed": [ints(fixed[i]) for i in range(num_fixed)], "advice": [ints(advice[i]) for i in range(num_advice)],
            "instance": [[0] * n for _ in range(num_instance)]}
    assert all(len(c) == n for kind in cols.values() for c in kind), "a column is not 2^k rows long"
    if used_rows is not None:
        assert used_rows + 1 <= usable, "more rows than the domain has usable"
        for c in cols["advice"]:
            assert not any(c[used_rows:]), "advice beyond the used rows"
    for g, gate in enumerate(gates):
        for r in range(usable):
            if _evaluate(gate, cols, r, n, p):
                raise AssertionError("gate %d is not satisfied at row %d" % (g, r))
    checked_lookups = 0
    for li, (inputs, table) in enumerate(lookups):
        rows = set(tuple(_evaluate(t, cols, r, n, p) for t in table) for r in range(usable))
        for r in range(usable):
            if tuple(_evaluate(e, cols, r, n, p) for e in inputs) not in rows:
                raise AssertionError("lookup %d: the input of row %d is not in the table" % (li, r))
            checked_lookups += 1
    m = np.asarray(mapping, dtype=np.int64).reshape(-1)
    assert m.size == len(perm_cols) * n
    flat = [v for kind, idx in perm_cols for v in cols[kind][idx]]
    assert sorted(m.tolist()) == list(range(m.size)), "the mapping is not a permutation"
    copies = 0
    for cell in np.nonzero(m != np.arange(m.size))[0].tolist():
        if flat[cell] != flat[int(m[cell])]:
            raise AssertionError("copy constraint between different values at cell %d" % cell)
        copies += 1
    return {"rows": usable, "lookup_inputs": checked_lookups, "cells_in_cycles": copies}


def verify_range_table(fixed, k: int, num_limbs: int = 32):
    """The two table columns hold RangeChip::load_table's rows from row 0 on and the default row (0, 0) below them."""
    n = 1 << k
    tag, val = ints(fixed[shapes.T_TAG]), ints(fixed[shapes.T_VALUE])
    want = expected_range_table(num_limbs)
    got = list(zip(tag, val))
    assert got[:len(want)] == want, "the range table differs from RangeChip::load_table's"
    assert not any(t or v for t, v in got[len(want):n]), "table rows beyond the last bit length"


# ---- the row counts the reference publishes (benches/README.md), as data --------------------------------------------------------------
# (k, published advice rows, exponent bits) -- "performance of modulo power circuit", benches/README.md:77-91
README_MOD_POW = [(15, 17822, 2), (15, 25803, 3), (16, 33784, 4), (16, 41766, 5), (16, 49747, 6), (16, 57728, 7), (17, 65709, 8), (17, 121578, 15),
                  (17, 129559, 16), (18, 137541, 17), (18, 249278, 31), (18, 257259, 32), (19, 265241, 33)]
# (k, rows, exponent bits as printed, message words) -- "performance of delay encryption circuit", benches/README.md:56-65
README_DELAY_ENC = [(15, 26461, 2, 2), (16, 34473, 3, 2), (16, 58417, 6, 2), (17, 122267, 7, 2), (17, 130248, 15, 2), (18, 138229, 16, 2), (18, 257948, 31, 2),
                    (19, 265929, 32, 2)]
# (k, rows, message words) -- "performance of poseidon encryption circuit", benches/README.md:97-107
README_POSE_ENC = [(11, 1446, 1), (11, 1450, 2), (11, 1454, 3), (11, 1458, 4), (12, 2180, 5), (12, 2184, 6), (12, 3660, 16), (13, 4394, 17), (13, 4394, 20), (13, 5116, 21),
                   (13, 6592, 31)]
