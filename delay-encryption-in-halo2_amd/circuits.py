"""Synthetic, SATISFIED instances of the reference's two circuit shapes (plonk.maingate_cs): fixed columns, copy
constraints and a witness for which every gate, lookup and copy constraint holds, so that a proof made from them
verifies.  They stand in for the reference's front-end (`DelayEncryptCircuit::synthesize`, src/lib.rs:164-318 --
BigInt / RSA / Poseidon chips over MainGate + RangeChip -- which is out of scope) with the
same column shape and a similar value distribution (SURVEY.md 8(d) "witness-like"): range-decomposition rows
(four sub-limbs of 8 / 4 / 1 bits composed into e), overflow rows (6-bit a), arithmetic rows on 64-bit limbs and
<= 134-bit accumulators, an unused zero tail.  Host-side Python integers; one-time setup, not on the timed path.
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from typing import List

import numpy as np

from . import plonk
from .keygen import ints_to_array


@dataclass
class SyntheticCircuit:
    cs: plonk.ConstraintSystem
    k: int
    fixed: np.ndarray          # (num_fixed, n, 4) u64 canonical
    advice: np.ndarray         # (num_advice, n, 4) u64 canonical (rows >= usable are zero: the prover blinds them)
    assembly: plonk.Assembly
    selectors: List[np.ndarray]
    used_rows: int


TILE_FROM_K = 17


def _tile(small: SyntheticCircuit, k: int) -> SyntheticCircuit:
    """A 2^k-row circuit made of 2^(k - small.k) copies of `small` stacked row-wise: every block keeps its own table rows,
    copy constraints and zero tail (the gate and the lookups are row-local, the last block's tail covers the blinding rows)."""
    nb, n = 1 << small.k, 1 << k
    reps = n // nb
    fixed, advice = np.tile(small.fixed, (1, reps, 1)), np.tile(small.advice, (1, reps, 1))
    ncol = small.assembly.num_columns
    asm = plonk.Assembly(ncol, n)
    m = small.assembly.mapping.reshape(ncol, nb)
    tc, tr = m // nb, m % nb                                        # target column / row inside the block
    big = np.empty((ncol, reps, nb), dtype=np.int64)
    for b in range(reps):
        big[:, b, :] = tc * n + b * nb + tr
    asm.mapping = big.reshape(-1)
    selectors = [np.tile(s, reps) for s in small.selectors]
    return SyntheticCircuit(small.cs, k, fixed, advice, asm, selectors, small.used_rows * reps)


def synthesize(p: int, k: int, range_lookups: bool, seed: int = 1, fill: float = 0.8, copies_per_row: float = 0.25) -> SyntheticCircuit:
    """p: the scalar field's modulus.  `fill`: fraction of the usable rows that hold circuit rows.  Above 2^17 rows the
    circuit is a stack of 2^17-row blocks (Python-integer synthesis of a million rows would take a minute)."""
    if k > TILE_FROM_K:
        return _tile(synthesize(p, TILE_FROM_K, range_lookups, seed, fill, copies_per_row), k)
    cs = plonk.maingate_cs(range_lookups)
    n = 1 << k
    bf = cs.blinding_factors()
    u = n - (bf + 1)
    rnd = random.Random(seed)
    nf = cs.num_fixed
    fixed = [[0] * n for _ in range(nf)]
    adv = [[0] * n for _ in range(5)]
    table = plonk.range_table() if range_lookups else []
    if range_lookups:
        if len(table) > u:
            raise ValueError("k too small for the range table")
        for r, (tag, v) in enumerate(table):
            fixed[plonk.RC_T_TAG][r], fixed[plonk.RC_T_VALUE][r] = tag, v
    rows = max(1, min(u - 1, int(fill * u)))
    kinds = []
    comp_bits = tuple(sorted(plonk.COMPOSITION_BIT_LENS, reverse=True))      # (8, 4, 1): the order the draws below have always indexed
    for r in range(rows):
        t = rnd.random()
        if range_lookups and t < 0.35:
            bits = comp_bits[rnd.randrange(len(comp_bits))]
            a, b, c, d = (rnd.randrange(1 << bits) for _ in range(4))
            adv[0][r], adv[1][r], adv[2][r], adv[3][r] = a, b, c, d
            adv[4][r] = a + (b << bits) + (c << (2 * bits)) + (d << (3 * bits))
            fixed[plonk.RC_S_COMPOSITION][r], fixed[plonk.RC_TAG_COMPOSITION][r] = 1, plonk.range_tag(bits)
            fixed[plonk.MG_SA][r], fixed[plonk.MG_SB][r], fixed[plonk.MG_SC][r], fixed[plonk.MG_SD][r] = 1, 1 << bits, 1 << (2 * bits), 1 << (3 * bits)
            fixed[plonk.MG_SE][r] = p - 1
            kinds.append("range")
        elif range_lookups and t < 0.40:
            adv[0][r] = rnd.randrange(1 << plonk.OVERFLOW_BIT_LENS[0])
            for c in range(1, 5):
                adv[c][r] = rnd.getrandbits(64)
            fixed[plonk.RC_S_OVERFLOW][r], fixed[plonk.RC_TAG_OVERFLOW][r] = 1, plonk.range_tag(plonk.OVERFLOW_BIT_LENS[0])
            fixed[plonk.MG_SA][r], fixed[plonk.MG_SE][r] = 1, rnd.randrange(1, 1 << 16)
            kinds.append("overflow")
        elif t < 0.90:
            adv[0][r], adv[1][r] = rnd.getrandbits(64), rnd.getrandbits(64)
            adv[2][r], adv[3][r] = rnd.getrandbits(134), rnd.getrandbits(134)
            adv[4][r] = rnd.getrandbits(134) if rnd.random() < 0.9 else rnd.randrange(p)
            fixed[plonk.MG_MUL_AB][r] = 1
            fixed[plonk.MG_SC][r], fixed[plonk.MG_SD][r] = 1, rnd.choice((0, 1, p - 1))
            fixed[plonk.MG_SE][r] = p - 1
            if rnd.random() < 0.3:
                fixed[plonk.MG_NEXT][r] = rnd.choice((1, p - 1, 1 << 64))
            if rnd.random() < 0.1:
                fixed[plonk.MG_MUL_CD][r] = 1
            kinds.append("arith")
        else:
            kinds.append("zero")
    # copy constraints: a source cell in a row that is never rewritten, a destination cell in an arithmetic row
    asm = plonk.Assembly(len(cs.permutation_columns), n)
    src_rows = [r for r, t in enumerate(kinds) if t != "arith"]
    dst_rows = [r for r, t in enumerate(kinds) if t == "arith"]
    if not src_rows:                                              # MainGate-only shape: even arithmetic rows feed odd ones
        src_rows, dst_rows = dst_rows[0::2], dst_rows[1::2]
    taken = set()
    for _ in range(int(copies_per_row * rows)) if src_rows and dst_rows else []:
        sc, sr = rnd.randrange(5), src_rows[rnd.randrange(len(src_rows))]
        dc, dr = rnd.randrange(5), dst_rows[rnd.randrange(len(dst_rows))]
        if (dc, dr) in taken:
            continue
        taken.add((dc, dr))
        adv[dc][dr] = adv[sc][sr]
        asm.copy(sc, sr, dc, dr)
    # close every row: s_constant = -(the rest of the gate)
    sa, sb, sc_, sd, se, mab, mcd, nxt = (fixed[i] for i in range(8))
    const = fixed[plonk.MG_CONST]
    a, b, c, d, e = adv
    for r in range(rows):
        acc = a[r] * sa[r] + b[r] * sb[r] + c[r] * sc_[r] + d[r] * sd[r] + e[r] * se[r] + a[r] * b[r] * mab[r] + c[r] * d[r] * mcd[r] + e[r + 1] * nxt[r]
        const[r] = -acc % p
    fixed_arr = np.stack([ints_to_array(col) for col in fixed])
    adv_arr = np.stack([ints_to_array(col) for col in adv])
    selectors = [np.array(fixed[plonk.RC_S_COMPOSITION], dtype=bool), np.array(fixed[plonk.RC_S_OVERFLOW], dtype=bool)] if range_lookups else []
    return SyntheticCircuit(cs, k, fixed_arr, adv_arr, asm, selectors, rows)
