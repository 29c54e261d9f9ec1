"""What the proving calls share on the Python side: the two scalar sources a proof draws its blinding from (`OsRng`, `SeededRng` -- the PCG64 stream the
C++ side reproduces, include/dehalo.h `dehalo_rng`) and the layout of a proof's bytes (`proof_layout`, `proof_commitments`: what a batch prover all-gathers,
SURVEY.md 8(e)).

`create_proof` itself -- halo2_proofs::plonk::create_proof over KZG / GWC [UPSTREAM halo2_proofs/src/plonk/prover.rs @ v2023_04_20], the call the reference
times at benches/delay_enc.rs:123-131 -- is ONE library call: `dehalo_create_proof` (csrc/prover.hip), bound in native.py.  Until round 4 this file also
held a Python driver of the same proof over the fine-grained `*_device` entry points; it is test infrastructure now (tests/fine_grained_prover.py: the
executable proof that those ~45 entry points compose into upstream's proof, byte for byte), not a second prover to maintain.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import plonk


class OsRng:
    """The default source of the prover's random scalars (blinding rows, blinds, the vanishing argument's random polynomial):
    operating-system entropy, as the reference passes (`OsRng`, benches/delay_enc.rs:128).  Elements are uniform over the whole
    field (256 random bits masked to the modulus' bit length, rejected when >= p); the value is handed over as a Montgomery
    representation -- multiplication by R is a bijection of the field, so that is uniform too."""

    def __init__(self, p: int):
        self.p = p
        self._limbs = np.array([(p >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
        self._mask = np.uint64((1 << (p.bit_length() - 192)) - 1)

    def scalars(self, count: int) -> np.ndarray:
        import os
        out = np.empty((count, 4), dtype=np.uint64)
        have = 0
        while have < count:
            m = max(16, int(1.6 * (count - have)))
            a = np.frombuffer(os.urandom(32 * m), dtype=np.uint64).reshape(m, 4).copy()
            a[:, 3] &= self._mask
            lt, eq = np.zeros(m, dtype=bool), np.ones(m, dtype=bool)
            for i in (3, 2, 1, 0):
                lt |= eq & (a[:, i] < self._limbs[i])
                eq &= a[:, i] == self._limbs[i]
            good = a[lt][:count - have]
            out[have:have + len(good)] = good
            have += len(good)
        return out

    def fork(self, skip: int) -> "OsRng":
        return OsRng(self.p)                                   # fresh entropy has no position

    def skip(self, count: int):
        pass


class SeededRng:
    """TESTS AND BENCHMARKS ONLY -- not a CSPRNG (PCG64), and with a known seed the blinding rows and the random polynomial are
    predictable, so a proof made with it leaks the witness.  It exists so that a proof is reproducible and can be compared byte for
    byte with the CPU restatement's: upstream draws its scalars from the caller's RngCore one by one, in program order, and this
    stream is consumed in that order.  Elements are raw 253-bit values taken as Montgomery representations (every 253-bit integer is
    below the four moduli)."""

    def __init__(self, seed: int):
        self.gen = np.random.Generator(np.random.PCG64(seed))

    def scalars(self, count: int) -> np.ndarray:
        a = self.gen.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)      # one 64-bit output of the generator per word
        a[:, 3] &= np.uint64((1 << 61) - 1)
        return a

    def fork(self, skip: int) -> "SeededRng":
        """A generator positioned `skip` scalars ahead of this one (which is left where it is): lets a helper thread produce a
        later, large draw -- the vanishing argument's random polynomial -- while the proof's earlier phases run."""
        bg = np.random.PCG64()
        bg.state = self.gen.bit_generator.state
        bg.advance(4 * skip)
        out = SeededRng.__new__(SeededRng)
        out.gen = np.random.Generator(bg)
        return out

    def skip(self, count: int):
        self.gen.bit_generator.advance(4 * count)


def proof_layout(cs: plonk.ConstraintSystem) -> Tuple[int, int]:
    """(commitments written before the evaluations, evaluations) of a proof of this constraint system; the opening quotients
    (one per distinct opening point) follow the evaluations."""
    L, S = len(cs.lookups), cs.num_permutation_sets()
    head = cs.num_advice + 2 * L + S + L + 1 + (cs.degree() - 1)
    evals = len(cs.advice_queries) + len(cs.fixed_queries) + 1 + len(cs.permutation_columns) + max(0, 3 * S - 1) + 5 * L
    return head, evals


def proof_commitments(cs: plonk.ConstraintSystem, proof: bytes) -> bytes:
    """The proof's commitments (32-byte compressed points): what a batch prover all-gathers."""
    head, evals = proof_layout(cs)
    return proof[:32 * head] + proof[32 * (head + evals):]
