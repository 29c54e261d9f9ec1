"""Field / curve identities and the O(log n) setup constants of EvaluationDomain::new
(halo2_proofs/src/poly/domain.rs).  Host logic only: a handful
of modular exponentiations with Python integers per domain, never per element."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

MASK64 = (1 << 64) - 1


@dataclass(frozen=True)
class FieldSpec:
    name: str
    id: int          # dehalo_field
    p: int
    gen: int         # multiplicative generator (halo2curves MULTIPLICATIVE_GENERATOR)
    zeta: int        # F::ZETA, the literal constant of halo2curves / pasta_curves [UPSTREAM; SURVEY.md A.3]: a primitive cube root of unity

    @property
    def two_adicity(self) -> int:
        t, s = self.p - 1, 0
        while t % 2 == 0:
            t //= 2
            s += 1
        return s

    @property
    def root_of_unity(self) -> int:
        return pow(self.gen, (self.p - 1) >> self.two_adicity, self.p)

    # halo2curves in-memory form: 4 x u64 LE limbs of a * 2^256 mod p
    def encode(self, a: int) -> np.ndarray:
        return np.frombuffer(((a % self.p) * (1 << 256) % self.p).to_bytes(32, "little"), dtype=np.uint64).copy()

    def decode(self, limbs) -> int:
        m = sum(int(x) << (64 * i) for i, x in enumerate(limbs))
        return m * pow(1 << 256, -1, self.p) % self.p

    def encode_many(self, vals) -> np.ndarray:
        p = self.p
        return np.frombuffer(b"".join(((v % p) * (1 << 256) % p).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()

    def decode_many(self, arr) -> list:
        return [self.decode(r) for r in np.asarray(arr).reshape(-1, 4)]


BN254_FR = FieldSpec("bn254_fr", 0, 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 7,
                     0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD)
BN254_FQ = FieldSpec("bn254_fq", 1, 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47, 3,
                     0x30644E72E131A0295E6DD9E7E0ACCCB0C28F069FBB966E3DE4BD44E5607CFD48)
PASTA_FP = FieldSpec("pasta_fp", 2, 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001, 5,
                     0x12CCCA834ACDBA712CAAD5DC57AAB1B01D1F8BD237AD31491DAD5EBDFDFE4AB9)
PASTA_FQ = FieldSpec("pasta_fq", 3, 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001, 5,
                     0x06819A58283E528E511DB4D81CF70F5A0FED467D47C033AF2AA9D2E050AA0E4F)
FIELDS = {f.name: f for f in (BN254_FR, BN254_FQ, PASTA_FP, PASTA_FQ)}


@dataclass(frozen=True)
class CurveSpec:
    name: str
    id: int          # dehalo_curve
    base: FieldSpec
    scalar: FieldSpec
    b: int


BN254 = CurveSpec("bn254", 0, BN254_FQ, BN254_FR, 3)
PALLAS = CurveSpec("pallas", 1, PASTA_FP, PASTA_FQ, 5)
VESTA = CurveSpec("vesta", 2, PASTA_FQ, PASTA_FP, 5)
CURVES = {c.name: c for c in (BN254, PALLAS, VESTA)}
