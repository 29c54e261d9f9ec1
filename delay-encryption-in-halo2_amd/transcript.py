"""The proof transcript: a mirror of halo2_proofs::transcript::{Blake2bWrite, Blake2bRead, Challenge255}
[UPSTREAM halo2_proofs/src/transcript.rs @ v2023_04_20], which the reference instantiates at
benches/delay_enc.rs:120 (`Blake2bWrite::<_, _, Challenge255<_>>::init(vec![])`) and reads back at :151.

Wire format: a point is written as its 32-byte compressed encoding (x little-endian, bit 7
of the last byte = y is odd; the identity is all zeros -- halo2curves' GroupEncoding for bn256::G1Affine and
pasta_curves' for EpAffine), a scalar as its 32-byte little-endian canonical repr.  The hash absorbs, for a
point, the prefix byte 1 and then x and y as two 32-byte canonical reprs (uncompressed); for a scalar the prefix 2
and its repr; a challenge is Blake2b-512("Halo2-Transcript" personalisation) over everything absorbed so far plus
the prefix byte 0, read as a 512-bit little-endian integer and reduced modulo the scalar field.

Host-side only (hashlib); the prover hands it affine points that the device produced.
"""
from __future__ import annotations

import hashlib
from typing import Optional, Tuple

from .fields import CurveSpec

PREFIX_CHALLENGE, PREFIX_POINT, PREFIX_SCALAR = b"\x00", b"\x01", b"\x02"
PERSONAL = b"Halo2-Transcript"

Affine = Optional[Tuple[int, int]]     # canonical coordinates; None = identity


def compress(curve: CurveSpec, P: Affine) -> bytes:
    if P is None:
        return bytes(32)
    b = bytearray(P[0].to_bytes(32, "little"))
    b[31] |= (P[1] & 1) << 7
    return bytes(b)


def sqrt_mod(a: int, p: int) -> Optional[int]:
    """Tonelli-Shanks (p odd prime); None when a is a non-residue."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


def decompress(curve: CurveSpec, data: bytes) -> Affine:
    """GroupEncoding::from_bytes; raises ValueError for an encoding that is not a curve point."""
    if len(data) != 32:
        raise ValueError("compressed point must be 32 bytes")
    sign = data[31] >> 7
    x = int.from_bytes(data[:31] + bytes([data[31] & 0x7F]), "little")
    p = curve.base.p
    if x >= p:
        raise ValueError("x coordinate is not canonical")
    if x == 0 and sign == 0:
        return None
    y = sqrt_mod((x * x % p * x + curve.b) % p, p)
    if y is None:
        raise ValueError("not on the curve")
    if (y & 1) != sign:
        y = p - y
    return (x, y)


class Blake2bWrite:
    """TranscriptWrite + TranscriptWriterBuffer: write_point / write_scalar append to the proof and absorb;
    common_point / common_scalar only absorb; squeeze_challenge_scalar returns a canonical int."""

    def __init__(self, curve: CurveSpec):
        self.curve = curve
        self.state = hashlib.blake2b(digest_size=64, person=PERSONAL)
        self.proof = bytearray()

    def squeeze_challenge_scalar(self) -> int:
        self.state.update(PREFIX_CHALLENGE)
        digest = self.state.copy().digest()
        return int.from_bytes(digest, "little") % self.curve.scalar.p        # Challenge255: from_uniform_bytes of the 64-byte digest

    def common_point(self, P: Affine):
        if P is None:
            raise ValueError("cannot write points at infinity to the transcript")       # upstream: io::Error
        self.state.update(PREFIX_POINT)
        self.state.update(P[0].to_bytes(32, "little"))
        self.state.update(P[1].to_bytes(32, "little"))

    def common_scalar(self, s: int):
        self.state.update(PREFIX_SCALAR)
        self.state.update((s % self.curve.scalar.p).to_bytes(32, "little"))

    def write_point(self, P: Affine):
        self.common_point(P)
        self.proof += compress(self.curve, P)

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += (s % self.curve.scalar.p).to_bytes(32, "little")

    def finalize(self) -> bytes:
        return bytes(self.proof)


class Blake2bRead(Blake2bWrite):
    """TranscriptRead + TranscriptReadBuffer over a proof produced by Blake2bWrite."""

    def __init__(self, curve: CurveSpec, proof: bytes):
        super().__init__(curve)
        self.data, self.pos = bytes(proof), 0

    def _take(self) -> bytes:
        if self.pos + 32 > len(self.data):
            raise ValueError("proof is too short")
        out = self.data[self.pos:self.pos + 32]
        self.pos += 32
        return out

    def read_point(self) -> Affine:
        P = decompress(self.curve, self._take())
        self.common_point(P)
        return P

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(), "little")
        if s >= self.curve.scalar.p:
            raise ValueError("invalid field element encoding in proof")
        self.common_scalar(s)
        return s
