"""ParamsKZG, VerifyingKey, ProvingKey and keygen: the objects the reference builds once and caches on disk
before timing `create_proof` (benches/delay_enc.rs:41-54 params, :84-115 vk / pk), device-resident, with
upstream's on-disk formats.

  ParamsKZG::{write, read}      [UPSTREAM halo2_proofs/src/poly/kzg/commitment.rs, SerdeFormat::RawBytes]
      k: u32 LE | g[0..n) | g_lagrange[0..n) | g2 | s_g2 ; a G1 point = x, y as 4 x u64 LE Montgomery limbs (64 B),
      a G2 point = x.c0, x.c1, y.c0, y.c1 (128 B).
  VerifyingKey::{write, read}   [UPSTREAM halo2_proofs/src/plonk.rs]
      k: u32 BE | #fixed commitments: u32 BE | commitments (64 B raw each) | permutation commitments |
      selectors packed 8 bools per byte (LSB first), ceil(n / 8) bytes per selector.
  ProvingKey::{write, read}     [UPSTREAM halo2_proofs/src/plonk.rs]
      vk | l0 | l_last | l_active_row (extended polynomials) | fixed_values | fixed_polys | fixed_cosets |
      permutation values | polys | cosets ; a polynomial = len: u32 BE | elements (32 B raw Montgomery);
      a slice of polynomials = count: u32 BE | polynomials.
Reading a key needs the circuit's ConstraintSystem, as upstream's `read::<_, ConcreteCircuit>` does.

keygen_vk / keygen_pk [UPSTREAM halo2_proofs/src/plonk/keygen.rs] run on the device: commit_lagrange of every fixed
and permutation column, lagrange_to_coeff and coeff_to_extended of each, l0 / l_last / l_active_row.
Device arrays are torch tensors (plumbing only): int64 views of n x 4 u64 limbs.
"""
from __future__ import annotations

import hashlib
import io
import struct
from typing import List, Optional, Sequence

import numpy as np

from . import evaluation as ev
from . import plonk
from ._lib import Bases, Context
from .domain import EvaluationDomain
from .fields import CurveSpec, FieldSpec


# ---- bulk conversions --------------------------------------------------------------------------------
def ints_to_array(vals: Sequence[int]) -> np.ndarray:
    """canonical ints -> (len, 4) u64 little-endian limbs (NOT Montgomery)."""
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()


def array_to_ints(arr) -> List[int]:
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def to_device(arr):
    """host array of 64-bit words -> int64 tensor in HBM, through the library's staged upload (never `tensor.cuda()` of a pageable array: _lib.Context.upload)"""
    from ._lib import transfer_context

    with transfer_context() as c:
        return c.upload(np.ascontiguousarray(arr, dtype=np.uint64))


def to_host(t) -> np.ndarray:
    import torch
    from ._lib import transfer_context

    torch.cuda.current_stream().synchronize()      # what `t.cpu()` would have waited for
    with transfer_context() as c:
        return c.download_tensor(t.contiguous())


_RINV: dict = {}


def _rinv(p: int) -> int:
    if p not in _RINV:
        _RINV[p] = pow(1 << 256, -1, p)
    return _RINV[p]


def decode_points(curve: CurveSpec, xy) -> list:
    """(count, 8) u64 Montgomery affine -> [(x, y) | None] canonical."""
    out = []
    f = curve.base
    rinv = _rinv(f.p)
    vals = array_to_ints(np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 4))
    for i in range(0, len(vals), 2):
        x, y = vals[i], vals[i + 1]
        out.append(None if x == 0 and y == 0 else (x * rinv % f.p, y * rinv % f.p))
    return out


def encode_points(curve: CurveSpec, pts) -> np.ndarray:
    f = curve.base
    R = (1 << 256) % f.p
    flat = []
    for P in pts:
        flat += [0, 0] if P is None else [P[0] * R % f.p, P[1] * R % f.p]
    return ints_to_array(flat).reshape(-1, 8)


# ---- ParamsKZG -----------------------------------------------------------------------------------------
class ParamsKZG:
    """ParamsKZG<Bn256> (or its Pasta-curve analogue for the MSM/NTT configs): g, g_lagrange resident on the device."""

    def __init__(self, ctx: Context, curve: CurveSpec, k: int, g, g_lagrange, g2: bytes = b"", s_g2: bytes = b"", window_bits: int = 0):
        self.ctx, self.curve, self.k, self.n = ctx, curve, k, 1 << k
        self.g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
        self.g_lagrange = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
        if self.g.shape[0] != self.n or self.g_lagrange.shape[0] != self.n:
            raise ValueError("g and g_lagrange must hold 2^k points")
        self.g2, self.s_g2 = bytes(g2), bytes(s_g2)
        self.bases_g: Bases = ctx.register_bases(curve.id, self.g, window_bits, True)
        self.bases_g_lagrange: Bases = ctx.register_bases(curve.id, self.g_lagrange, window_bits, True)

    def release(self):
        self.bases_g.release()
        self.bases_g_lagrange.release()

    # commit / commit_lagrange on device-resident columns: `batch` polynomials of `length` <= n coefficients, n apart
    def commit_device(self, d_polys: int, batch: int, d_out: int, lagrange: bool, length: Optional[int] = None, ctx: Optional[Context] = None):
        (ctx or self.ctx).msm_device(self.bases_g_lagrange if lagrange else self.bases_g, d_polys, self.n if length is None else length, batch, d_out, 0)

    def commit_affine_device(self, d_polys: int, batch: int, d_out_affine: int, lagrange: bool, length: Optional[int] = None, ctx: Optional[Context] = None):
        """commit(..).to_affine() of `batch` columns in one call: the points the transcript absorbs."""
        (ctx or self.ctx).msm_device_affine(self.bases_g_lagrange if lagrange else self.bases_g, d_polys, self.n if length is None else length, batch, 0,
                                            d_out_affine, 0)

    def write(self, fh):
        fh.write(struct.pack("<I", self.k))
        fh.write(self.g.tobytes())
        fh.write(self.g_lagrange.tobytes())
        fh.write(self.g2.ljust(128, b"\0"))
        fh.write(self.s_g2.ljust(128, b"\0"))

    @classmethod
    def read(cls, ctx: Context, curve: CurveSpec, fh, window_bits: int = 0) -> "ParamsKZG":
        (k,) = struct.unpack("<I", _exact(fh, 4))
        if k > 28:
            raise ValueError("params: k out of range")
        n = 1 << k
        g = np.frombuffer(_exact(fh, 64 * n), dtype=np.uint64).reshape(n, 8)
        gl = np.frombuffer(_exact(fh, 64 * n), dtype=np.uint64).reshape(n, 8)
        g2, s_g2 = _exact(fh, 128), _exact(fh, 128)
        return cls(ctx, curve, k, g, gl, g2, s_g2, window_bits)


def _exact(fh, nbytes: int) -> bytes:
    b = fh.read(nbytes)
    if len(b) != nbytes:
        raise ValueError("unexpected end of file")
    return b


def _write_poly(fh, arr: np.ndarray):
    fh.write(struct.pack(">I", arr.shape[0]))
    fh.write(np.ascontiguousarray(arr, dtype=np.uint64).tobytes())


def _read_poly(fh, expect_len: int) -> np.ndarray:
    (ln,) = struct.unpack(">I", _exact(fh, 4))
    if ln != expect_len:
        raise ValueError("polynomial length %d, expected %d" % (ln, expect_len))
    return np.frombuffer(_exact(fh, 32 * ln), dtype=np.uint64).reshape(ln, 4)


def _write_poly_slice(fh, arrs):
    fh.write(struct.pack(">I", len(arrs)))
    for a in arrs:
        _write_poly(fh, a)


def _read_poly_slice(fh, expect_count: int, expect_len: int) -> np.ndarray:
    (cnt,) = struct.unpack(">I", _exact(fh, 4))
    if cnt != expect_count:
        raise ValueError("polynomial slice of %d, expected %d" % (cnt, expect_count))
    return np.stack([_read_poly(fh, expect_len) for _ in range(cnt)]) if cnt else np.zeros((0, expect_len, 4), dtype=np.uint64)


# ---- keys -----------------------------------------------------------------------------------------------
class VerifyingKey:
    def __init__(self, curve: CurveSpec, k: int, cs: plonk.ConstraintSystem, fixed_commitments: np.ndarray, permutation_commitments: np.ndarray,
                 selectors: Sequence[np.ndarray] = ()):
        self.curve, self.k, self.cs = curve, k, cs
        self.fixed_commitments = np.ascontiguousarray(fixed_commitments, dtype=np.uint64).reshape(-1, 8)        # Montgomery affine
        self.permutation_commitments = np.ascontiguousarray(permutation_commitments, dtype=np.uint64).reshape(-1, 8)
        self.selectors = [np.asarray(s, dtype=bool) for s in selectors]
        self.transcript_repr = self._transcript_repr()

    def _transcript_repr(self) -> int:
        """vk.transcript_repr.  [UPSTREAM hashes `format!("{:?}", vk.pinned())` with Blake2b-512 personalised
        "Halo2-Verify-Key"; Rust's Debug text of the pinned key cannot be reproduced without the crate, so the same
        hash is taken over this key's RawBytes serialisation and the constraint system's description instead.]"""
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
        buf = io.BytesIO()
        self.write(buf)
        body = buf.getvalue() + repr(self.cs.description()).encode()
        h.update(struct.pack("<Q", len(body)))
        h.update(body)
        return int.from_bytes(h.digest(), "little") % self.curve.scalar.p

    def write(self, fh):
        fh.write(struct.pack(">I", self.k))
        fh.write(struct.pack(">I", self.fixed_commitments.shape[0]))
        fh.write(self.fixed_commitments.tobytes())
        fh.write(self.permutation_commitments.tobytes())
        for sel in self.selectors:
            fh.write(np.packbits(sel, bitorder="little").tobytes())

    @classmethod
    def read(cls, curve: CurveSpec, cs: plonk.ConstraintSystem, fh, num_selectors: int = 0) -> "VerifyingKey":
        (k,) = struct.unpack(">I", _exact(fh, 4))
        (nf,) = struct.unpack(">I", _exact(fh, 4))
        if nf != cs.num_fixed:
            raise ValueError("vk holds %d fixed commitments, the circuit has %d fixed columns" % (nf, cs.num_fixed))
        fixed = np.frombuffer(_exact(fh, 64 * nf), dtype=np.uint64).reshape(nf, 8)
        npc = len(cs.permutation_columns)
        perm = np.frombuffer(_exact(fh, 64 * npc), dtype=np.uint64).reshape(npc, 8)
        n = 1 << k
        sels = [np.unpackbits(np.frombuffer(_exact(fh, (n + 7) // 8), dtype=np.uint8), bitorder="little")[:n].astype(bool) for _ in range(num_selectors)]
        return cls(curve, k, cs, fixed, perm, sels)


class ProvingKey:
    """Device-resident proving key.  Extended-domain columns are kept in the kernels' internal element form
    (dehalo.h: DEHALO_EVAL_COLUMNS_INTERNAL) and converted back only when the key is written."""

    def __init__(self, ctx: Context, vk: VerifyingKey, domain: EvaluationDomain, l_ext, fixed_values, fixed_polys, fixed_cosets, perm_values, perm_polys,
                 perm_cosets):
        self.ctx, self.vk, self.domain = ctx, vk, domain
        self.l_ext = l_ext                        # (3, ext_n, 4): l0, l_last, l_active_row -- internal form
        self.fixed_values, self.fixed_polys, self.fixed_cosets = fixed_values, fixed_polys, fixed_cosets      # cosets internal form
        self.perm_values, self.perm_polys, self.perm_cosets = perm_values, perm_polys, perm_cosets
        f = vk.curve.scalar
        p = f.p
        self.custom_gates = plonk.custom_gates_graph(vk.cs, p).compile(ctx, f)
        self.lookup_graphs = [plonk.lookup_table_value_graph(i, t, p).compile(ctx, f) for i, t in vk.cs.lookups]
        self.compress_graphs = [(plonk.compress_graph(i, p).compile(ctx, f), plonk.compress_graph(t, p).compile(ctx, f)) for i, t in vk.cs.lookups]

    def _std(self, t):
        import torch

        out = torch.empty_like(t)
        self.ctx.convert_form_device(self.vk.curve.scalar.id, t.data_ptr(), out.data_ptr(), t.numel() // 4, False, 0)
        self.ctx.synchronize()
        return to_host(out)

    def write(self, fh):
        with self.ctx.torch_stream():
            self._write(fh)

    def _write(self, fh):
        self.vk.write(fh)
        l = self._std(self.l_ext)
        for i in range(3):
            _write_poly(fh, l[i])
        _write_poly_slice(fh, list(to_host(self.fixed_values)))
        _write_poly_slice(fh, list(to_host(self.fixed_polys)))
        _write_poly_slice(fh, list(self._std(self.fixed_cosets)) if self.fixed_cosets.shape[0] else [])
        _write_poly_slice(fh, list(to_host(self.perm_values)))
        _write_poly_slice(fh, list(to_host(self.perm_polys)))
        _write_poly_slice(fh, list(self._std(self.perm_cosets)) if self.perm_cosets.shape[0] else [])

    @classmethod
    def read(cls, ctx: Context, curve: CurveSpec, cs: plonk.ConstraintSystem, fh, num_selectors: int = 0) -> "ProvingKey":
        with ctx.torch_stream():
            return cls._read(ctx, curve, cs, fh, num_selectors)

    @classmethod
    def _read(cls, ctx: Context, curve: CurveSpec, cs: plonk.ConstraintSystem, fh, num_selectors: int = 0) -> "ProvingKey":
        vk = VerifyingKey.read(curve, cs, fh, num_selectors)
        f = curve.scalar
        domain = EvaluationDomain(ctx, f, cs.degree(), vk.k)
        n, m = domain.n, domain.extended_len()
        l = np.stack([_read_poly(fh, m) for _ in range(3)])
        nf, npc = cs.num_fixed, len(cs.permutation_columns)
        fv, fp, fc = _read_poly_slice(fh, nf, n), _read_poly_slice(fh, nf, n), _read_poly_slice(fh, nf, m)
        pv, pp, pc = _read_poly_slice(fh, npc, n), _read_poly_slice(fh, npc, n), _read_poly_slice(fh, npc, m)
        dev = [to_device(a) for a in (l, fv, fp, fc, pv, pp, pc)]
        for t in (dev[0], dev[3], dev[6]):
            if t.numel():
                ctx.convert_form_device(f.id, t.data_ptr(), t.data_ptr(), t.numel() // 4, True, 0)
        ctx.synchronize()
        return cls(ctx, vk, domain, *dev)


def vk_size(cs: plonk.ConstraintSystem, k: int, num_selectors: int) -> int:
    """Bytes VerifyingKey.write produces (RawBytes): what the reference lists as |vk| (benches/README.md:56-60)."""
    return 4 + 4 + 64 * cs.num_fixed + 64 * len(cs.permutation_columns) + num_selectors * (((1 << k) + 7) // 8)


def pk_size(cs: plonk.ConstraintSystem, k: int, num_selectors: int, field: FieldSpec) -> int:
    """Bytes ProvingKey.write produces (RawBytes): the reference's |pk|."""
    n = 1 << k
    m = EvaluationDomain(None, field, cs.degree(), k).extended_len()
    poly = lambda ln: 4 + 32 * ln
    sl = lambda cnt, ln: 4 + cnt * poly(ln)
    nf, npc = cs.num_fixed, len(cs.permutation_columns)
    return vk_size(cs, k, num_selectors) + 3 * poly(m) + sl(nf, n) + sl(nf, n) + sl(nf, m) + sl(npc, n) + sl(npc, n) + sl(npc, m)


def delta_of(f: FieldSpec) -> int:
    """PrimeField::DELTA = MULTIPLICATIVE_GENERATOR^(2^S): generates the odd-order subgroup."""
    return pow(f.gen, 1 << f.two_adicity, f.p)


def omega_powers_device(ctx: Context, domain: EvaluationDomain):
    """(n, 4) device column of omega^i, as the forward NTT of the unit vector e_1."""
    import torch

    f = domain.field
    col = torch.zeros((domain.n, 4), dtype=torch.int64, device="cuda")
    if domain.n > 1:
        col[1] = to_device(f.encode(1).reshape(1, 4))[0]
        ctx.ntt_device(f.id, col.data_ptr(), domain.k, f.encode(domain.omega), 1, 0)
    else:
        col[0] = to_device(f.encode(1).reshape(1, 4))[0]
    return col


def keygen(ctx: Context, params: ParamsKZG, cs: plonk.ConstraintSystem, fixed_canonical: np.ndarray, assembly: plonk.Assembly,
           selectors: Sequence[np.ndarray] = ()) -> ProvingKey:
    with ctx.torch_stream():           # torch's copies and fills go on the context's stream, ordered with the kernels
        return _keygen(ctx, params, cs, fixed_canonical, assembly, selectors)


def _keygen(ctx: Context, params: ParamsKZG, cs: plonk.ConstraintSystem, fixed_canonical: np.ndarray, assembly: plonk.Assembly,
            selectors: Sequence[np.ndarray] = ()) -> ProvingKey:
    """keygen_vk + keygen_pk for a circuit whose fixed columns (selectors already compressed into them) and copy
    constraints are given: fixed_canonical = (num_fixed, n, 4) u64 canonical values (rows >= usable must be zero)."""
    import torch

    curve, f, k = params.curve, params.curve.scalar, params.k
    domain = EvaluationDomain(ctx, f, cs.degree(), k)
    n, m, ek = domain.n, domain.extended_len(), domain.extended_k
    bf = cs.blinding_factors()
    if n < bf + 3:
        raise ValueError("not enough rows available")                          # upstream: Error::NotEnoughRowsAvailable
    fixed_canonical = np.ascontiguousarray(fixed_canonical, dtype=np.uint64).reshape(cs.num_fixed, n, 4)
    enc = f.encode
    c = dict(omega_inv=enc(domain.omega_inv), ifft=enc(domain.ifft_divisor), ext_omega=enc(domain.extended_omega), zeta=enc(domain.g_coset))

    def lagrange_to_all(values):
        """values (cnt, n, 4) Montgomery on device -> (commitments Montgomery affine on host, polys, cosets internal)."""
        cnt = values.shape[0]
        polys = values.clone()
        cosets = torch.zeros((cnt, m, 4), dtype=torch.int64, device="cuda")
        jac = torch.zeros((max(cnt, 1), 12), dtype=torch.int64, device="cuda")
        aff = torch.zeros((max(cnt, 1), 8), dtype=torch.int64, device="cuda")
        if cnt:
            params.commit_device(values.data_ptr(), cnt, jac.data_ptr(), True)
            ctx.to_affine_device(curve.id, jac.data_ptr(), cnt, aff.data_ptr(), 0)
            ctx.intt_scaled_device(f.id, polys.data_ptr(), k, c["omega_inv"], c["ifft"], cnt, 0)
            ctx.coset_ntt_form_device(f.id, polys.data_ptr(), k, cosets.data_ptr(), ek, c["ext_omega"], c["zeta"], cnt, ev.FORM_OUT_INTERNAL, 0)
        ctx.synchronize()
        return to_host(aff)[:cnt], polys, cosets

    # fixed columns
    fixed_values = to_device(fixed_canonical)
    if cs.num_fixed:
        ctx.field_op_device(f.id, "to_mont", fixed_values.data_ptr(), 0, fixed_values.data_ptr(), cs.num_fixed * n, 0)
    fixed_commitments, fixed_polys, fixed_cosets = lagrange_to_all(fixed_values)

    # permutation: sigma_j(omega^i) = delta^(column of the mapped cell) * omega^(its row)
    npc = len(cs.permutation_columns)
    if assembly.num_columns != npc or assembly.n != n:
        raise ValueError("assembly does not match the constraint system")
    ident = torch.zeros((npc, n, 4), dtype=torch.int64, device="cuda")
    w = omega_powers_device(ctx, domain)
    delta, dj = delta_of(f), 1
    for j in range(npc):
        ident[j].copy_(w)
        if j:
            ctx.scale_device(f.id, ident[j].data_ptr(), n, enc(dj).reshape(1, 4), 0, 0)
        dj = dj * delta % f.p
    ctx.synchronize()
    perm_values = ident.view(npc * n, 4)[to_device_index(assembly.mapping)].view(npc, n, 4).contiguous() if npc else ident
    perm_commitments, perm_polys, perm_cosets = lagrange_to_all(perm_values)

    # l0, l_last, l_active_row = 1 - (l_last + l_blind) in the extended domain
    one = to_device(enc(1).reshape(1, 4))[0]
    u = n - bf - 1
    lag = torch.zeros((3, n, 4), dtype=torch.int64, device="cuda")
    lag[0, 0] = one
    lag[1, u] = one
    lag[2, :u] = one
    _, _, l_ext = lagrange_to_all(lag)

    vk = VerifyingKey(curve, k, cs, fixed_commitments, perm_commitments, selectors)
    return ProvingKey(ctx, vk, domain, l_ext, fixed_values, fixed_polys, fixed_cosets, perm_values, perm_polys, perm_cosets)


def to_device_index(idx: np.ndarray):
    from ._lib import transfer_context

    with transfer_context() as c:
        return c.upload(np.ascontiguousarray(idx, dtype=np.int64))
