"""ParamsKZG / ParamsIPA ``commit`` / ``commit_lagrange`` mirror
(halo2_proofs/src/poly/kzg/commitment.rs, .../ipa/commitment.rs @ v2023_04_20; built at
benches/delay_enc.rs:41-54, the IPA variant at :42,51-52): thin wrappers -> best_multiexp over the device-resident SRS.

``Params``     the KZG flavour: commit(poly) = <poly, g>, the blind is ignored (as upstream's ParamsKZG::commit does).
``ParamsIPA``  the IPA flavour over the Pasta curves: commit(poly, r) = <poly, g> + r * w  -- an MSM of n + 1 terms
               ("tmp_scalars.extend(poly); tmp_scalars.push(r.0); tmp_bases.extend(g); tmp_bases.push(w)" upstream), served by
               tables registered over g || w and g_lagrange || w."""
from __future__ import annotations

from typing import Optional

import numpy as np

from ._lib import Bases, Context
from .fields import CurveSpec


class Params:
    def __init__(self, ctx: Context, curve: CurveSpec, k: int, g, g_lagrange=None, window_bits: int = 0, precompute: bool = True):
        self.ctx, self.curve, self.k, self.n = ctx, curve, k, 1 << k
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
        if g.shape[0] != self.n:
            raise ValueError("g must hold 2^k points")
        self.g: Bases = ctx.register_bases(curve.id, g, window_bits, precompute)
        self.g_lagrange: Optional[Bases] = None
        if g_lagrange is not None:
            gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
            if gl.shape[0] != self.n:
                raise ValueError("g_lagrange must hold 2^k points")
            self.g_lagrange = ctx.register_bases(curve.id, gl, window_bits, precompute)

    def commit(self, poly) -> np.ndarray:
        """best_multiexp(poly, g[..len]) -> Jacobian (KZG ignores the blind)."""
        return self.ctx.msm(self.g, poly)

    def commit_lagrange(self, poly) -> np.ndarray:
        if self.g_lagrange is None:
            raise ValueError("no g_lagrange registered")
        return self.ctx.msm(self.g_lagrange, poly)

    def commit_many(self, polys, lagrange: bool = False) -> np.ndarray:
        """One launch for a phase's independent columns (msm_batch)."""
        b = self.g_lagrange if lagrange else self.g
        return self.ctx.msm_batch(b, polys)

    def release(self):
        self.g.release()
        if self.g_lagrange is not None:
            self.g_lagrange.release()


class ParamsIPA:
    """ParamsIPA<C>: g, g_lagrange (2^k points each) and the blinding base w."""

    def __init__(self, ctx: Context, curve: CurveSpec, k: int, g, g_lagrange, w, window_bits: int = 0, precompute: bool = True):
        self.ctx, self.curve, self.k, self.n = ctx, curve, k, 1 << k
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
        gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
        w = np.ascontiguousarray(w, dtype=np.uint64).reshape(1, 8)
        if g.shape[0] != self.n or gl.shape[0] != self.n:
            raise ValueError("g and g_lagrange must hold 2^k points")
        self.g: Bases = ctx.register_bases(curve.id, np.concatenate([g, w]), window_bits, precompute)
        self.g_lagrange: Bases = ctx.register_bases(curve.id, np.concatenate([gl, w]), window_bits, precompute)

    def _commit(self, bases: Bases, poly, blind) -> np.ndarray:
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        if poly.shape[0] != self.n:
            raise ValueError("poly.len() != params.n")                 # upstream: assert_eq!(poly.len(), self.n as usize)
        blind = np.ascontiguousarray(blind, dtype=np.uint64).reshape(1, 4)
        return self.ctx.msm(bases, np.concatenate([poly, blind]))

    def commit(self, poly, blind) -> np.ndarray:
        """<poly, g> + blind * w -> Jacobian; poly and blind in Montgomery limbs."""
        return self._commit(self.g, poly, blind)

    def commit_lagrange(self, poly, blind) -> np.ndarray:
        return self._commit(self.g_lagrange, poly, blind)

    def release(self):
        self.g.release()
        self.g_lagrange.release()
