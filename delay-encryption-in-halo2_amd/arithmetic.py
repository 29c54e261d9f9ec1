"""halo2_proofs::arithmetic::{best_multiexp, best_fft} mirrors
(halo2_proofs/src/arithmetic.rs @ v2023_04_20)."""
from __future__ import annotations

import numpy as np

from ._lib import Context
from .fields import CurveSpec, FieldSpec


def best_multiexp(ctx: Context, curve: CurveSpec, coeffs, bases) -> np.ndarray:
    """sum_i coeffs[i] * bases[i] -> Jacobian {x, y, z} (12 u64).  coeffs: n x 4 u64
    Montgomery scalars; bases: n x 8 u64 affine points.  Raises ValueError on a length
    mismatch (upstream: assert_eq! panic)."""
    return ctx.best_multiexp(curve.id, coeffs, bases)


def best_fft(ctx: Context, field: FieldSpec, a, omega, log_n: int) -> np.ndarray:
    """a'[i] = sum_j a[j] * omega^(i*j); natural order in and out; no scaling.  Returns the
    transformed copy (upstream works in place on &mut [F])."""
    return ctx.ntt(field.id, a, log_n, omega)
