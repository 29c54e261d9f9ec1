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


def eval_polynomial(ctx: Context, field: FieldSpec, poly, point) -> np.ndarray:
    """sum_i poly[i] * point^i (4 u64, Montgomery); 0 for an empty polynomial.  Mirrors
    halo2_proofs::arithmetic::eval_polynomial(poly: &[F], point: F) -> F."""
    return ctx.eval_polynomial(field.id, poly, point)


def batch_invert(ctx: Context, field: FieldSpec, values) -> np.ndarray:
    """Element-wise inverses; zero elements stay zero (ff::BatchInvert::batch_invert as upstream's
    grand-product builders use it).  Returns the inverted copy (upstream works in place)."""
    return ctx.batch_invert(field.id, values)


def grand_product(ctx: Context, field: FieldSpec, num, den) -> np.ndarray:
    """z[0] = 1, z[i] = prod_{j<i} num[j] / den[j]: the running product of
    permutation::Argument::commit / lookup::Permuted::commit_product without the blinding rows.
    Raises ValueError on a length mismatch."""
    return ctx.grand_product(field.id, num, den)
