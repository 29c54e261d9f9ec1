"""ctypes wrapper of oracle/liboracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        P = C.c_void_p
        _LIB.orc_best_multiexp.argtypes = [C.c_int, P, P, C.c_size_t, C.c_int, P]
        _LIB.orc_to_affine.argtypes = [C.c_int, P, P]
        _LIB.orc_best_fft.argtypes = [C.c_int, P, P, C.c_uint32, C.c_int]
        _LIB.orc_lagrange_to_coeff.argtypes = [C.c_int, P, C.c_uint32, P, P, C.c_int]
        _LIB.orc_coeff_to_extended.argtypes = [C.c_int, P, C.c_uint32, P, C.c_uint32, P, P, C.c_int]
        _LIB.orc_extended_to_coeff.argtypes = [C.c_int, P, C.c_uint32, P, P, P, C.c_int]
        _LIB.orc_field_op.argtypes = [C.c_int, C.c_int, P, P, P, C.c_size_t]
        _LIB.orc_field_info.argtypes = [C.c_int, P, P, P, P]
        _LIB.orc_fill_scalars.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_size_t, P]
        _LIB.orc_synth_bases.argtypes = [C.c_int, C.c_size_t, P]
        _LIB.orc_graph_evaluate.argtypes = [C.c_int, P, P, C.c_uint32, P, C.c_uint32, P, C.c_uint32, P, P, P, P, P, C.c_uint32, C.c_uint32, P, P, C.c_int]
        _LIB.orc_permutation_h.argtypes = [C.c_int, P, C.c_uint32, P, P, C.c_uint32, C.c_uint32, C.c_int32, P, P, P, P, C.c_uint32, C.c_uint32, P, C.c_int]
        _LIB.orc_lookup_h.argtypes = [C.c_int, P, P, P, P, P, P, P, P, C.c_uint32, C.c_uint32, P, C.c_int]
        _LIB.orc_permute_expression_pair.argtypes = [C.c_int, P, P, C.c_size_t, P, P]
        _LIB.orc_eval_polynomial.argtypes = [C.c_int, P, C.c_size_t, P, C.c_int, P]
        _LIB.orc_batch_invert.argtypes = [C.c_int, P, C.c_size_t]
        _LIB.orc_grand_product.argtypes = [C.c_int, P, P, C.c_size_t, P]
        _LIB.orc_kate_division.argtypes = [C.c_int, P, C.c_size_t, P, P]
        _LIB.orc_lincomb.argtypes = [C.c_int, P, P, C.c_size_t, C.c_size_t, P, P]
        _LIB.orc_scale_periodic.argtypes = [C.c_int, P, C.c_size_t, P, C.c_size_t]
        _LIB.orc_fixed_base_mul.argtypes = [C.c_int, P, C.c_size_t, C.c_int, P]
        _LIB.orc_powers.argtypes = [C.c_int, P, P, C.c_size_t, P]
    return _LIB


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def limbs(x: int) -> np.ndarray:
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


OPS = {"add": 0, "sub": 1, "mul": 2, "inv": 3, "to_mont": 4, "from_mont": 5}


def field_op(field: int, op: str, a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    bp = _p(np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)) if b is not None else None
    rc = lib().orc_field_op(field, OPS[op], _p(a), bp, _p(out), a.shape[0])
    assert rc == 0
    return out


def best_multiexp(curve: int, scalars: np.ndarray, bases: np.ndarray, threads: int = 1) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    assert scalars.shape[0] == bases.shape[0]
    out = np.zeros(12, dtype=np.uint64)
    rc = lib().orc_best_multiexp(curve, _p(scalars), _p(bases), scalars.shape[0], threads, _p(out))
    assert rc == 0
    return out


def to_affine(curve: int, jac: np.ndarray) -> np.ndarray:
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(12)
    out = np.zeros(8, dtype=np.uint64)
    assert lib().orc_to_affine(curve, _p(jac), _p(out)) == 0
    return out


def best_fft(field: int, a: np.ndarray, omega: np.ndarray, log_n: int, threads: int = 1) -> np.ndarray:
    a = np.array(a, dtype=np.uint64).reshape(-1, 4)
    assert a.shape[0] == 1 << log_n
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    assert lib().orc_best_fft(field, _p(a), _p(omega), log_n, threads) == 0
    return a


def lagrange_to_coeff(field, a, k, omega_inv, divisor, threads=1):
    a = np.array(a, dtype=np.uint64).reshape(-1, 4)
    assert lib().orc_lagrange_to_coeff(field, _p(a), k, _p(np.ascontiguousarray(omega_inv)), _p(np.ascontiguousarray(divisor)), threads) == 0
    return a


def coeff_to_extended(field, coeffs, k, ext_k, ext_omega, zeta, threads=1):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    ext = np.zeros(((1 << ext_k), 4), dtype=np.uint64)
    assert lib().orc_coeff_to_extended(field, _p(coeffs), k, _p(ext), ext_k, _p(np.ascontiguousarray(ext_omega)), _p(np.ascontiguousarray(zeta)), threads) == 0
    return ext


def extended_to_coeff(field, a, ext_k, ext_omega_inv, divisor, zeta, threads=1):
    a = np.array(a, dtype=np.uint64).reshape(-1, 4)
    assert lib().orc_extended_to_coeff(field, _p(a), ext_k, _p(np.ascontiguousarray(ext_omega_inv)), _p(np.ascontiguousarray(divisor)), _p(np.ascontiguousarray(zeta)), threads) == 0
    return a


DISTS = {"uniform": 0, "witness": 1, "lookup": 2}


def fill_scalars(field: int, dist: str, n: int, seed: int = 0x64656C6179656E63) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    assert lib().orc_fill_scalars(field, DISTS[dist], seed, n, _p(out)) == 0
    return out


def synth_bases(curve: int, n: int) -> np.ndarray:
    out = np.zeros((n, 8), dtype=np.uint64)
    assert lib().orc_synth_bases(curve, n, _p(out)) == 0
    return out


def eval_polynomial(field: int, coeffs: np.ndarray, point: np.ndarray, threads: int = 1) -> np.ndarray:
    c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    assert lib().orc_eval_polynomial(field, _p(c), c.shape[0], _p(np.ascontiguousarray(point, dtype=np.uint64)), threads, _p(out)) == 0
    return out


def batch_invert(field: int, values: np.ndarray) -> np.ndarray:
    v = np.array(values, dtype=np.uint64).reshape(-1, 4)
    assert lib().orc_batch_invert(field, _p(v), v.shape[0]) == 0
    return v


def grand_product(field: int, num: np.ndarray, den: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(num, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(den, dtype=np.uint64).reshape(-1, 4)
    z = np.zeros_like(a)
    assert lib().orc_grand_product(field, _p(a), _p(b), a.shape[0], _p(z)) == 0
    return z


# ---- quotient numerator (Montgomery n x 4 u64 columns) ----
class _Src(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("rotation", C.c_uint32)]


class _Calc(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", _Src), ("b", _Src), ("parts_begin", C.c_uint32), ("parts_len", C.c_uint32), ("target", C.c_uint32)]


def _cols(cols):
    keep = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4) for c in cols]
    return keep, (C.c_void_p * max(1, len(keep)))(*[k.ctypes.data for k in keep])


def graph_evaluate(field: int, constants, rotations, calcs, num_intermediates, fixed, advice, instance, challenges, beta, gamma, theta, y, log_rows: int,
                   rot_scale: int, previous=None, threads: int = 1) -> np.ndarray:
    """constants / challenges: k x 4 u64; calcs: [(op, a, b, parts, target)] with sources (kind, index, rot); columns: lists of rows x 4 u64."""
    cst = np.ascontiguousarray(constants, dtype=np.uint64).reshape(-1, 4) if len(constants) else np.zeros((1, 4), dtype=np.uint64)
    rot = (C.c_int32 * max(1, len(rotations)))(*rotations)
    flat = []
    cc = (_Calc * max(1, len(calcs)))()
    for i, (op, a, b, parts, target) in enumerate(calcs):
        cc[i] = _Calc(op, _Src(*a), _Src(*b), len(flat), len(parts), target)
        flat.extend(parts)
    pp = (_Src * max(1, len(flat)))(*[_Src(*q) for q in flat])
    kf, tf = _cols(fixed); ka, ta = _cols(advice); ki, ti = _cols(instance)
    ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4) if challenges is not None and len(challenges) else np.zeros((1, 4), dtype=np.uint64)
    bgty = np.zeros((4, 4), dtype=np.uint64)
    for i, v in enumerate((beta, gamma, theta, y)):
        if v is not None:
            bgty[i] = v
    prev = np.ascontiguousarray(previous, dtype=np.uint64).reshape(-1, 4) if previous is not None else None
    out = np.zeros((1 << log_rows, 4), dtype=np.uint64)
    rc = lib().orc_graph_evaluate(field, _p(cst), rot, len(rotations), cc, len(calcs), pp, num_intermediates, tf, ta, ti, _p(ch), _p(bgty), log_rows, rot_scale,
                                  _p(prev) if prev is not None else None, _p(out), threads)
    assert rc == 0
    return out


def permutation_h(field: int, values, z, columns, sigma, chunk_len: int, last_rotation: int, l0, l_last, l_active, beta, gamma, y, delta, beta_zeta,
                  extended_omega, log_rows: int, rot_scale: int, threads: int = 1) -> np.ndarray:
    kz, tz = _cols(z); kc, tc = _cols(columns); ks, ts = _cols(sigma)
    sc = np.stack([np.asarray(v, dtype=np.uint64).reshape(4) for v in (beta, gamma, y, delta, beta_zeta, extended_omega)])
    v = np.array(values, dtype=np.uint64).reshape(-1, 4)
    a0, a1, a2 = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (l0, l_last, l_active))
    rc = lib().orc_permutation_h(field, tz, len(kz), tc, ts, len(kc), chunk_len, last_rotation, _p(a0), _p(a1), _p(a2), _p(sc), log_rows, rot_scale, _p(v), threads)
    assert rc == 0
    return v


def lookup_h(field: int, values, product, permuted_input, permuted_table, table_value, l0, l_last, l_active, beta, gamma, y, log_rows: int, rot_scale: int,
             threads: int = 1) -> np.ndarray:
    arrs = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (product, permuted_input, permuted_table, table_value, l0, l_last, l_active)]
    sc = np.stack([np.asarray(v, dtype=np.uint64).reshape(4) for v in (beta, gamma, y)])
    v = np.array(values, dtype=np.uint64).reshape(-1, 4)
    rc = lib().orc_lookup_h(field, *[_p(a) for a in arrs], _p(sc), log_rows, rot_scale, _p(v), threads)
    assert rc == 0
    return v


def permute_expression_pair(field: int, input_values, table_values, usable_rows: int):
    """-> (permuted_input, permuted_table) or None (upstream: Err(ConstraintSystemFailure))."""
    a = np.ascontiguousarray(input_values, dtype=np.uint64).reshape(-1, 4)
    t = np.ascontiguousarray(table_values, dtype=np.uint64).reshape(-1, 4)
    pi, pt = np.zeros((usable_rows, 4), dtype=np.uint64), np.zeros((usable_rows, 4), dtype=np.uint64)
    rc = lib().orc_permute_expression_pair(field, _p(a), _p(t), usable_rows, _p(pi), _p(pt))
    assert rc in (0, 1)
    return None if rc else (pi, pt)


# ---- round 2: element-wise steps of create_proof, SRS generation ----
def kate_division(field: int, a: np.ndarray, point: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    q = np.zeros((a.shape[0] - 1, 4), dtype=np.uint64)
    assert lib().orc_kate_division(field, _p(a), a.shape[0], _p(np.ascontiguousarray(point, dtype=np.uint64)), _p(q) if q.shape[0] else None) == 0
    return q


def lincomb(field: int, cols, coefs, sub0=None) -> np.ndarray:
    keep, tbl = _cols(cols)
    cf = np.ascontiguousarray(coefs, dtype=np.uint64).reshape(-1, 4)
    assert cf.shape[0] == len(keep) and len(keep) > 0
    out = np.zeros_like(keep[0])
    sp = _p(np.ascontiguousarray(sub0, dtype=np.uint64).reshape(4)) if sub0 is not None else None
    assert lib().orc_lincomb(field, tbl, _p(cf), len(keep), out.shape[0], _p(out), sp) == 0
    return out


def scale_periodic(field: int, a, pattern) -> np.ndarray:
    a = np.array(a, dtype=np.uint64).reshape(-1, 4)
    pat = np.ascontiguousarray(pattern, dtype=np.uint64).reshape(-1, 4)
    assert lib().orc_scale_periodic(field, _p(a), a.shape[0], _p(pat), pat.shape[0]) == 0
    return a


def fixed_base_mul(curve: int, scalars, threads: int = 1) -> np.ndarray:
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((s.shape[0], 8), dtype=np.uint64)
    assert lib().orc_fixed_base_mul(curve, _p(s), s.shape[0], threads, _p(out)) == 0
    return out


def powers(field: int, base, first, n: int) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    assert lib().orc_powers(field, _p(np.ascontiguousarray(base, dtype=np.uint64).reshape(4)), _p(np.ascontiguousarray(first, dtype=np.uint64).reshape(4)), n, _p(out)) == 0
    return out
