"""The MSM / NTT schedule of one halo2 `create_proof` for the delay-encryption circuit shape
(reference call site benches/delay_enc.rs:123-131; shape derived in SURVEY.md 8(d) and
Appendix C: 5 advice columns, 5 lookup arguments, 2 permutation products, degree 5 ->
extended domain 4n), with synthetic device-resident columns standing in for the witness.

Per proof (KZG/GWC over BN254, or the Pasta variant):
  phase 1   5 x commit_lagrange(advice)                 witness-like scalars
  phase 2  10 x commit_lagrange(permuted lookup cols)   lookup-like scalars
  phase 3   7 x commit_lagrange(grand products)         uniform
  phase 4   1 x commit(random poly)                     uniform
  phase 5  24 x lagrange_to_coeff (iNTT n), 23 x coeff_to_extended (coset NTT n -> 4n),
            [evaluate_h: field-only, not on this path], 1 x extended_to_coeff (iNTT 4n),
            4 x commit(h pieces)
  phase 6   4 x commit(opening quotients)
Phases are separated by a host synchronisation (the transcript squeeze happens on the host);
inside a phase the independent columns go through ONE batched launch.

Only the hot-path calls are made; nothing here computes a real proof.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List

import numpy as np

from ._lib import Bases, Context
from .domain import EvaluationDomain
from .fields import CurveSpec

MSM_PHASES = [("advice", 5, "witness"), ("lookup_permuted", 10, "lookup"), ("grand_products", 7, "uniform"),
              ("random", 1, "uniform"), ("h_pieces", 4, "uniform"), ("openings", 4, "uniform")]
N_INTT, N_COSET, N_EXT_INTT = 24, 23, 1


@dataclass
class ProverShapeResult:
    commitments: np.ndarray      # 31 x 12 u64 Jacobian
    ms_total: float
    ms_msm: float
    ms_ntt: float


class ProverShape:
    """Owns the device buffers of one proof's columns (torch tensors) and replays the schedule."""

    def __init__(self, ctx: Context, curve: CurveSpec, k: int, g_lagrange: Bases, g: Bases, columns: dict, j: int = 5):
        import torch

        self.torch = torch
        self.ctx, self.curve, self.k, self.n = ctx, curve, k, 1 << k
        self.g_lagrange, self.g = g_lagrange, g
        self.domain = EvaluationDomain(ctx, curve.scalar, j, k)
        self.ext_n = 1 << self.domain.extended_k
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()
        # columns[name]: (count, n, 4) u64 Montgomery scalars
        self.cols = {name: to_dev(columns[name]) for name, _, _ in MSM_PHASES}
        self.polys = to_dev(columns["polys"])                      # (24, n, 4): columns taken through iNTT
        self.ext = torch.zeros((N_COSET, self.ext_n, 4), dtype=torch.int64, device="cuda")
        self.out = {name: torch.zeros((cnt, 12), dtype=torch.int64, device="cuda") for name, cnt, _ in MSM_PHASES}
        d, e = self.domain, curve.scalar.encode
        self._c = dict(omega_inv=e(d.omega_inv), ifft=e(d.ifft_divisor), ext_omega=e(d.extended_omega), ext_omega_inv=e(d.extended_omega_inv),
                       ext_ifft=e(d.extended_ifft_divisor), zeta=e(d.g_coset))

    def _commit(self, name: str, lagrange: bool):
        t = self.cols[name]
        b = self.g_lagrange if lagrange else self.g
        self.ctx.msm_device(b, t.data_ptr(), self.n, t.shape[0], self.out[name].data_ptr(), 0)

    def run(self) -> ProverShapeResult:
        ctx, c, f = self.ctx, self._c, self.curve.scalar
        sync = ctx.synchronize
        t_msm = t_ntt = 0.0
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for name in ("advice", "lookup_permuted", "grand_products"):
            self._commit(name, True); sync()                       # transcript squeeze on the host
        self._commit("random", False); sync()
        t1 = time.perf_counter(); t_msm += t1 - t0
        ctx.intt_scaled_device(f.id, self.polys.data_ptr(), self.k, c["omega_inv"], c["ifft"], N_INTT, 0)
        ctx.coset_ntt_device(f.id, self.polys.data_ptr(), self.k, self.ext.data_ptr(), self.domain.extended_k, c["ext_omega"], c["zeta"], N_COSET, 0)
        ctx.coset_intt_device(f.id, self.ext.data_ptr(), self.domain.extended_k, c["ext_omega_inv"], c["ext_ifft"], c["zeta"], N_EXT_INTT, 0)
        sync()
        t2 = time.perf_counter(); t_ntt += t2 - t1
        self._commit("h_pieces", False); sync()
        self._commit("openings", False); sync()
        t3 = time.perf_counter(); t_msm += t3 - t2
        outs = np.concatenate([self.out[name].cpu().numpy().view(np.uint64) for name, _, _ in MSM_PHASES])
        return ProverShapeResult(outs, 1e3 * (t3 - t0), 1e3 * t_msm, 1e3 * t_ntt)


def synthetic_columns(fill_scalars, scalar_field_id: int, k: int, seed: int = 1) -> dict:
    """fill_scalars(field_id, dist, n, seed) -> (n, 4) u64 (the test/bench side supplies the generator)."""
    n = 1 << k
    cols = {}
    s = seed
    for name, cnt, dist in MSM_PHASES:
        cols[name] = np.stack([fill_scalars(scalar_field_id, dist, n, s + i) for i in range(cnt)])
        s += cnt
    cols["polys"] = np.stack([fill_scalars(scalar_field_id, "uniform", n, s + i) for i in range(N_INTT)])
    return cols
