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
            evaluate_h over 4n rows (with_quotient=True: custom gates, 2 permutation sets, 5 lookups on
            the device; otherwise skipped), 1 x extended_to_coeff (iNTT 4n),
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

from . import evaluation as ev
from ._lib import Bases, Context
from .domain import EvaluationDomain
from .fields import CurveSpec, FieldSpec  # noqa: F401

MSM_PHASES = [("advice", 5, "witness"), ("lookup_permuted", 10, "lookup"), ("grand_products", 7, "uniform"),
              ("random", 1, "uniform"), ("h_pieces", 4, "uniform"), ("openings", 4, "uniform")]
N_INTT, N_COSET, N_EXT_INTT = 24, 23, 1


N_FIXED, N_SIGMA, N_LOOKUPS, PERM_CHUNK = 15, 6, 5, 3      # 15 fixed, 6 permutation columns, 5 lookups, degree 5
LAST_ROTATION = -6                                         # -(blinding_factors + 1), blinding_factors = 5


@dataclass
class ProverShapeResult:
    commitments: np.ndarray      # 31 x 12 u64 Jacobian
    ms_total: float
    ms_msm: float
    ms_ntt: float
    ms_eval_h: float = 0.0
    ms_arguments: float = 0.0     # lookup permutations + grand products
    ms_openings: float = 0.0      # eval_polynomial of the opened polynomials


def maingate_graph() -> ev.GraphEvaluator:
    """The custom-gate program of the circuit shape: sum_i q_i a_i + q_ab a b + q_cd c d + q_e' e(wX)
    + q_const + instance, folded into the running value with y (values * y + gate)."""
    g = ev.GraphEvaluator()
    adv = [g.column(ev.ADVICE, i) for i in range(5)]
    fx = [g.column(ev.FIXED, i) for i in range(9)]
    terms = [g.add_calculation(ev.MUL, adv[i], fx[i]) for i in range(5)]
    terms.append(g.add_calculation(ev.MUL, g.add_calculation(ev.MUL, adv[0], adv[1]), fx[5]))
    terms.append(g.add_calculation(ev.MUL, g.add_calculation(ev.MUL, adv[2], adv[3]), fx[6]))
    terms.append(g.add_calculation(ev.MUL, g.column(ev.ADVICE, 4, 1), fx[7]))
    terms += [fx[8], g.column(ev.INSTANCE, 0)]
    acc = terms[0]
    for t in terms[1:]:
        acc = g.add_calculation(ev.ADD, acc, t)
    g.add_calculation(ev.HORNER, (ev.PREVIOUS, 0, 0), (ev.Y, 0, 0), (acc,))
    return g


def lookup_graph(i: int) -> ev.GraphEvaluator:
    """table_value of range lookup i: (q_i * a_i + beta) * (table + gamma)."""
    g = ev.GraphEvaluator()
    inp = g.add_calculation(ev.MUL, g.column(ev.ADVICE, i), g.column(ev.FIXED, 9 + i))
    lhs = g.add_calculation(ev.ADD, inp, (ev.BETA, 0, 0))
    rhs = g.add_calculation(ev.ADD, g.column(ev.FIXED, 14), (ev.GAMMA, 0, 0))
    g.add_calculation(ev.MUL, lhs, rhs)
    return g


class ProverShape:
    """Owns the device buffers of one proof's columns (torch tensors) and replays the schedule."""

    def __init__(self, ctx: Context, curve: CurveSpec, k: int, g_lagrange: Bases, g: Bases, columns: dict, j: int = 5, with_quotient: bool = False):
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
        self.out_affine = {name: torch.zeros((cnt, 8), dtype=torch.int64, device="cuda") for name, cnt, _ in MSM_PHASES}
        self.transcript_points = {name: torch.zeros((cnt, 8), dtype=torch.int64).pin_memory() for name, cnt, _ in MSM_PHASES}
        d, e = self.domain, curve.scalar.encode
        self._c = dict(omega_inv=e(d.omega_inv), ifft=e(d.ifft_divisor), ext_omega=e(d.extended_omega), ext_omega_inv=e(d.extended_omega_inv),
                       ext_ifft=e(d.extended_ifft_divisor), zeta=e(d.g_coset))
        self.with_quotient = with_quotient
        if with_quotient:
            # proving-key columns in the extended domain (resident for the life of the key) and the challenges
            self.pk = {name: to_dev(columns["pk_" + name]) for name in ("fixed", "sigma", "l")}          # (15 | 6 | 3, 4n, 4)
            for t in self.pk.values():   # key columns go to the kernels' internal form once (dehalo.h): no conversion per load later
                ctx.convert_form_device(curve.scalar.id, t.data_ptr(), t.data_ptr(), t.shape[0] * t.shape[1], True, 0)
            self.forms = ev.COLUMNS_INTERNAL | ev.VALUES_INTERNAL
            self.instance = torch.zeros((self.ext_n, 4), dtype=torch.int64, device="cuda")               # the circuit has no public inputs
            self.h = torch.zeros((self.ext_n, 4), dtype=torch.int64, device="cuda")
            self.table_value = torch.zeros((self.ext_n, 4), dtype=torch.int64, device="cuda")
            self.challenges = dict(columns["challenges"])                                                  # theta, beta, gamma, y, delta: canonical ints
            # lookup arguments: a 2^12-entry table padded with its first value (how halo2 pads range tables), inputs drawn from it
            self.usable = self.n - (-LAST_ROTATION)
            self.lk_in = to_dev(columns["lookup_inputs"])                                                      # (5, n, 4)
            self.lk_table = to_dev(np.repeat(columns["lookup_table"][None], N_LOOKUPS, axis=0))               # (5, n, 4): one table column per lookup
            self.lk_out = torch.zeros((2, N_LOOKUPS, self.n, 4), dtype=torch.int64, device="cuda")
            self.z = torch.zeros((7, self.n, 4), dtype=torch.int64, device="cuda")
            self.evals = torch.zeros((N_INTT, 4), dtype=torch.int64, device="cuda")
            self.gate_graph = maingate_graph().compile(ctx, curve.scalar)
            self.lookup_graphs = [lookup_graph(i).compile(ctx, curve.scalar) for i in range(N_LOOKUPS)]

    def evaluate_h(self):
        """Evaluator::evaluate_h on the device: advice = ext[0:5], (a', s') of lookup i = ext[5+2i], ext[6+2i],
        permutation z = ext[15:17], lookup z = ext[17:22]; result (the numerator, not yet divided by the
        vanishing polynomial -- that is one more element-wise product upstream) in self.h."""
        f, ch, rot_scale, log_rows = self.curve.scalar, self.challenges, self.ext_n // self.n, self.domain.extended_k
        col = lambda t, i: t[i].data_ptr()
        fixed = [col(self.pk["fixed"], i) for i in range(N_FIXED)]
        advice = [col(self.ext, i) for i in range(5)]
        l0, l_last, l_active = (col(self.pk["l"], i) for i in range(3))
        ff = self.forms
        self.gate_graph.evaluate_device(fixed, advice, [self.instance.data_ptr()], [], None, None, None, ch["y"], log_rows, rot_scale, 0, self.h.data_ptr(), 0, ff)
        ev.permutation_h_device(self.ctx, f, [col(self.ext, 15), col(self.ext, 16)], advice + [fixed[14]], [col(self.pk["sigma"], i) for i in range(N_SIGMA)],
                                PERM_CHUNK, LAST_ROTATION, l0, l_last, l_active, ch["beta"], ch["gamma"], ch["y"], ch["delta"], self.domain.g_coset,
                                self.domain.extended_omega, log_rows, rot_scale, self.h.data_ptr(), 0, ff)
        for i in range(N_LOOKUPS):
            self.lookup_graphs[i].evaluate_device(fixed, advice, [], [], ch["beta"], ch["gamma"], ch["theta"], None, log_rows, rot_scale, 0, self.table_value.data_ptr(),
                                                  0, ff)
            ev.lookup_h_device(self.ctx, f, col(self.ext, 17 + i), col(self.ext, 5 + 2 * i), col(self.ext, 6 + 2 * i), self.table_value.data_ptr(), l0, l_last,
                               l_active, ch["beta"], ch["gamma"], ch["y"], log_rows, rot_scale, self.h.data_ptr(), 0, ff)

    def arguments(self):
        """The data-parallel part of lookup::commit_permuted (5 x permute_expression_pair) and of the seven
        grand products (2 permutation sets + 5 lookups): batch-inverted denominators and the running product.
        (The element-wise numerator / denominator products that feed them are not modelled.)"""
        f = self.curve.scalar
        self.ctx.permute_expression_pair_batch_device(f.id, self.lk_in.data_ptr(), self.lk_table.data_ptr(), self.usable, N_LOOKUPS, self.n,
                                                      self.lk_out[0].data_ptr(), self.lk_out[1].data_ptr(), 0)          # the five lookups share the sort passes
        num, den = self.cols["grand_products"], self.cols["lookup_permuted"]
        self.ctx.grand_product_batch_device(f.id, num.data_ptr(), den.data_ptr(), self.n, 7, self.n, self.z.data_ptr(), 0)   # one inversion for all seven

    def openings(self):
        """eval_polynomial of every opened polynomial: three rotation sets of the 24 coefficient-form columns."""
        f, x = self.curve.scalar, self.curve.scalar.encode(self.challenges["y"])
        for _ in range(3):
            self.ctx.eval_polynomial_device(f.id, self.polys.data_ptr(), self.n, self.n, N_INTT, x, self.evals.data_ptr(), 0)

    def _commit(self, name: str, lagrange: bool):
        t = self.cols[name]
        b = self.g_lagrange if lagrange else self.g
        self.ctx.msm_device(b, t.data_ptr(), self.n, t.shape[0], self.out[name].data_ptr(), 0)
        # what the transcript absorbs: the phase's commitments as affine points, on the host
        self.ctx.to_affine_device(self.curve.id, self.out[name].data_ptr(), t.shape[0], self.out_affine[name].data_ptr(), 0)
        self.ctx.synchronize()
        self.transcript_points[name].copy_(self.out_affine[name])

    # columns whose coefficient form / coset can be computed as soon as the column exists: (first poly, count) per commit phase
    NTT_GROUPS = {"advice": (0, 6), "lookup_permuted": (6, 10), "grand_products": (16, 8)}

    def _ntt_group(self, ctx2: Context, first: int, count: int):
        """lagrange_to_coeff + coeff_to_extended of polys[first : first + count] on the second context's stream."""
        f, c = self.curve.scalar, self._c
        n_coset = min(count, N_COSET - first)
        p = self.polys[first].data_ptr()
        ctx2.intt_scaled_device(f.id, p, self.k, c["omega_inv"], c["ifft"], count, 0)
        if n_coset > 0:
            ctx2.coset_ntt_form_device(f.id, p, self.k, self.ext[first].data_ptr(), self.domain.extended_k, c["ext_omega"], c["zeta"], n_coset, ev.FORM_OUT_INTERNAL, 0)

    def run_overlapped(self, ctx2: Context) -> float:
        """The same schedule (with_quotient) with the NTTs of a phase's columns issued on a SECOND context while the
        first one runs that phase's MSMs: they fill the latency-bound tails of the bucket reductions.  A column's
        NTT starts when the column exists, never earlier.  Returns the wall time in ms."""
        assert self.with_quotient
        ctx, c, f = self.ctx, self._c, self.curve.scalar
        sync = ctx.synchronize
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for name in ("advice", "lookup_permuted", "grand_products"):
            self._ntt_group(ctx2, *self.NTT_GROUPS[name])
            self._commit(name, True); sync()
            if name == "advice":
                self.arguments(); sync()
        self._commit("random", False); sync()
        ctx2.synchronize()
        self.evaluate_h()
        ctx.coset_intt_form_device(f.id, self.h.data_ptr(), self.domain.extended_k, c["ext_omega_inv"], c["ext_ifft"], c["zeta"], N_EXT_INTT, ev.FORM_IN_INTERNAL, 0)
        sync()
        self._commit("h_pieces", False); sync()
        self.openings(); sync()
        self._commit("openings", False); sync()
        return 1e3 * (time.perf_counter() - t0)

    def run(self) -> ProverShapeResult:
        ctx, c, f = self.ctx, self._c, self.curve.scalar
        sync = ctx.synchronize
        t_msm = t_ntt = 0.0
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_arg = t_open = 0.0
        for name in ("advice", "lookup_permuted", "grand_products"):
            self._commit(name, True); sync()                       # transcript squeeze on the host
            if self.with_quotient and name == "advice":
                ta = time.perf_counter(); self.arguments(); sync(); t_arg = time.perf_counter() - ta
        self._commit("random", False); sync()
        t1 = time.perf_counter(); t_msm += t1 - t0 - t_arg
        ctx.intt_scaled_device(f.id, self.polys.data_ptr(), self.k, c["omega_inv"], c["ifft"], N_INTT, 0)
        t_h = 0.0
        if self.with_quotient:   # the cosets stay in the kernels' internal form from the NTT to evaluate_h and back
            ctx.coset_ntt_form_device(f.id, self.polys.data_ptr(), self.k, self.ext.data_ptr(), self.domain.extended_k, c["ext_omega"], c["zeta"], N_COSET,
                                      ev.FORM_OUT_INTERNAL, 0)
            sync(); th0 = time.perf_counter()
            self.evaluate_h()
            sync(); t_h = time.perf_counter() - th0
            ctx.coset_intt_form_device(f.id, self.h.data_ptr(), self.domain.extended_k, c["ext_omega_inv"], c["ext_ifft"], c["zeta"], N_EXT_INTT, ev.FORM_IN_INTERNAL, 0)
        else:
            ctx.coset_ntt_device(f.id, self.polys.data_ptr(), self.k, self.ext.data_ptr(), self.domain.extended_k, c["ext_omega"], c["zeta"], N_COSET, 0)
            ctx.coset_intt_device(f.id, self.ext.data_ptr(), self.domain.extended_k, c["ext_omega_inv"], c["ext_ifft"], c["zeta"], N_EXT_INTT, 0)
        sync()
        t2 = time.perf_counter(); t_ntt += t2 - t1 - t_h
        self._commit("h_pieces", False); sync()
        if self.with_quotient:
            to = time.perf_counter(); self.openings(); sync(); t_open = time.perf_counter() - to
        self._commit("openings", False); sync()
        t3 = time.perf_counter(); t_msm += t3 - t2 - t_open
        outs = np.concatenate([self.out[name].cpu().numpy().view(np.uint64) for name, _, _ in MSM_PHASES])
        return ProverShapeResult(outs, 1e3 * (t3 - t0), 1e3 * t_msm, 1e3 * t_ntt, 1e3 * t_h, 1e3 * t_arg, 1e3 * t_open)


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


def synthetic_proving_key(fill_scalars, field, k: int, extended_k: int, seed: int = 101) -> dict:
    """Extended-domain proving-key columns (fixed cosets, permutation cosets, l0 / l_last / l_active_row) and
    the five challenges, to be merged into synthetic_columns' dict for ProverShape(with_quotient=True)."""
    m = 1 << extended_k
    scalar_field_id = field.id
    out = {"pk_fixed": np.stack([fill_scalars(scalar_field_id, "uniform", m, seed + i) for i in range(N_FIXED)]),
           "pk_sigma": np.stack([fill_scalars(scalar_field_id, "uniform", m, seed + 20 + i) for i in range(N_SIGMA)]),
           "pk_l": np.stack([fill_scalars(scalar_field_id, "uniform", m, seed + 30 + i) for i in range(3)])}
    n = 1 << k
    tsize = min(n // 2, 1 << 12)                              # inside the usable rows
    tvals = field.encode_many(list(range(tsize)))            # a range table 0 .. 2^12 - 1, like the circuit's RangeChip tables
    out["lookup_table"] = np.concatenate([tvals, np.repeat(tvals[:1], n - tsize, axis=0)])
    rng = np.random.default_rng(seed)
    out["lookup_inputs"] = np.stack([tvals[rng.integers(0, tsize, size=n)] for _ in range(N_LOOKUPS)])
    ch = field.decode_many(fill_scalars(scalar_field_id, "uniform", 5, seed + 40))
    out["challenges"] = dict(zip(("theta", "beta", "gamma", "y", "delta"), ch))
    return out
