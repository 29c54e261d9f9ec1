"""Test infrastructure: `create_proof` driven from Python over the FINE-GRAINED C-ABI entry points (dehalo_msm_device_affine, dehalo_lagrange_to_coeff_device,
dehalo_coset_ntt_form_device, dehalo_permute_expression_pair_*_device, dehalo_grand_product_batch_device, dehalo_graph_evaluate_batch_device, dehalo_kate_division_*,
...) -- the data flow of halo2_proofs::plonk::create_proof over KZG / GWC [UPSTREAM halo2_proofs/src/plonk/prover.rs, plonk/{lookup,permutation,vanishing}/prover.rs,
poly/kzg/multiopen/gwc/prover.rs @ v2023_04_20] with every column resident in HBM and the host hashing the transcript.

The product's prover is `dehalo_create_proof` (csrc/prover.hip, native.Prover): ONE call.  This file exists so that the GPU suite can show that the ~45 `*_device` entry
points a Rust maintainer could bind one by one compose into the same proof bytes as the CPU restatement's and the one-call prover's
(tests/test_proof.py, tests/test_witness.py); it was `dehalo2_amd/prover.py`'s `Prover` until round 4.  Nothing in the package, bench.py or smoke() imports it.

Given the witness (advice columns), what is committed is what was computed:
  advice + blinding rows -> commit -> theta -> theta-compressed lookup expressions -> permute_expression_pair ->
  commit a', s' -> beta, gamma -> permutation and lookup grand products -> commit z -> random polynomial -> y ->
  lagrange_to_coeff, coeff_to_extended, evaluate_h, division by the vanishing polynomial, extended_to_coeff,
  split -> commit h pieces -> x -> evaluations -> v -> per opening point: sum v^i poly_i, kate_division, commit.

Field values are canonical ints on the host side of this file and 4 x u64 Montgomery limbs on the device.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass, field as dc_field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as _entry  # noqa: E402

_entry.load_package()
from dehalo2_amd import evaluation as ev  # noqa: E402
from dehalo2_amd import plonk  # noqa: E402
from dehalo2_amd.keygen import ParamsKZG, ProvingKey, array_to_ints, decode_points, delta_of, omega_powers_device, to_device, to_host  # noqa: E402
from dehalo2_amd.prover import OsRng, SeededRng, proof_commitments, proof_layout  # noqa: E402,F401
from dehalo2_amd.transcript import Blake2bWrite  # noqa: E402


@dataclass
class ProofTimings:
    phases_ms: Dict[str, float] = dc_field(default_factory=dict)
    total_ms: float = 0.0
    fine: bool = False                                     # also record host timestamps inside the phases (no extra syncs)
    ticks: List[Tuple[str, float]] = dc_field(default_factory=list)
    msm_shapes: List[dict] = dc_field(default_factory=list)   # fine: Context.msm_last_shape() after each commitment phase

    def tick(self, label: str):
        if self.fine:
            self.ticks.append((label, time.perf_counter()))


def rotate_omega(domain, x: int, rot: int) -> int:
    p = domain.field.p
    return x * pow(domain.omega if rot >= 0 else domain.omega_inv, abs(rot), p) % p


class Prover:
    """Owns the device buffers of one proof for a given proving key (allocated once, reused by every create_proof)."""

    def __init__(self, params: ParamsKZG, pk: ProvingKey, ctx=None, side_ctx=None):
        """`ctx`: the context (stream + workspace) this prover runs on; default the key's.  Several provers over one key, each on
        its own context, may run concurrently from different threads (batch proving): the SRS tables, the key's columns and
        the compiled programs are shared, the proof buffers are per prover.
        `side_ctx`: a second context of the same device.  Work that no transcript challenge waits for -- lagrange_to_coeff and
        coeff_to_extended of a phase's columns, the random polynomial's commitment -- is then queued there as soon as its inputs
        exist and runs beside the main context's commitment phases (ordered by events), instead of after y."""
        self.ctx = ctx if ctx is not None else pk.ctx
        self.side = side_ctx
        self._timings = None
        self._plan = None
        self._rinv = pow(1 << 256, -1, pk.vk.curve.scalar.p)
        with self.ctx.torch_stream():      # torch's copies and fills go on the context's stream, ordered with the kernels
            self._init(params, pk)

    def _init(self, params: ParamsKZG, pk: ProvingKey):
        import torch

        self.torch = torch
        self.params, self.pk = params, pk
        self.cs, self.domain, self.curve = pk.vk.cs, pk.domain, pk.vk.curve
        self.f = self.curve.scalar
        cs, d = self.cs, self.domain
        self.n, self.m, self.k, self.ek = d.n, d.extended_len(), d.k, d.extended_k
        self.bf = cs.blinding_factors()
        self.u = self.n - (self.bf + 1)                         # usable rows
        self.A, self.L, self.S = cs.num_advice, len(cs.lookups), cs.num_permutation_sets()
        self.I = cs.num_instance
        A, L, S, n, m = self.A, self.L, self.S, self.n, self.m
        z = lambda *shape: torch.zeros(shape, dtype=torch.int64, device="cuda")
        # committed columns, one contiguous block: [advice | permuted (input_0, table_0, input_1, ...) | perm z | lookup z | random]
        self.NC = A + 2 * L + S + L + 1
        self.cols = z(self.NC, n, 4)
        self.polys = z(self.NC, n, 4) if self.side is not None else self.cols    # coefficient forms (in place without a side context)
        self.o_adv, self.o_perm, self.o_pz, self.o_lz, self.o_rand = 0, A, A + 2 * L, A + 2 * L + S, A + 2 * L + S + L
        self.instance = z(max(self.I, 1), n, 4)
        self.compressed = z(max(2 * L, 1), n, 4)                # theta-compressed (input_l, table_l)
        self.num = z(S + L, n, 4)
        self.den = z(S + L, n, 4)
        self.ext = z(self.NC - 1 + self.I, m, 4)                # cosets of every committed column but the random one, then the instance columns
        self.h = z(m, 4)
        self.table_value = z(max(L, 1), m, 4)                   # one column per lookup: the lookups' terms are folded in ONE pass
        self.hfold = z(n, 4)
        self.qbuf = z(4, n, 4)                                  # per opening point: the batched polynomial
        self.wbuf = z(8, n, 4)                                  # ... and its quotient (last coefficient zero)
        self.jac_side, self.aff_side = z(1, 12), z(1, 8)
        self.jac = z(max(self.NC, 8), 12)
        self.aff = z(max(self.NC, 8), 8)
        self.evals = z(64 + 4 * (self.NC + cs.num_fixed + len(cs.permutation_columns) + 8), 4)
        # every blinding value of a proof but the random polynomial: drawn at the start (program order), uploaded once, copied into
        # the columns on the device phase by phase (a pinned torch staging buffer is not used: torch's host allocator would touch
        # the context's stream again when the buffer is freed, possibly after the context is gone)
        rows = self.n - self.u
        self.blind_counts = (A * rows, A, L * (2 * rows + 2), (S + L) * (self.bf + 1))
        self.blind_dev = z(max(1, sum(self.blind_counts)), 4)
        self.omega_col = omega_powers_device(self.ctx, d)
        e = self.f.encode
        self._c = dict(omega_inv=e(d.omega_inv), ifft=e(d.ifft_divisor), ext_omega=e(d.extended_omega), ext_omega_inv=e(d.extended_omega_inv),
                       ext_ifft=e(d.extended_ifft_divisor), zeta=e(d.g_coset))
        # t(X)^-1 on the coset: 2^(extended_k - k) values (EvaluationDomain::new)
        p = self.f.p
        orig, step = pow(d.g_coset, n, p), pow(d.extended_omega, n, p)
        self.t_inv = self.f.encode_many([pow((orig * pow(step, i, p) - 1) % p, -1, p) for i in range(1 << (self.ek - self.k))])
        # permutation product programs: per set, denominator prod(col + beta sigma + gamma) and numerator prod(col + delta^j beta omega^i + gamma)
        self.perm_graphs = self._permutation_graphs()
        self.lookup_product_graphs = self._lookup_product_graphs()
        self.ctx.synchronize()

    # ---- programs for the grand products (run over the n rows of the original domain) ----
    def _permutation_graphs(self):
        """advice slots: the circuit's advice columns; fixed slots: [circuit fixed..., sigma_0.., omega column]; instance: instance columns;
        challenges: delta^j * beta per permutation column."""
        cs, p, out = self.cs, self.f.p, []
        chunk = cs.permutation_chunk_len()
        nf, npc = cs.num_fixed, len(cs.permutation_columns)
        kind = {plonk.ADVICE: ev.ADVICE, plonk.FIXED: ev.FIXED, plonk.INSTANCE: ev.INSTANCE}
        for s in range(self.S):
            gd, gn = ev.GraphEvaluator(), ev.GraphEvaluator()
            dacc = nacc = None
            for j in range(s * chunk, min((s + 1) * chunk, npc)):
                ck, ci = cs.permutation_columns[j]
                for g, is_den in ((gd, True), (gn, False)):
                    col = g.column(kind[ck], ci)
                    if is_den:
                        t = g.add_calculation(ev.MUL, (ev.BETA, 0, 0), g.column(ev.FIXED, nf + j))
                    else:
                        t = g.add_calculation(ev.MUL, (ev.CHALLENGE, j, 0), g.column(ev.FIXED, nf + npc))
                    t = g.add_calculation(ev.ADD, g.add_calculation(ev.ADD, col, t), (ev.GAMMA, 0, 0))
                    if is_den:
                        dacc = t if dacc is None else g.add_calculation(ev.MUL, dacc, t)
                    else:
                        nacc = t if nacc is None else g.add_calculation(ev.MUL, nacc, t)
            gd.add_calculation(ev.STORE, dacc)
            gn.add_calculation(ev.STORE, nacc)
            out.append((gd.compile(self.ctx, self.f), gn.compile(self.ctx, self.f)))
        return out

    def _lookup_product_graphs(self):
        """advice slots: [compressed_input, compressed_table, permuted_input, permuted_table]."""
        gd, gn = ev.GraphEvaluator(), ev.GraphEvaluator()
        gd.add_calculation(ev.MUL, gd.add_calculation(ev.ADD, gd.column(ev.ADVICE, 2), (ev.BETA, 0, 0)), gd.add_calculation(ev.ADD, gd.column(ev.ADVICE, 3), (ev.GAMMA, 0, 0)))
        gn.add_calculation(ev.MUL, gn.add_calculation(ev.ADD, gn.column(ev.ADVICE, 0), (ev.BETA, 0, 0)), gn.add_calculation(ev.ADD, gn.column(ev.ADVICE, 1), (ev.GAMMA, 0, 0)))
        return gd.compile(self.ctx, self.f), gn.compile(self.ctx, self.f)

    # ---- helpers ----
    def _commit(self, transcript: Blake2bWrite, first: int, count: int, lagrange: bool, src=None, before_sync=None):
        """commit `count` consecutive columns, normalise, absorb: the transcript needs affine points on the host.
        `before_sync()` runs once the launches are queued, just before this thread blocks on them (work for the side context,
        the helper thread's start: nothing the commitment waits for)."""
        t = self.cols if src is None else src
        self.params.commit_affine_device(t[first].data_ptr(), count, self.aff.data_ptr(), lagrange, ctx=self.ctx)
        tk = self._tick
        tk("commit queued")
        if before_sync is not None:
            before_sync()
        host = self.ctx.download(self.aff.data_ptr(), count, 8)      # waits for the commitment, then copies: one call
        tk("points on host")
        if self._timings is not None and self._timings.fine:
            self._timings.msm_shapes.append(dict(self.ctx.msm_last_shape(), columns=count))
        pts = decode_points(self.curve, host)
        for P in pts:
            transcript.write_point(P)
        tk("points in transcript")
        return pts

    def _draw_blinds(self, rng):
        """advice blinding rows (column after column), the advice commitments' blinds (unused by KZG), per lookup (bf + 1 rows
        for the permuted input, bf + 1 for the permuted table, two unused blinds), per grand product (bf rows + one unused
        blind): upstream's draws up to the random polynomial, in its order; one upload."""
        total = sum(self.blind_counts)
        if total == 0:
            return
        self.blind_dev[:total].copy_(to_device(rng.scalars(total)))

    def _blind_slice(self, which: int):
        off = sum(self.blind_counts[:which])
        return self.blind_dev[off:off + self.blind_counts[which]]

    def _opening_plan(self):
        """What depends on the circuit only: the opened polynomials' device pointers (one multi-point evaluation), where each
        evaluation lands (evals[rotation index * polys + poly]), the order in which the transcript takes them
        [UPSTREAM plonk/prover.rs: advice, fixed, vanishing random_eval, permutation (sigma; products), lookups] and the opening
        queries grouped by point in order of first appearance [UPSTREAM permutation::Constructed::open, lookup::Evaluated::open,
        pk.permutation.open, vanishing::Evaluated::open; poly/kzg/multiopen/gwc/prover.rs]."""
        cs, pk, polys = self.cs, self.pk, self.polys
        bf, S, L, n, m = self.bf, self.S, self.L, self.n, self.m
        pieces = self.domain.quotient_poly_degree
        rots = sorted({r for _, r in cs.advice_queries} | {r for _, r in cs.fixed_queries} | {0, 1, -1, -(bf + 1)})
        if len(rots) > 4:
            raise ValueError("more than four distinct opening rotations")
        nfix, npc = cs.num_fixed, len(cs.permutation_columns)
        hp_ptrs = self._ptrs(self.h.view(m // n, n, 4), 0, pieces)
        plist = self._ptrs(polys, 0, self.NC) + self._ptrs(pk.fixed_polys) + self._ptrs(pk.perm_polys) + hp_ptrs
        ntot = len(plist)
        base = {"cols": 0, "fixed": self.NC, "sigma": self.NC + nfix, "hpiece": self.NC + nfix + npc}
        idx = lambda name, col, r: rots.index(r) * ntot + base[name] + col
        last = -(bf + 1)
        write: List[int] = []
        for col, r in cs.advice_queries:
            write.append(idx("cols", self.o_adv + col, r))
        for col, r in cs.fixed_queries:
            write.append(idx("fixed", col, r))
        write.append(idx("cols", self.o_rand, 0))                               # vanishing: random_eval
        for j in range(npc):                                                     # pk.permutation.evaluate: sigma(x)
            write.append(idx("sigma", j, 0))
        for s in range(S):                                                       # permutation products
            write.append(idx("cols", self.o_pz + s, 0))
            write.append(idx("cols", self.o_pz + s, 1))
            if s != S - 1:
                write.append(idx("cols", self.o_pz + s, last))
        for l in range(L):                                                       # lookups
            zc, ai, ti = self.o_lz + l, self.o_perm + 2 * l, self.o_perm + 2 * l + 1
            for colr in ((zc, 0), (zc, 1), (ai, 0), (ai, -1), (ti, 0)):
                write.append(idx("cols", *colr))
        # queries (point rotation, device polynomial, index of its evaluation; -1: the folded quotient's), in upstream's order
        col_ptr = self._ptrs(polys, 0, self.NC)
        fixed_ptr, sigma_ptr = self._ptrs(pk.fixed_polys), self._ptrs(pk.perm_polys)
        Q: List[Tuple[int, int, int]] = []
        for col, r in cs.advice_queries:
            Q.append((r, col_ptr[self.o_adv + col], idx("cols", self.o_adv + col, r)))
        for s in range(S):                                                       # permutation::Constructed::open
            zc = self.o_pz + s
            Q.append((0, col_ptr[zc], idx("cols", zc, 0)))
            Q.append((1, col_ptr[zc], idx("cols", zc, 1)))
        for s in reversed(range(S - 1)):                                         # ... x_last: sets.iter().rev().skip(1)
            zc = self.o_pz + s
            Q.append((last, col_ptr[zc], idx("cols", zc, last)))
        for l in range(L):                                                       # lookup::Evaluated::open
            zc, ai, ti = self.o_lz + l, self.o_perm + 2 * l, self.o_perm + 2 * l + 1
            for colr in ((zc, 0), (ai, 0), (ti, 0), (ai, -1), (zc, 1)):
                Q.append((colr[1], col_ptr[colr[0]], idx("cols", *colr)))
        for col, r in cs.fixed_queries:
            Q.append((r, fixed_ptr[col], idx("fixed", col, r)))
        for j in range(npc):                                                     # pk.permutation.open
            Q.append((0, sigma_ptr[j], idx("sigma", j, 0)))
        Q.append((0, self.hfold.data_ptr(), -1))                                 # vanishing::Evaluated::open
        Q.append((0, col_ptr[self.o_rand], idx("cols", self.o_rand, 0)))
        groups: List[Tuple[int, List[int], List[int]]] = []
        for r, ptr, i in Q:
            for g in groups:
                if g[0] == r:
                    g[1].append(ptr); g[2].append(i)
                    break
            else:
                groups.append((r, [ptr], [i]))
        if len(groups) > self.qbuf.shape[0]:
            raise ValueError("more opening points than the prover's buffers hold")
        need = sorted(set(write) | {i for _, _, i in Q if i >= 0} | {idx("hpiece", j, 0) for j in range(pieces)})
        self._plan = (rots, plist, write, groups, hp_ptrs, idx("hpiece", 0, 0), len(rots) * ntot, need)
        return self._plan

    def _tick(self, label: str):
        if self._timings is not None:
            self._timings.tick(label)

    def _ptrs(self, t, first=0, count=None):
        """device pointers of t[first], t[first + 1], ... (pointer arithmetic: indexing a tensor costs a microsecond per row)"""
        count = t.shape[0] - first if count is None else count
        if count <= 0:
            return []
        base, pitch = t.data_ptr(), t.stride(0) * t.element_size()
        return [base + (first + i) * pitch for i in range(count)]

    # ---- the proof ----
    def create_proof(self, advice, instances: Sequence[Sequence[int]], rng, transcript: Blake2bWrite,
                     timings: Optional[ProofTimings] = None):
        """`rng`: None = operating-system entropy (OsRng, what the reference passes); SeededRng for reproducible test proofs."""
        self._timings = timings
        if rng is None:
            rng = OsRng(self.f.p)
        self._prefetch = None
        try:
            with self.ctx.torch_stream():
                return self._create_proof(advice, instances, rng, transcript, timings)
        finally:
            # a helper thread that was started must have finished before the caller may reuse or close the contexts
            t = self._prefetch
            if t is not None and t.is_alive():
                t.join()

    def _create_proof(self, advice, instances: Sequence[Sequence[int]], rng, transcript: Blake2bWrite,
                      timings: Optional[ProofTimings] = None):
        """advice: (num_advice, n, 4) u64 Montgomery (host array or device tensor); instances: one list of canonical ints per
        instance column (the reference passes &[&[&[]]]: none).  Appends the proof to `transcript`."""
        torch, ctx, f, cs, d, pk, c = self.torch, self.ctx, self.f, self.cs, self.domain, self.pk, self._c
        n, m, k, ek, u, bf, A, L, S = self.n, self.m, self.k, self.ek, self.u, self.bf, self.A, self.L, self.S
        p, enc = f.p, f.encode
        cols, fid = self.cols, f.id
        rot_scale = m // n
        t_phase = time.perf_counter()
        t_start = t_phase

        def mark(name):
            nonlocal t_phase
            if timings is not None and timings.fine:
                timings.tick("== " + name)
            elif timings is not None:
                ctx.synchronize()
                now = time.perf_counter()
                timings.phases_ms[name] = timings.phases_ms.get(name, 0.0) + 1e3 * (now - t_phase)
                t_phase = now

        # the one large random draw of the proof (the vanishing argument's random polynomial, n scalars, drawn after every
        # blinding value) is produced and uploaded by a helper thread while the earlier phases run
        draws_before = A * (n - u) + A + L * (2 * (n - u) + 2) + (S + L) * (bf + 1)
        prefetch = None
        if hasattr(rng, "fork"):
            import threading
            box = {}

            def _draw(r=rng.fork(draws_before)):
                try:
                    if self.side is None:
                        with ctx.torch_stream():
                            box["poly"] = to_device(r.scalars(n))
                        return
                    # with a side context the random polynomial is also COMMITTED there, long before its phase
                    with self.side.torch_stream():
                        self.polys[self.o_rand].copy_(to_device(r.scalars(n)))
                        self.params.commit_affine_device(self.polys[self.o_rand].data_ptr(), 1, self.aff_side.data_ptr(), False, ctx=self.side)
                        self.side.synchronize()
                        box["point"] = decode_points(self.curve, to_host(self.aff_side[:1]))[0]
                except BaseException as exc:      # noqa: BLE001 -- re-raised on the proving thread after join()
                    box["error"] = exc
            prefetch = threading.Thread(target=_draw)      # started when the advice commitments are queued (below): the host is idle then
            self._prefetch = prefetch

        def columns_ready():
            """an event after everything queued so far on the main context (a phase's columns and their blinding rows)"""
            if self.side is None:
                return None
            e = torch.cuda.Event()
            e.record(ctx.torch_stream_obj())
            return e

        def side_ntt(first, count, e):
            """polys[first : first + count] = lagrange_to_coeff(cols[...]), ext[...] = coeff_to_extended(...) on the side context, once
            the event `e` (recorded BEFORE the phase's commitment was queued) has passed: the launches themselves are made after the
            commitment's, while this thread would otherwise only wait."""
            sb = self.side.torch_stream_obj()
            sb.wait_event(e)
            with torch.cuda.stream(sb):
                self.polys[first:first + count].copy_(cols[first:first + count])
            self.side.intt_scaled_device(fid, self.polys[first].data_ptr(), k, c["omega_inv"], c["ifft"], count, 0)
            self.side.coset_ntt_form_device(fid, self.polys[first].data_ptr(), k, self.ext[first].data_ptr(), ek, c["ext_omega"], c["zeta"], count, ev.FORM_OUT_INTERNAL, 0)
        self._draw_blinds(rng)                                   # (first: the upload blocks this thread until the stream has caught up -- nothing is queued yet)
        transcript.common_scalar(pk.vk.transcript_repr)          # vk.hash_into
        # -- instance columns: values into the transcript (KZG: QUERY_INSTANCE = false), polynomials on the device
        if len(instances) != self.I:
            raise ValueError("instances.len() != num_instance_columns")       # upstream: Error::InvalidInstances
        self.instance.zero_()
        for i, vals in enumerate(instances):
            if len(vals) > u:
                raise ValueError("instance column too long")                   # upstream: Error::InstanceTooLarge
            for v in vals:
                transcript.common_scalar(v)
            if len(vals):
                self.instance[i, :len(vals)] = to_device(f.encode_many(list(vals)))
        instance_values = self.instance.clone() if self.I else self.instance
        nco = self.NC - 1
        inst_ready = columns_ready() if self.I and self.side is not None else None

        def side_instance():     # instance polynomials and their cosets depend on the inputs only: side context (launched behind the advice commitment's launches)
            self.side.torch_stream_obj().wait_event(inst_ready)
            self.side.intt_scaled_device(fid, self.instance.data_ptr(), k, c["omega_inv"], c["ifft"], self.I, 0)
            self.side.coset_ntt_form_device(fid, self.instance.data_ptr(), k, self.ext[nco].data_ptr(), ek, c["ext_omega"], c["zeta"], self.I, ev.FORM_OUT_INTERNAL, 0)
        if self.I and self.side is None:
            ctx.intt_scaled_device(fid, self.instance.data_ptr(), k, c["omega_inv"], c["ifft"], self.I, 0)      # instance polys

        # -- advice: witness, blinding rows, commitments
        adv = advice if torch.is_tensor(advice) else to_device(np.ascontiguousarray(advice, dtype=np.uint64).reshape(A, n, 4))
        if tuple(adv.shape) != (A, n, 4):
            raise ValueError("advice must be num_advice x n x 4")
        cols[self.o_adv:self.o_adv + A].copy_(adv)
        cols[self.o_adv:self.o_adv + A, u:] = self._blind_slice(0).view(A, n - u, 4)
        ready = columns_ready()

        # one gate polynomial: Horner(0, [g], y) = g does not depend on y, so the custom-gate pass of evaluate_h needs the advice (and
        # instance) cosets only -- queued on the side context right behind them, ~7 ms before y exists
        gates_early = self.side is not None and len(cs.gates) == 1
        FF = ev.COLUMNS_INTERNAL | ev.VALUES_INTERNAL
        rot_scale_h = m // n

        def after_advice_queued():
            if inst_ready is not None:
                side_instance()
            if self.side is not None:
                side_ntt(self.o_adv, A, ready)
            if gates_early:
                pk.custom_gates.evaluate_device(self._ptrs(pk.fixed_cosets), self._ptrs(self.ext, self.o_adv, A), self._ptrs(self.ext, nco, self.I), [], None, None,
                                                None, 0, ek, rot_scale_h, 0, self.h.data_ptr(), 0, FF, self.side)
            if prefetch is not None:
                prefetch.start()
        self._commit(transcript, self.o_adv, A, True, before_sync=after_advice_queued)
        mark("advice")
        theta = transcript.squeeze_challenge_scalar()
        self._tick("theta")

        # -- lookups: compress, permute, blind, commit
        fixed_v = self._ptrs(pk.fixed_values)
        adv_v = self._ptrs(cols, self.o_adv, A)
        inst_v = self._ptrs(instance_values, 0, self.I)
        if L:
            # every lookup's compressed (input, table) pair, interleaved like the output buffer: one call
            graphs = [g.handle for pair in pk.compress_graphs for g in pair]
            ctx.graph_evaluate_batch_device(graphs, fixed_v, adv_v, inst_v, None, None, None, enc(theta), None, k, 1, self._ptrs(self.compressed, 0, 2 * L))
            # permuted columns are interleaved (input_l, table_l) with a stride of two columns
            base = cols[self.o_perm].data_ptr()
            self._tick("compress queued")
            # the blinding rows [u, n) first: the permutation writes rows [0, u) only, and it ends with a read-back -- whatever is
            # queued before it does not wait for the host afterwards
            rows = n - u                                         # (input_0, table_0, input_1, ...): bf + 1 rows each
            cols[self.o_perm:self.o_perm + 2 * L, u:] = self._blind_slice(2).view(L, 2 * rows + 2, 4)[:, :2 * rows].reshape(2 * L, rows, 4)
            self._tick("blinds copied")
            ctx.permute_expression_pair_batch_device(fid, self.compressed[0].data_ptr(), self.compressed[1].data_ptr(), u, L, 2 * n, base, base + 32 * n, 0)
            self._tick("permute returned")
            ready = columns_ready()
            self._commit(transcript, self.o_perm, 2 * L, True, before_sync=(lambda: side_ntt(self.o_perm, 2 * L, ready)) if self.side is not None else None)
        mark("lookup_permuted")
        beta = transcript.squeeze_challenge_scalar()
        gamma = transcript.squeeze_challenge_scalar()
        self._tick("beta gamma")

        # -- grand products: permutation sets, then lookups; one batched inversion
        npc = len(cs.permutation_columns)
        delta, chal, dj = delta_of(f), [], beta
        for _ in range(npc):
            chal.append(dj)
            dj = dj * delta % p
        perm_fixed = fixed_v + self._ptrs(pk.perm_values) + [self.omega_col.data_ptr()]
        chal_e, beta_e, gamma_e, none = f.encode_many(chal), enc(beta), enc(gamma), []      # encoded once, not once per graph
        den_p, num_p = self._ptrs(self.den), self._ptrs(self.num)
        if S:                                                    # every set's denominator and numerator programs: same inputs, one call
            ctx.graph_evaluate_batch_device([g.handle for pair in self.perm_graphs for g in pair], perm_fixed, adv_v, inst_v, chal_e, beta_e, gamma_e, None, None,
                                            k, 1, [q for s in range(S) for q in (den_p[s], num_p[s])])
        comp_p, perm_p = self._ptrs(self.compressed), self._ptrs(cols, self.o_perm, 2 * L)
        for l in range(L):
            gd, gn = self.lookup_product_graphs
            four = [comp_p[2 * l], comp_p[2 * l + 1], perm_p[2 * l], perm_p[2 * l + 1]]
            gd.evaluate_device(none, four, none, none, beta_e, gamma_e, None, None, k, 1, 0, den_p[S + l], 0, 0, ctx)
            gn.evaluate_device(none, four, none, none, beta_e, gamma_e, None, None, k, 1, 0, num_p[S + l], 0, 0, ctx)
        self._tick("product graphs queued")
        if S + L:
            ctx.grand_product_batch_device(fid, self.num.data_ptr(), self.den.data_ptr(), n, S + L, n, cols[self.o_pz].data_ptr(), 0)
        for s in range(1, S):                                    # z_s starts where z_{s-1} ended: z = vec![last_z]
            ctx.scale_device(fid, cols[self.o_pz + s].data_ptr(), n, None, cols[self.o_pz + s - 1][u].data_ptr(), 0)
        if S + L:                                                # per column: bf blinding rows (n - bf .. n), then the (unused) commitment blind
            cols[self.o_pz:self.o_pz + S + L, n - bf:] = self._blind_slice(3).view(S + L, bf + 1, 4)[:, :bf]
        if S + L:
            ready = columns_ready()

            def after_products_queued():
                side_ntt(self.o_pz, S + L, ready)
                # the lookups' (compressed input + beta)(compressed table + gamma) over the extended domain need theta, beta, gamma and
                # the advice / fixed cosets: all there -- on the side context, beside the products' commitment, instead of after y
                if L:
                    be, ge, te, none = enc(beta), enc(gamma), enc(theta), []
                    fc, ac, ic, tvp = self._ptrs(pk.fixed_cosets), self._ptrs(self.ext, self.o_adv, A), self._ptrs(self.ext, nco, self.I), self._ptrs(self.table_value)
                    self.side.graph_evaluate_batch_device([g.handle for g in pk.lookup_graphs], fc, ac, ic, None, be, ge, te, None, ek, rot_scale_h, tvp, 0, FF)
            self._commit(transcript, self.o_pz, S + L, True, before_sync=after_products_queued if self.side is not None else None)
        mark("grand_products")

        # -- vanishing argument: a random polynomial
        polys = self.polys
        if prefetch is not None:
            prefetch.join()
            if "error" in box:
                raise box["error"]
            rng.skip(n)
        if self.side is not None and prefetch is not None:
            rng.scalars(1)
            transcript.write_point(box["point"])                 # committed on the side context while the earlier phases ran
        else:
            cols[self.o_rand].copy_(box["poly"] if prefetch is not None else to_device(rng.scalars(n)))
            if self.side is not None:
                polys[self.o_rand].copy_(cols[self.o_rand])
            rng.scalars(1)
            self._commit(transcript, self.o_rand, 1, False)
        mark("random_poly")
        y = transcript.squeeze_challenge_scalar()
        self._tick("y")

        # -- coefficient forms and cosets of everything committed so far
        nco = self.NC - 1
        if self.side is None:
            ctx.intt_scaled_device(fid, cols.data_ptr(), k, c["omega_inv"], c["ifft"], nco, 0)
            ctx.coset_ntt_form_device(fid, cols.data_ptr(), k, self.ext.data_ptr(), ek, c["ext_omega"], c["zeta"], nco, ev.FORM_OUT_INTERNAL, 0)
        else:                                                    # queued phase by phase on the side context: wait for it
            e = torch.cuda.Event()
            e.record(self.side.torch_stream_obj())
            ctx.torch_stream_obj().wait_event(e)
        if self.I and self.side is None:
            ctx.coset_ntt_form_device(fid, self.instance.data_ptr(), k, self.ext[nco].data_ptr(), ek, c["ext_omega"], c["zeta"], self.I, ev.FORM_OUT_INTERNAL, 0)
        mark("ntt")

        # -- evaluate_h, / t(X), extended_to_coeff
        FF = ev.COLUMNS_INTERNAL | ev.VALUES_INTERNAL
        fixed_c, adv_c, inst_c = self._ptrs(pk.fixed_cosets), self._ptrs(self.ext, self.o_adv, A), self._ptrs(self.ext, nco, self.I)
        l0, l_last, l_active = (pk.l_ext[i].data_ptr() for i in range(3))
        if not gates_early:
            pk.custom_gates.evaluate_device(fixed_c, adv_c, inst_c, [], None, None, None, y, ek, rot_scale, 0, self.h.data_ptr(), 0, FF, ctx)
        if S:
            kindmap = {plonk.ADVICE: adv_c, plonk.FIXED: fixed_c, plonk.INSTANCE: inst_c}
            pcols = [kindmap[ck][ci] for ck, ci in cs.permutation_columns]
            ev.permutation_h_device(ctx, f, self._ptrs(self.ext, self.o_pz, S), pcols, self._ptrs(pk.perm_cosets), cs.permutation_chunk_len(), -(bf + 1), l0, l_last,
                                    l_active, beta, gamma, y, delta, d.g_coset, d.extended_omega, ek, rot_scale, self.h.data_ptr(), 0, FF)
        if L:
            be, ge, te, none = enc(beta), enc(gamma), enc(theta), []
            tv = self._ptrs(self.table_value)
            if self.side is None:
                for l in range(L):
                    pk.lookup_graphs[l].evaluate_device(fixed_c, adv_c, inst_c, none, be, ge, te, None, ek, rot_scale, 0, tv[l], 0, FF, ctx)
            zc, pc = self._ptrs(self.ext, self.o_lz, L), self._ptrs(self.ext, self.o_perm, 2 * L)
            for first in range(0, L, 8):
                ev.lookup_h_batch_device(ctx, f, [(zc[l], pc[2 * l], pc[2 * l + 1], tv[l]) for l in range(first, min(L, first + 8))], l0, l_last, l_active,
                                         beta, gamma, y, ek, rot_scale, self.h.data_ptr(), 0, FF)
        mark("evaluate_h")
        ctx.scale_device(fid, self.h.data_ptr(), m, self.t_inv, 0, 0)                                     # divide_by_vanishing_poly
        ctx.coset_intt_form_device(fid, self.h.data_ptr(), ek, c["ext_omega_inv"], c["ext_ifft"], c["zeta"], 1, ev.FORM_IN_INTERNAL, 0)
        pieces = d.quotient_poly_degree
        rng.scalars(pieces)                                      # h_blinds
        self._commit(transcript, 0, pieces, False, src=self.h.view(m // n, n, 4))
        mark("h_pieces")
        x = transcript.squeeze_challenge_scalar()
        xn = pow(x, n, p)
        self._tick("x")

        # -- evaluations, in upstream's order.  Every opened polynomial (committed columns, fixed, sigma, the quotient's pieces) at
        # every rotation in ONE call; which value goes where -- the transcript's order, the opening queries, their grouping by point --
        # depends on the circuit only and is laid out once per prover (_opening_plan)
        plan = self._plan if self._plan is not None else self._opening_plan()
        rots4, plist, write_idx, groups, hp_ptrs, hpiece0, count, need = plan
        point = {r: rotate_omega(d, x, r) for r in rots4}
        ctx.eval_polynomial_multi_device(fid, plist, n, f.encode_many([point[r] for r in rots4]), self.evals.data_ptr(), 0)
        # the folded quotient  h(X) = sum_i x^(n i) h_i(X)  (opened below; its value at x comes from the pieces' values: four products on
        # the host instead of two more launches in front of the read-back)
        xs, cur = [], 1
        for _ in range(pieces):
            xs.append(cur)
            cur = cur * xn % p
        ctx.lincomb_device(fid, hp_ptrs, f.encode_many(xs), n, self.hfold.data_ptr(), None, 0)
        mark("evaluations_launch")
        rinv = self._rinv
        raw, E = array_to_ints(ctx.download(self.evals.data_ptr(), count, 4)), [0] * count
        for i in need:                                           # (only what the transcript and the queries use)
            E[i] = raw[i] * rinv % p
        hfold_eval = sum(xs[i] * E[hpiece0 + i] for i in range(pieces)) % p
        for i in write_idx:
            transcript.write_scalar(E[i])
        mark("evaluations")

        # -- ProverGWC::create_proof: one witness polynomial per distinct point, in order of first appearance
        v = transcript.squeeze_challenge_scalar()
        self._tick("v")
        self.wbuf.zero_()
        for gi, (r, ptrs, idxs) in enumerate(groups):
            coefs, eval_batch, pw = [], 0, 1
            for i in idxs:
                coefs.append(pw)
                eval_batch += pw * (E[i] if i >= 0 else hfold_eval)
                pw = pw * v % p
            ctx.lincomb_device(fid, ptrs, f.encode_many(coefs), n, self.qbuf[gi].data_ptr(), enc(eval_batch % p), 0)
        self._tick("lincombs queued")
        order = [r for r, _, _ in groups]
        ctx.kate_division_batch_device(fid, self._ptrs(self.qbuf, 0, len(order)), n, f.encode_many([point[r] for r in order]), self._ptrs(self.wbuf, 0, len(order)), 0)
        self._commit(transcript, 0, len(order), False, src=self.wbuf)
        mark("openings")
        if timings is not None:
            timings.total_ms = 1e3 * (time.perf_counter() - t_start)
        return transcript


def create_proof(params: ParamsKZG, pk: ProvingKey, advice, instances, rng, transcript: Blake2bWrite) -> Blake2bWrite:
    """One-shot form (allocates the proof's device buffers for this call)."""
    return Prover(params, pk).create_proof(advice, instances, rng, transcript)
