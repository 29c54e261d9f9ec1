#!/usr/bin/env python3
"""bench.py -- one JSON line for BASELINE.json's metric on its configs[1]:
"Standalone 2^20 Pallas MSM + 2^20 pasta-Fp NTT on 1 MI355X".

A step = one pass of the hot path over one batch of synthetic input, inputs already resident
in HBM: one 2^20-term Pallas MSM (== best_multiexp / commit over a registered SRS) plus one
2^20-point NTT over pasta::Fp (== best_fft), then -- because the north star names it -- the
all-gather of the commitment vector across ranks (a no-op at N = 1).  One process per GPU;
N > 1 is launched by torch.distributed.run, units are independent per rank (weak scaling).

After the timed region (never inside it) rank 0 also (a) checks the 2^20 MSM and NTT results of the measured configuration
against the CPU port that serves as `cpu_baseline`, (b) makes REAL proofs with the device `create_proof` for the reference's two
circuit shapes -- delay_enc (k = 17, 5 lookups, degree 5) and pose_enc (k = 11) -- asserts them byte-identical to the CPU
restatement's proofs (whose time is the CPU figure beside them) and accepted by the verifier, and (c) proves a small batch per
GPU and all-gathers the commitments (configs[4]: `--proofs 64` on 8 GPUs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--no-cpu-baseline] [--proofs P] [--proof-k 17]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy ceiling
MSM_BYTES_PER_TERM = 96  # 64 B affine base + 32 B scalar
NTT_BYTES_PER_ELEM = 64  # read once + write once
# v_mad_u64_u32 per XYZZ mixed addition in k_msm_accum0's loop (ISA count) and the measured issue
# peak of that instruction on MI355X (profiles/r01_ubench_instruction_rates.txt)
MADS_PER_MIXED_ADD = {"pallas": 1224, "vesta": 1224, "bn254": 1467}
VMAD_PEAK_TMADS = 30.5
KERNEL_REV = "r04a"   # bumped whenever k_msm_accum0 changes: a PMC traffic figure measured on another revision is not reported
MADS_PER_FIELD_MUL = {"pasta_fp": 135, "pasta_fq": 135, "bn254_fr": 162, "bn254_fq": 162}   # f29_mul, ISA count


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-single-stream", action="store_true", help="skip the extra one-step-at-a-time pass (profiling: keeps per-kernel averages of the timed region clean)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="pallas")
    ap.add_argument("--ntt-field", default="pasta_fp")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "witness", "lookup"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --force-device rehearse N > 1 on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=-1)
    ap.add_argument("--proof-k", type=int, default=17, help="k of the delay_enc-shaped create_proof (0 = skip every proof section)")
    ap.add_argument("--proofs", type=int, default=-1, help="batch mode: total proofs dealt round-robin to the ranks (default 32 per GPU; 0 = skip)")
    ap.add_argument("--fixed-batch", type=int, default=64, help="batch mode: a second timed batch of exactly this many proofs whatever the number of GPUs (BASELINE configs[4]: 64; 0 = skip)")
    ap.add_argument("--proofs-inflight", type=int, default=4, help="batch mode: proofs in flight per GPU (one context + host thread each)")
    ap.add_argument("--no-verify", action="store_true", help="skip the pairing check of the proofs (the byte comparison with the oracle stays)")
    ap.add_argument("--acc-waves", type=int, default=0, help="dehalo_ctx_set_tuning msm_acc_waves with the whole-rounds rule (0 = library default: msm_acc_points)")
    ap.add_argument("--sort-block", type=int, default=0, choices=[0, 512, 1024], help="dehalo_ctx_set_tuning msm_sort_block on every context (0 = library default, 1024)")
    ap.add_argument("--preheat-s", type=float, default=1.0, help="untimed device work of the measured kind right before every warm-up + timed region (steps one at a time / "
                    "proofs), so that the region runs at the clocks of a busy prover instead of ramping up from idle after the host-side setup (0 = off)")
    ap.add_argument("--in-process", action="store_true", help="measure in this process: the default for one GPU (kept as a flag for the ranks torchrun starts)")
    ap.add_argument("--supervise", action="store_true", help="opt-in: measure in a fresh child process and start ONE more if that child is killed by a signal (supervise()); "
                    "by default a device fault ends the run with a non-zero exit code")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"), help="where the verbose record (every field, with its sentences) is written; "
                    "stdout carries the compact line only ('' = nowhere)")
    ap.add_argument("--inflight", type=int, default=4, help="independent steps in flight, each on its own HIP stream / workspace")
    return ap.parse_args()


PREHEAT_S = 1.0     # --preheat-s


def preheat(fn, sync=None, seconds=None):
    """Calls fn until PREHEAT_S (or `seconds`) seconds have passed (at least once).  The device clocks ramp up over ~0.2 s of load: 20 timed steps after
    5 / 60 / 200 warm-up steps measured 737-743 / 758 / 777 Mpoints/s on one box (gpurun_out/warmup_sweep.txt -> DESIGN.md 6)."""
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        if sync is not None and n % 8 == 0:
            sync()
        if time.perf_counter() - t0 >= (PREHEAT_S if seconds is None else seconds):
            break
    if sync is not None:
        sync()
    return n


def note(msg):
    """progress on stderr (the JSON line on stdout stays the only stdout output): which section a long run -- or a failed one -- was in"""
    print("[bench %.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def cpu_baseline(co, po, curve, field, log_n, bases, scalars, a):
    """The oracle's C restatement of best_multiexp + best_fft (same chunk-per-thread split as upstream's rayon code) on
    every core this process may use: a reported baseline, not the target.  It runs on the SAME bases / scalars / polynomial
    as rank 0's timed steps, so its results double as the parity check of the measured configuration.
    -> (json dict, MSM result as affine Montgomery limbs, NTT result)."""
    cores = host_cores()
    n = 1 << log_n
    omega = field.encode(po.FIELDS[field.name].omega(log_n))
    # the port parallelises with plain pthreads per call (one Pippenger per chunk, a thread per FFT task): on a many-core host a
    # smaller thread count can be the faster one, so both are timed and the better is reported with its thread count
    best = None
    msm_res = ntt_res = None
    for threads in sorted({min(cores, 256), min(cores, 64)}):
        reps, t_msm, t_ntt = 0, 0.0, 0.0
        t_start = time.time()
        while reps < 2 and (reps == 0 or time.time() - t_start < 10.0):
            t0 = time.time()
            msm_res = co.best_multiexp(curve.id, scalars, bases, threads)
            t1 = time.time()
            ntt_res = co.best_fft(field.id, a, omega, log_n, threads)
            t2 = time.time()
            t_msm += t1 - t0
            t_ntt += t2 - t1
            reps += 1
        rec = (t_msm / reps + t_ntt / reps, threads, reps, t_msm / reps, t_ntt / reps)
        if best is None or rec[0] < best[0]:
            best = rec
    step_s, threads, reps, msm_s, ntt_s = best
    return {
        "value": round(n / step_s / 1e6, 4), "unit": "Mpoints/s", "cores": threads, "host_cores": cores, "kind": "port",
        "sample": "%d x (2^%d-term MSM + 2^%d-point NTT) on rank 0's own inputs, oracle/oracle.c best_multiexp+best_fft, %d threads (the faster of 64 / all host threads)" % (reps, log_n, log_n, threads),
        "msm_ms": round(1e3 * msm_s, 2), "ntt_ms": round(1e3 * ntt_s, 2),
    }, co.to_affine(curve.id, msm_res), ntt_res


def by_k_numbers(pkg, po, co, ctx, curve, field, bases_by_k, threads):
    """The north star's reporting sentence: "throughput on synthetic witnesses at k in {14, 17, 20} ... as absolute numbers and as fraction of the HBM
    roofline" (BASELINE.json), with SURVEY.md 8(d)'s three scalar distributions and the NTT at n and at the extended size 4n.  One launch at a time on one
    context (HIP events on its stream around the sort / accumulate / reduce regions and the NTT passes), after a short preheat of the same launches; every
    timed result is then compared with the CPU port (oracle.c best_multiexp / best_fft) -- `ok`.
      msm_mpts[d]  = 2^k / (device ms of one MSM of 2^k terms over the resident table), d in u(niform) / w(itness-like) / l(ookup-like)
      msm_frac[d]  = 96 B x 2^k / that time / 8 TB/s      (64 B base + 32 B scalar per term)
      ntt_ms       = [2^k points, 2^(k+2) points];   ntt_frac = 64 B x N / that time / 8 TB/s"""
    import numpy as np
    import torch
    from dehalo2_amd import _lib
    out = []
    kids = (_lib.K_MSM_SORT, _lib.K_MSM_ACCUMULATE, _lib.K_MSM_REDUCE)
    for k in sorted(bases_by_k):
        n = 1 << k
        bases_h, bases = bases_by_k[k]
        d_out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
        rec = {"k": k, "msm_mpts": {}, "msm_frac": {}, "ntt_ms": [], "ntt_frac": [], "ok": True}
        reps = 24 if k <= 17 else 8
        for tag, dist in (("u", "uniform"), ("w", "witness"), ("l", "lookup")):
            sc_h = co.fill_scalars(curve.scalar.id, dist, n, 7000 + k)
            d_sc = ctx.upload(sc_h)
            run = lambda: ctx.msm_device(bases, d_sc.data_ptr(), n, 1, d_out.data_ptr(), 0)
            preheat(run, ctx.synchronize, 0.15)
            ctx.timing_reset(); ctx.timing_enable(True)
            for _ in range(reps):
                run()
            ctx.synchronize()
            ctx.timing_enable(False)
            ms = sum(ctx.timing_get(kid)[0] for kid in kids) / reps
            rec["msm_mpts"][tag] = round(n / ms / 1e3, 1)
            rec["msm_frac"][tag] = round(MSM_BYTES_PER_TERM * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            got = ctx.to_affine(curve.id, ctx.download_tensor(d_out))
            want = co.to_affine(curve.id, co.best_multiexp(curve.id, sc_h, bases_h, threads))
            rec["ok"] = rec["ok"] and bool(np.array_equal(np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)))
        for log_n in (k, k + 2):
            N = 1 << log_n
            a_h = co.fill_scalars(field.id, "uniform", N, 8000 + log_n)
            omega = field.encode(po.FIELDS[field.name].omega(log_n))
            d_a = ctx.upload(a_h)
            d_w = torch.empty_like(d_a)
            def run():
                d_w.copy_(d_a)
                ctx.ntt_device(field.id, d_w.data_ptr(), log_n, omega, 1, 0)
            with ctx.torch_stream():
                preheat(run, ctx.synchronize, 0.15)
                ctx.timing_reset(); ctx.timing_enable(True)
                for _ in range(reps):
                    run()
                ctx.synchronize()
                ctx.timing_enable(False)
            ms = ctx.timing_get(_lib.K_NTT_PASS)[0] / reps
            rec["ntt_ms"].append(round(ms, 4))
            rec["ntt_frac"].append(round(NTT_BYTES_PER_ELEM * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            want = co.best_fft(field.id, a_h, omega, log_n, threads)
            rec["ok"] = rec["ok"] and bool(np.array_equal(ctx.download_tensor(d_w), want))
            del d_a, d_w
        out.append(rec)
    return out


def secondary_numbers(pkg, co, ctx, curve, bases_h, scalars_h, log_n):
    """What the headline does not show: the cost of building the resident SRS table, the same MSM over a single-row table
    (no precomputed window multiples) and the literal drop-in `best_multiexp(coeffs, bases)` over host buffers that are not
    registered (upload + one-row table + MSM + download: PCIe-inclusive)."""
    import numpy as np
    import torch
    n = 1 << log_n
    out = {}
    t = time.perf_counter(); b = ctx.register_bases(curve.id, bases_h, 0, True); out["table_build_ms_precomputed"] = round(1e3 * (time.perf_counter() - t), 2)
    out["table_windows"], out["table_bytes"] = b.windows, b.windows * n * 64
    b.release()
    t = time.perf_counter(); b1 = ctx.register_bases(curve.id, bases_h, 0, False); out["table_build_ms_single_row"] = round(1e3 * (time.perf_counter() - t), 2)
    d_s = ctx.upload(scalars_h)
    d_o = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t = time.perf_counter(); ctx.msm_device(b1, d_s.data_ptr(), n, 1, d_o.data_ptr(), 0); ctx.synchronize(); ts.append(time.perf_counter() - t)
    out["msm_single_row_device_ms"] = round(1e3 * min(ts[1:]), 3)
    b1.release()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); ctx.best_multiexp(curve.id, scalars_h, bases_h); ts.append(time.perf_counter() - t)
    out["best_multiexp_host_buffers_ms"] = round(1e3 * min(ts[1:]), 3)
    out["best_multiexp_host_buffers_mpoints_per_s"] = round(n / min(ts[1:]) / 1e6, 1)
    return out


def real_witness(p, k, circuit):
    """The reference's circuits with real witnesses (dehalo2_amd/witness.py, halo2wrong's row layout).  `circuit`:
      "delay_enc"  DelayEncryptCircuit over a 2048-bit modulus with as many exponent bits as 2^k rows hold (15 bits at k = 17: the
                   north-star shape, benches/README.md:59-60), the all-zero message the reference's benches encrypt (benches/delay_enc.rs:68-70);
      "mod_pow"    benches/mod_pow.rs's RSACircuit: 2048-bit modulus, EXP_LIMB_BITS = 5 (:47-49), at the K = 17 the bench runs (:258);
      "pose_enc"   PoseidonEncCircuit (benches/pose_enc.rs, K = 11).
    Returns (circuit, description, rows the reference publishes for this configuration or None)."""
    import random
    from dehalo2_amd import circuits, witness as W
    rnd = random.Random(0x64656C6179)
    n_big = rnd.getrandbits(2048) | (1 << 2047) | 1
    x = rnd.getrandbits(2040)
    message = [0, 0]
    if circuit == "pose_enc":
        key = [rnd.getrandbits(250), rnd.getrandbits(250)]
        circ, info = W.pose_enc_witness(p, k, key, message)
        circ.native_spec = dict(circuit=2, k=k, key=key, message=message)
        return circ, "PoseidonEncCircuit (benches/pose_enc.rs), |msg| = 2: %d rows (the reference publishes the last row's index, 1,450: benches/README.md:98)" % info.total_rows, 1450
    if circuit == "mod_pow":
        e = rnd.getrandbits(5) | (1 << 4)
        circ, info = W.mod_pow_witness(p, k, n_big, e, x, 5)
        assert info.rsa_result == pow(x, e, n_big)
        circ.native_spec = dict(circuit=1, k=k, n_big=n_big, e=e, x=x, exp_bits=5)
        return circ, "RSACircuit (benches/mod_pow.rs), 2048-bit modulus, 5-bit exponent: %d rows (published: 41,766, benches/README.md:80)" % info.total_rows, 41766
    kk = min(k, 17)
    tail = W.hash_region_rows() + W.cipher_region_rows(len(message), True)
    bits = max(b for b in range(1, 16) if W.rsa_region_rows(b) + tail + 1 <= (1 << kk) - 6)
    e = rnd.getrandbits(bits) | (1 << (bits - 1))
    circ, info = W.delay_enc_witness(p, kk, n_big, e, x, bits, message)
    assert info.rsa_result == pow(x, e, n_big) and info.total_rows == W.rsa_region_rows(bits) + tail
    desc = "DelayEncryptCircuit (src/lib.rs), 2048-bit modulus, %d-bit exponent: %d RSA rows + %d hash / cipher rows" % (bits, info.rsa_rows, info.total_rows - info.rsa_rows)
    published = None
    if bits == 15:      # the north-star shape: say what the row count is and is not
        published = 130248
        desc += (" = %d rows, halo2wrong's layout of the CHECKED-IN source; the reference publishes 130,248 for this configuration (benches/README.md:60): its RSA region is "
                 "the same 121,579 rows, its hash region is an earlier revision's (7,217 rows where src/lib.rs:222-259 now takes 2,182)" % info.total_rows)
    spec = dict(circuit=0, k=kk, n_big=n_big, e=e, x=x, exp_bits=bits, message=message)
    circ.native_spec = spec if k == kk else None
    if k > kk:
        circ = circuits._tile(circ, k)
        circ.tiled_spec = spec          # end_to_end_tiled(): the 2^kk-row circuit is synthesized once per proof and stacked on the device
        desc += ", stacked %d times" % (1 << (k - kk))
        published = None
    return circ, desc, published


def ctx_device(ctx):
    import torch
    return torch.cuda.current_device()


class ProofSetup:
    """Circuit, SRS, proving key and a prover for one (k, circuit): the one-time work the reference caches on disk
    (benches/delay_enc.rs:41-54, 84-115).  Everything goes through the whole-call C ABI (dehalo_params_read, dehalo_keygen,
    dehalo_prover_create, dehalo_create_proof): Python only passes pointers.  The SRS travels through ParamsKZG's RawBytes format, as
    the reference's does."""

    def __init__(self, pkg, ctx, k, circuit, threads):
        import io
        import plonk_oracle as PO          # synthetic-input generation (the SRS) + the cpu_baseline / checker leg
        import pairing as pr
        import numpy as np
        from dehalo2_amd import circuits, keygen, native

        po = sys.modules["pyoracle"]
        self.curve, self.ocurve, self.k, self.circuit = pkg.fields.BN254, po.BN254, k, circuit
        self.s = 0x64656C6179656E63 * 0x9E3779B97F4A7C15 % self.curve.scalar.p
        t0 = time.time()
        self.circ, self.witness, self.rows_reference = real_witness(self.curve.scalar.p, k, circuit)
        tw = time.time()
        # ParamsKZG::setup (benches/delay_enc.rs:43) on the device: dehalo_params_setup.  The CPU restatement builds ITS OWN SRS from the same secret when a
        # CPU leg asks for it (`srs` below), and the two are compared there: the product takes nothing from oracle/.
        self.params = native.ParamsKZG.setup(ctx, self.curve, k, self.s)
        t1 = t2 = time.time()
        self._threads, self._srs = threads, None
        self.pk = native.ProvingKey.keygen(ctx, self.params, self.circ.cs, self.circ.fixed, self.circ.assembly, self.circ.selectors)
        ctx.synchronize()
        t3 = time.time()
        # the verifying key as the (Python) verifier reads it, and the transcript_repr convention shared with the CPU restatement
        self.vk = keygen.VerifyingKey.read(self.curve, self.circ.cs, io.BytesIO(self.pk.vk_bytes()), len(self.circ.selectors))
        self.pk.transcript_repr = self.vk.transcript_repr
        self.ctx = ctx
        self.side = pkg.Context(ctx_device(ctx))                      # NTTs and the random commitment run beside the commitment phases
        self.prover = native.Prover(self.params, self.pk, ctx, self.side)
        with ctx.torch_stream():
            self.advice = keygen.to_device(self.circ.advice)          # the witness, resident in HBM (Montgomery form)
            ctx.field_op_device(self.curve.scalar.id, "to_mont", self.advice.data_ptr(), 0, self.advice.data_ptr(), self.advice.numel() // 4, 0)
        ctx.synchronize()
        self.setup_s = {"witness_python": round(tw - t0, 2), "params_setup_gpu": round(t1 - tw, 3), "keygen_gpu": round(t3 - t2, 3)}
        self.witness_ms = round(1e3 * (tw - t0), 1)

    @property
    def srs(self):
        """the CPU restatement's SRS for the same secret (checker legs only), asserted equal to the device-made one"""
        if self._srs is None:
            import numpy as np
            import plonk_oracle as PO
            self._srs = PO.setup_srs(self.ocurve, self.k, self.s, self._threads)
            raw, n = self.params.write(), 1 << self.k
            assert raw[4:4 + 64 * n] == np.ascontiguousarray(self._srs["g"]).tobytes() and raw[4 + 64 * n:4 + 128 * n] == np.ascontiguousarray(self._srs["g_lagrange"]).tobytes(), \
                "dehalo_params_setup's SRS differs from the CPU restatement's"
        return self._srs

    def prove(self, seed, which=None):
        from dehalo2_amd import prover
        return (which or self.prover).create_proof(self.advice, [[]], prover.SeededRng(seed)).finalize()

    def release(self):
        self.prover.release()
        self.pk.release()
        self.params.release()
        self.side.close()


def end_to_end(st, want_proof, reps=5):
    """What the reference's timed create_proof covers: Circuit::synthesize (witness generation: dehalo_synthesize, C++, host) + upload of the
    advice columns + the proof.  The advice it produces must be the witness the proof above was made from (asserted through the proof bytes)."""
    import numpy as np
    from dehalo2_amd import native, prover
    spec = getattr(st.circ, "native_spec", None)
    if spec is None:
        return None
    kw = {a: b for a, b in spec.items() if a not in ("circuit", "k")}
    import torch
    # one page-locked buffer for the advice columns, as a host that proves repeatedly keeps: the witness generator writes its rows straight into it
    # and the upload is one DMA from it (a fresh pageable array per proof costs the page faults of 21 MB and a staged copy: `ms_with_a_fresh_pageable_array_per_proof`)
    pinned = torch.empty((5, 1 << spec["k"], 4), dtype=torch.int64).pin_memory()
    buf = pinned.numpy().view(np.uint64)
    ts, tw = [], []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        nat = native.synthesize(spec["circuit"], spec["k"], out=buf, **kw)
        t1 = time.perf_counter()
        proof = st.prover.create_proof(nat["advice"], [[]], prover.SeededRng(7), canonical=True).finalize()
        t2 = time.perf_counter()
        if i:
            tw.append(1e3 * (t1 - t0)); ts.append(1e3 * (t2 - t0))
    assert proof == want_proof, "the proof from the natively synthesized witness differs"
    tc = []
    for i in range(reps + 1):      # the reference's call shape: create_proof(&params, &pk, &[circuit], ...) synthesizes inside (dehalo_create_proof_circuit)
        t0 = time.perf_counter()
        tr, _info = st.prover.create_proof_circuit(spec["circuit"], [[]], prover.SeededRng(7), **kw)
        proof = tr.finalize()
        if i:
            tc.append(1e3 * (time.perf_counter() - t0))
    assert proof == want_proof, "the proof of dehalo_create_proof_circuit differs"
    tp = []
    for i in range(3):      # the same with a fresh pageable array per proof (numpy's default), for comparison
        t0 = time.perf_counter()
        nat = native.synthesize(spec["circuit"], spec["k"], **kw)
        proof = st.prover.create_proof(nat["advice"], [[]], prover.SeededRng(7), canonical=True).finalize()
        tp.append(1e3 * (time.perf_counter() - t0))
    assert proof == want_proof
    return {"ms": round(min(tc), 3), "ms_median": round(sorted(tc)[len(tc) // 2], 3), "ms_as_two_calls": round(min(ts), 3), "witness_ms": round(min(tw), 3), "ms_with_a_fresh_pageable_array_per_proof": round(min(tp), 3),
            "witness_threads": min(8, os.cpu_count() or 1) if os.environ.get("DEHALO_SYNTH_THREADS") is None else int(os.environ["DEHALO_SYNTH_THREADS"]),
            "what": "dehalo_create_proof_circuit: the reference's call (the circuit is synthesized inside; the random polynomial's commitment runs on the idle device meanwhile).  As two calls: dehalo_synthesize (C++ witness generation on the host: the RSA regions of the circuit written by up to `witness_threads` threads, the rest by one; rows go into a page-locked buffer kept across proofs) + upload of 5 x 2^k advice values from it + dehalo_create_proof; "
                    "same proof bytes as from the resident witness"}


def end_to_end_tiled(st, want_proof, reps=3):
    """The k > 17 configurations are 2^(k - 17) copies of the k = 17 circuit stacked row-wise (a synthetic size: the reference's circuit does not grow with k).
    What a host that proves this circuit does per proof: dehalo_synthesize of the 2^17-row block (C++, into a page-locked buffer), upload of its 5 x 2^17
    advice values, the copies made ON THE DEVICE, then dehalo_create_proof -- same proof bytes as from the resident witness."""
    import numpy as np
    import torch
    from dehalo2_amd import native, prover
    spec = getattr(st.circ, "tiled_spec", None)
    if spec is None:
        return None
    kw = {a: b for a, b in spec.items() if a not in ("circuit", "k")}
    kk, k = spec["k"], st.k
    pinned = torch.empty((5, 1 << kk, 4), dtype=torch.int64).pin_memory()
    buf = pinned.numpy().view(np.uint64)
    block = torch.empty((5, 1 << kk, 4), dtype=torch.int64, device="cuda")
    full = torch.empty((5, 1 << k, 4), dtype=torch.int64, device="cuda")
    ts, tw = [], []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        native.synthesize(spec["circuit"], kk, out=buf, **kw)
        t1 = time.perf_counter()
        st.ctx._check(st.ctx.lib.dehalo_upload(st.ctx.handle, buf.ctypes.data, buf.nbytes, block.data_ptr()))
        full.view(5, 1 << (k - kk), 1 << kk, 4).copy_(block.unsqueeze(1).expand(5, 1 << (k - kk), 1 << kk, 4))
        proof = st.prover.create_proof(full, [[]], prover.SeededRng(7), canonical=True).finalize()
        t2 = time.perf_counter()
        if i:
            tw.append(1e3 * (t1 - t0)); ts.append(1e3 * (t2 - t0))
    assert proof == want_proof, "the proof from the natively synthesized, device-stacked witness differs"
    return {"ms": round(min(ts), 3), "ms_median": round(sorted(ts)[len(ts) // 2], 3), "witness_ms": round(min(tw), 3),
            "what": "dehalo_synthesize of the 2^%d-row circuit (host, page-locked buffer) + upload of its 5 x 2^%d advice values + %d copies stacked on the device + dehalo_create_proof; "
                    "same proof bytes as from the resident witness" % (kk, kk, 1 << (k - kk))}


CIRCUIT_TEXT = {"delay_enc": "DelayEncryptCircuit shape (MainGate + RangeChip: 5 advice, 15 fixed, 5 lookups, degree 5)",
                "mod_pow": "benches/mod_pow.rs RSACircuit (same constraint system as delay_enc: MainGate + RangeChip)",
                "pose_enc": "pose_enc shape (MainGate only: 5 advice, 9 fixed, degree 3)"}


def verify_with_device_vk(st, proof):
    """verify_proof (oracle/verifier.py: pairing check) against the verifying key the DEVICE keygen produced."""
    import pairing as pr
    import verifier as V
    from dehalo2_amd import keygen
    vk = st.vk
    import shapes
    return V.verify_proof(st.ocurve, shapes.maingate_description(bool(st.circ.cs.lookups)), st.k, keygen.decode_points(st.curve, vk.fixed_commitments),
                          keygen.decode_points(st.curve, vk.permutation_commitments), vk.transcript_repr, (1, 2), pr.G2, pr.g2_mul(st.s, pr.G2), [[]], proof)


def proof_numbers(pkg, co, po, ctx, k, circuit, with_cpu, verify, reps=5):
    """One half of BASELINE's metric: create_proof of the delay_enc / mod_pow / pose_enc circuit -- a real proof: what is
    committed is what was computed, challenges come from the Blake2b transcript.  With `with_cpu` the CPU restatement makes the
    same proof from the same witness, blinding and SRS: its bytes must equal the device's (asserted) and its time is the CPU
    figure.  With `verify` the verifier (pairing check) must accept the device proof -- against the CPU restatement's verifying key
    when there is one (asserted equal to the device's), otherwise against the device-made key."""
    from dehalo2_amd import prover
    threads = min(host_cores(), 256)
    note("proof %s k = %d: setup" % (circuit, k))
    st = ProofSetup(pkg, ctx, k, circuit, threads)
    note("proof %s k = %d: proving" % (circuit, k))
    proof = st.prove(7)
    preheat(lambda: st.prove(7))
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        again = st.prove(7)
        ts.append(1e3 * (time.perf_counter() - t))
        assert again == proof, "the same witness, SRS and blinding gave different proof bytes"
    phases = st.prover.last_timings()
    note("proof %s k = %d: end to end" % (circuit, k))
    e2e = end_to_end(st, proof) or end_to_end_tiled(st, proof)
    note("proof %s k = %d: checks" % (circuit, k))
    cs = st.circ.cs
    n_evals = len(cs.advice_queries) + len(cs.fixed_queries) + 1 + len(cs.permutation_columns) + max(0, 3 * cs.num_permutation_sets() - 1) + 5 * len(cs.lookups)
    out = {"circuit": CIRCUIT_TEXT[circuit], "circuit_name": circuit,
           "k": k, "curve": "bn254 (KZG, GWC)", "rows_used": st.circ.used_rows, "rows_reference": st.rows_reference, "commitments": len(proof) // 32 - n_evals,
           "proof_bytes": len(proof), "gpu_ms": round(min(ts), 3), "gpu_ms_median": round(sorted(ts)[len(ts) // 2], 3),
           "gpu_phase_ms": {a: round(b, 3) for a, b in phases.items()},
           "driver": "dehalo_create_proof: phases, transcript and every launch in C++ behind the C ABI; Python passes pointers",
           "witness": st.witness + "; resident in HBM when the timed call starts; blinding scalars generated on the host inside the timed call",
           "witness_python_ms": st.witness_ms,
           "end_to_end": e2e,
           "one_time_setup_s": st.setup_s}
    if with_cpu:
        import plonk_oracle as PO
        import numpy as np
        t0 = time.time()
        import shapes
        assert shapes.maingate_description(bool(cs.lookups)) == cs.description(), "the product's constraint system differs from the checker's statement of it"
        key = PO.keygen(st.ocurve, st.srs, shapes.maingate_description(bool(cs.lookups)), k, st.circ.fixed, st.circ.assembly.mapping, threads)
        t1 = time.time()
        rep = PO.transcript_repr(st.ocurve, key, st.circ.selectors)
        adv = np.stack([co.field_op(st.curve.scalar.id, "to_mont", st.circ.advice[i]) for i in range(cs.num_advice)])
        # the port parallelises with plain pthreads per call: on a many-core host fewer threads can be faster, so both are timed
        cpu_runs = {}
        for th in sorted({threads, min(threads, 32)} if k <= 17 else {min(threads, 32)}):      # (k = 20: one CPU proof, at the thread count that wins at every smaller size)
            t2 = time.time()
            want, _ = PO.create_proof(st.ocurve, st.srs, key, adv, [[]], PO.ScalarStream(7), rep, th)
            cpu_runs[th] = time.time() - t2
        best_th = min(cpu_runs, key=cpu_runs.get)
        t2, t3 = 0.0, cpu_runs[best_th]
        assert rep == st.vk.transcript_repr and st.pk.vk_bytes() == PO.vk_bytes(st.ocurve, key, st.circ.selectors), "verifying key differs from the CPU restatement's"
        assert proof == want, "device proof differs from the CPU restatement's proof"
        out.update({"identical_to_cpu_proof": True, "cpu_ms": round(1e3 * (t3 - t2), 1), "cpu_keygen_ms": round(1e3 * (t1 - t0), 1), "cpu_cores": best_th, "host_cores": host_cores(),
                    "cpu_ms_by_threads": {str(a): round(1e3 * b, 1) for a, b in cpu_runs.items()},
                    "cpu_kind": "port (oracle/plonk_oracle.py over oracle/oracle.c)", "speedup_vs_cpu_port": round(1e3 * (t3 - t2) / min(ts), 1)})
    if verify:
        tv = time.time()
        assert verify_with_device_vk(st, proof), "the verifier rejected the device proof"
        out.update({"verifier_accepts": True, "verify_s_python": round(time.time() - tv, 2),
                    "verified_against": "the device-made verifying key" + (" (asserted equal to the CPU restatement's)" if with_cpu else "")})
    st.release()
    return out


def batch_proofs(pkg, ctx, k, total, rank, world, backend, device, inflight, check=True, also_total=0):
    """configs[4]: a batch of delay_enc proofs dealt round-robin to the ranks (proof p -> rank p mod N, SRS and proving key
    replicated), `inflight` provers per GPU (one context + one library thread each: dehalo_create_proofs, no interpreter in the loop),
    then ONE all-gather of every proof's commitments (31 x 32 B compressed each).  After the timed region every rank re-makes each of
    ITS proofs alone (one prover with a side context, nothing else in flight) from the same seed and requires the batch-made proof and
    the gathered blob to hold exactly that proof; rank 0 also puts one batch-made proof through the pairing check.
    `total` is the weak-scaling batch (32 proofs per GPU); `also_total` > 0 times a second batch of exactly that many proofs over the same
    setup -- BASELINE configs[4] read literally: 64 proofs whatever the number of GPUs -- returned under "fixed_batch"."""
    import torch
    from dehalo2_amd import native, prover, sharding
    note("batch mode: setup")
    st = ProofSetup(pkg, ctx, k, "delay_enc", max(8, min(host_cores(), 256) // max(1, world)))      # (every rank builds the synthetic SRS on the host at once: share the cores)
    if os.environ.get("DEHALO_BENCH_KEEP_TRANSFER") is None:
        from dehalo2_amd import _lib as _l
        _l.release_transfer_contexts()                        # (its stream would count against the pool of hardware queues the provers' streams are mapped onto)
    st.prove(1000)                                            # warm-up
    cs = st.circ.cs
    prios = (1, 0, -1) if os.environ.get("DEHALO_BENCH_FLAT_PRIORITIES") is None else (0,)      # (see main(): hardware queues are pooled per priority)
    ctxs = [pkg.Context(device, priority=prios[i % len(prios)]) for i in range(max(1, inflight))]
    provers = [native.Prover(st.params, st.pk, c) for c in ctxs]
    alone_cache = {}

    def run(total, with_synth):
        mine = sharding.units_for_rank(total, rank, world)
        preheat(lambda: native.create_proofs(provers, st.advice, [prover.SeededRng(999 - i) for i in range(2 * len(provers))]))      # warm-up of every prover's buffers, clocks up
        fence_all(world)
        note("batch mode: timed region, %d proofs" % total)
        t0 = time.perf_counter()
        full = native.create_proofs(provers, st.advice, [prover.SeededRng(1000 + unit) for unit in mine])          # every proof its own blinding
        blobs = [prover.proof_commitments(cs, pf) for pf in full]
        allc = sharding.gather_proof_commitments(blobs, total, rank, world, "cuda" if backend == "nccl" else "cpu")
        per = len(allc[0]) // 32
        fence_all(world)
        elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
        assert len(allc) == total and all(len(b) == 32 * per for b in allc)
        # the same batch with every proof's circuit synthesized inside its call (dehalo_create_proofs_circuit: witness generation of one proof beside the device
        # work of the others) -- reported beside the figure above, whose witness is resident
        synth = None
        spec = getattr(st.circ, "native_spec", None)
        if with_synth and spec is not None:      # (every rank takes this branch: the fences inside are collective)
            kw = {a: b for a, b in spec.items() if a not in ("circuit", "k")}
            fence_all(world)
            t1 = time.perf_counter()
            full_s = native.create_proofs_circuit(provers, spec["circuit"], [kw] * len(mine), [prover.SeededRng(1000 + unit) for unit in mine])
            fence_all(world)
            el_s = sharding.max_over_ranks(time.perf_counter() - t1)
            assert full_s == full, "batch proofs from circuits synthesized inside the calls differ from those of the resident witness"
            synth = {"proofs_per_s": round(total / el_s, 2), "ms_per_proof_per_gpu": round(1e3 * el_s / max(1, len(mine)), 3),
                     "what": "dehalo_create_proofs_circuit: the same proofs (bytes asserted equal), each circuit synthesized inside its call"}
        checked = None
        note("batch mode: checks")
        if check:
            for j, unit in enumerate(mine):                       # this rank's units, re-made alone: the gathered vector holds them at [unit]
                if unit not in alone_cache:
                    alone_cache[unit] = st.prove(1000 + unit)
                alone = alone_cache[unit]
                assert alone == full[j], "batch-mode proof %d differs from the same proof made alone" % unit
                assert allc[unit] == prover.proof_commitments(cs, alone), "gathered commitments of proof %d differ from the proof made alone" % unit
            checked = "every proof of this rank re-made alone from its seed: proof bytes and gathered commitments identical"
            if rank == 0 and mine:
                assert verify_with_device_vk(st, full[-1]), "the verifier rejected a batch-made proof"
                checked += "; verifier accepts a batch-made proof"
        return {"k": k, "proofs": total, "proofs_total": total, "n_gpus": world, "proofs_in_flight_per_gpu": len(provers), "proofs_per_s": round(total / elapsed, 2),
                "ms_per_proof_per_gpu": round(1e3 * elapsed / max(1, len(mine)), 3),
                "driver": "dehalo_create_proofs: one library thread per prover, no interpreter in the loop",
                "gathered": "%d proofs x %d compressed commitments (32 B each) on every rank, one all_gather" % (total, per),
                "with_witness_generation": synth,
                "checked_after_timed_region": checked,
                "parallelism": "proof p -> rank p mod N; SRS / proving key replicated; no data-path collective"}

    out = run(total, True)
    if also_total > 0:
        out["fixed_batch"] = run(also_total, False) if also_total != total else dict(out, with_witness_generation=None)
    for p in provers:
        p.release()
    for c in ctxs:
        c.close()
    st.release()
    return out


def measured_copy_ceiling():
    """"builder must also record a measured copy-kernel ceiling".  A 1 GiB device-to-device copy (torch's copy kernel:
    1 GiB read + 1 GiB written per launch), best of 5, timed with events on torch's stream; beside it the read-only streaming rate
    of tools/pmc_calib's k_stream16 when profiles/ holds it."""
    import torch
    try:
        src = torch.empty(1 << 27, dtype=torch.int64, device="cuda")
        dst = torch.empty_like(src)
        src.fill_(3)
        torch.cuda.synchronize()
        best = None
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dst.copy_(src); b.record(); b.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best
        out = {"copy_GBps": round(2 * (1 << 30) / (best * 1e-3) / 1e9, 1), "copy_ms_per_GiB": round(best, 4),
               "what": "1 GiB device-to-device copy (read + write = 2 GiB of traffic), best of 6"}
        del src, dst
        torch.cuda.empty_cache()
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)}
    rec = os.path.join(ROOT, "profiles", "copy_ceiling.json")
    if os.path.exists(rec):
        try:
            out["recorded"] = json.load(open(rec))
        except Exception:      # noqa: BLE001
            pass
    return out


def fence_all(world):
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()


def under_profiler() -> bool:
    return "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ)


def _section_of(stderr_tail: str) -> str:
    """the last `[bench … s] section` marker note() printed before a measuring process died"""
    last = ""
    for line in stderr_tail.splitlines():
        if line.startswith("[bench "):
            last = line.split("] ", 1)[-1]
    return last


def supervise() -> int:
    """--supervise (opt-in since round 5; the default run measures in-process, so a device fault is a non-zero exit code the caller sees): the
    measurement runs in a child process (nothing in this one has touched the GPU).  A child that dies by a signal -- seen once in round 3:
    `Memory access fault by GPU node` in one full run out of about ten; 52 / 52 clean since the pageable copies went -- is replaced ONCE by a
    FRESH child (never a re-exec); the JSON line then carries "attempts": 2 and "first_attempt": {"signal", "section"} of the one that died.
    A child that exits by itself (any code) is final."""
    import subprocess
    import tempfile
    first = None
    for attempt in (1, 2):
        env = dict(os.environ, DEHALO_BENCH_ATTEMPT=str(attempt))
        if first is not None:
            env["DEHALO_BENCH_FIRST_ATTEMPT"] = json.dumps(first)
        with tempfile.TemporaryFile() as errf:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--supervise"] + ["--in-process"], stdout=subprocess.PIPE, stderr=errf, env=env)
            errf.seek(0)
            err = errf.read().decode(errors="replace")
        sys.stderr.write(err)
        sys.stderr.flush()
        out = p.stdout.decode()
        if p.returncode >= 0 or attempt == 2:
            sys.stdout.write(out)
            sys.stdout.flush()
            return p.returncode if p.returncode >= 0 else 128 - p.returncode
        tail = [l for l in err.splitlines() if l.strip()][-6:]
        first = {"signal": -p.returncode, "section": _section_of(err), "stderr_tail": [l[:300] for l in tail]}
        sys.stderr.write("[bench] the measuring process was killed by signal %d in section %r; its partial output is dropped and ONE more is started\n" % (-p.returncode, first["section"]))
    return 1


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment (the driver's N = 1 command shape with N changed): this process has not
    touched the GPU, so it starts `python -m torch.distributed.run --nproc-per-node N bench.py ... --in-process` as a FRESH child process (one rank per GPU
    over RCCL), relays its output -- rank 0's single JSON line -- and returns its exit code."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--in-process"] + ["--in-process"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the pool's driver supports dmabuf IPC only (RCCL's intra-node transport)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    p = subprocess.run(cmd, env=env)
    return p.returncode if p.returncode >= 0 else 128 - p.returncode


LINE_LIMIT = 6000      # characters: the driver keeps the last ~8 KB of stdout; tests/test_host_logic.py holds a canned record to this bound


def _pick(d, *names):
    return {n: d[n] for n in names if d is not None and n in d and d[n] is not None}


def compact_proof(p):
    """A proof section of the line: numbers and one-word tags.  The sentences (what `e2e_ms` covers, where the witness lives, which verifying key the
    pairing check used) are in profiles/README.md, "bench line glossary", and in the verbose record (--full-out)."""
    if p is None:
        return None
    name = p.get("circuit_name")
    if name is None and isinstance(p.get("circuit"), str):      # (records written before round 5 carry the sentence only)
        name = {"D": "delay_enc", "b": "mod_pow", "p": "pose_enc"}.get(p["circuit"][0], p["circuit"])
    out = {"circuit": name, "k": p["k"], "rows": p.get("rows_used"), "rows_reference": p.get("rows_reference"), "commitments": p.get("commitments"), "proof_bytes": p.get("proof_bytes"),
           "gpu_ms": p["gpu_ms"], "gpu_ms_median": p.get("gpu_ms_median"), "phase_ms": p.get("gpu_phase_ms"), "witness": "resident"}
    e = p.get("end_to_end")
    if e:
        out["e2e_ms"], out["e2e_witness_ms"] = e.get("ms"), e.get("witness_ms")
    out.update(_pick(p, "identical_to_cpu_proof", "cpu_ms", "cpu_cores", "speedup_vs_cpu_port", "verifier_accepts"))
    if "cpu_ms" in out:
        out["cpu_kind"] = "port"
    return out


def compact_line(out):
    """The JSON line printed on stdout: every number of the verbose record `out`, none of its prose; the contract's keys first, then the roofline and
    cpu_baseline objects, then the proof sections with BASELINE's own metric -- the k = 17 delay_enc proof, the batch, `attempts` -- LAST, so that a
    reader who keeps only the tail of stdout keeps them.  The three scalars of that half of the metric are also inside `config`."""
    proof, batch = out.get("proof"), out.get("batch_proofs")
    cfg = dict(out["config"])
    cfg["delay_enc_k17_ms"] = proof["gpu_ms"] if proof and proof.get("k") == 17 else None
    cfg["batch_proofs_per_s"] = batch["proofs_per_s"] if batch else None
    cfg["attempts"] = out.get("attempts", 1)
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["dtype"] = "u256"
    line["config"] = cfg
    r = out["roofline"]
    rl = _pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "avg_kernel_ms_overlapped", "frac_of_measured_peak")
    if r.get("traffic") is not None:      # not measured by this run: read from profiles/pmc_traffic.json (separate rocprofv3 --pmc passes; the kernel revision is checked)
        rl["traffic_src"] = "pmc " + KERNEL_REV
    rl["algorithmic_bytes"] = r.get("algorithmic_bytes_per_launch")
    if r.get("measured_peak"):
        rl["measured_copy_GBps"] = r["measured_peak"].get("copy_GBps")
    rl["valu"] = _pick(r.get("valu") or {}, "achieved_tmad_per_s", "peak_tmad_per_s", "frac")
    if out.get("by_k"):      # the north star's per-k / per-distribution figures (by_k_numbers): MSM Mpoints/s and HBM fraction for u / w / l scalars, NTT at n and 4n --
        rl["by_k"] = out["by_k"]      # inside `roofline`, one of the objects the driver's record keeps whole
    line["roofline"] = rl
    if out.get("cpu_baseline"):
        c = out["cpu_baseline"]
        line["cpu_baseline"] = dict(_pick(c, "value", "unit", "cores", "host_cores", "kind", "msm_ms", "ntt_ms"), sample="2 x (MSM + NTT 2^%d), rank 0's inputs" % out.get("log_n", 20))
    n = out.get("ntt_roofline")
    if n:
        line["ntt_roofline"] = dict(_pick(n, "bound", "achieved", "peak", "unit", "frac", "avg_transform_ms", "avg_transform_ms_overlapped"),
                                    valu=_pick(n.get("valu") or {}, "achieved_tmad_per_s", "frac"))
    line["preheat_s"] = (out.get("preheat") or {}).get("seconds")
    if out.get("breakdown_ms_per_step"):
        line["alone_ms"] = out["breakdown_ms_per_step"]
    if out.get("breakdown_ms_per_step_overlapped"):
        line["overlapped_ms"] = {k: v for k, v in out["breakdown_ms_per_step_overlapped"].items() if k != "note"}
    if out.get("single_stream"):
        line["single_stream_ms_per_step"] = out["single_stream"].get("ms_per_step")
    if out.get("parity_of_timed_configuration"):
        line["parity_of_timed_steps"] = "all equal the CPU port's"
    if out.get("secondary"):
        line["secondary"] = _pick(out["secondary"], "table_build_ms_precomputed", "msm_single_row_device_ms", "best_multiexp_host_buffers_ms")
    if out.get("rccl_world"):
        line["rccl_world"] = out["rccl_world"]
    if out.get("proof_other_k"):
        line["proof_other_k"] = [compact_proof(p) for p in out["proof_other_k"]]
    for k in ("proof_pose_enc", "proof_mod_pow", "proof"):
        if out.get(k):
            line[k] = compact_proof(out[k])
    if batch:
        b = _pick(batch, "k", "proofs", "proofs_total", "n_gpus", "proofs_in_flight_per_gpu", "proofs_per_s", "ms_per_proof_per_gpu")
        if batch.get("with_witness_generation"):
            b["with_witness_generation_proofs_per_s"] = batch["with_witness_generation"].get("proofs_per_s")
        b["checked"] = "each proof re-made alone: identical" if batch.get("checked_after_timed_region") else None
        fb = batch.get("fixed_batch")
        if fb:      # configs[4] read literally: 64 proofs whatever the number of GPUs
            b["batch%d_proofs_per_s" % fb["proofs"]] = fb["proofs_per_s"]
            b["batch%d_checked" % fb["proofs"]] = bool(fb.get("checked_after_timed_region"))
        line["batch_proofs"] = b
    line["attempts"] = out.get("attempts", 1)
    if out.get("first_attempt"):
        f = out["first_attempt"]
        line["first_attempt"] = {"signal": f.get("signal"), "section": (f.get("section") or "")[:60]}
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("secondary", "overlapped_ms", "proof_pose_enc", "proof_mod_pow"):      # never needed at today's sizes; the bound holds whatever a section grows to
        if len(text) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT and len(line.get("proof_other_k") or []) > 2:      # (a run with more extra sizes than the default two)
        line["proof_other_k"] = line["proof_other_k"][:2]
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        line["roofline"].pop("by_k", None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1 and not args.in_process:
        raise SystemExit(launch_ranks(args.gpus))      # a fresh torchrun child; this process never touches the GPU
    if args.supervise and not args.in_process and "RANK" not in os.environ and not under_profiler():
        raise SystemExit(supervise())
    if os.environ.get("DEHALO_BENCH_SELFTEST_KILL") and os.environ.get("DEHALO_BENCH_SELFTEST_KILL") == os.environ.get("DEHALO_BENCH_ATTEMPT"):
        import signal      # tests/test_host_logic.py: the measuring process of this attempt dies by a signal, as a device fault would end it
        os.kill(os.getpid(), signal.SIGABRT)
    global PREHEAT_S
    PREHEAT_S = max(0.0, args.preheat_s)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the measured path)")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    rccl_world = None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" = RCCL over xGMI.  (The timeout covers rank 0's CPU baseline and proof checks at the end, which the other ranks wait out in a barrier.)
        dist.init_process_group(args.dist_backend, timeout=datetime.timedelta(minutes=30))
        seen = torch.tensor([dist.get_world_size()], dtype=torch.int64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        allseen = [torch.zeros_like(seen) for _ in range(world)]
        dist.all_gather(allseen, seen)
        rccl_world = {"backend": args.dist_backend + (" (RCCL)" if args.dist_backend == "nccl" else " (CPU rehearsal)"), "world_size_seen_by_rank": [int(t.item()) for t in allseen]}

    pkg = entry.load_package()
    po, co = entry.load_oracle()  # synthetic-input generators + the cpu_baseline leg only
    from dehalo2_amd import _lib, sharding
    if args.sort_block:
        _lib.Context.default_tuning["msm_sort_block"] = args.sort_block

    curve = pkg.fields.CURVES[args.curve]
    field = pkg.fields.FIELDS[args.ntt_field]
    log_n, n = args.log_n, 1 << args.log_n

    # One context (= one HIP stream + one workspace) per step in flight: consecutive steps are
    # independent units (different columns / proofs), so step i+1's sort and bucket accumulation
    # overlap step i's latency-bound bucket reduction.  The SRS tables are shared.
    inflight = max(1, args.inflight)
    # stream priorities alternate (highest / default / lowest): the runtime keeps a pool of hardware queues PER priority and maps streams of one
    # priority onto few of them -- four default-priority streams were measured running two at a time (tools/stream_concurrency.hip)
    prios = (1, 0, -1) if os.environ.get("DEHALO_BENCH_FLAT_PRIORITIES") is None else (0,)
    ctxs = [pkg.Context(local_rank, priority=prios[i % len(prios)]) for i in range(inflight)]
    ctx = ctxs[0]
    if args.acc_waves:
        for c in ctxs:
            c.set_tuning("msm_acc_points", 0)
            c.set_tuning("msm_acc_waves", args.acc_waves)
    # synthetic SRS and witnesses; every rank gets its own scalar column
    bases_h = co.synth_bases(curve.id, n)
    scalars_h = co.fill_scalars(curve.scalar.id, args.dist, n, 1000 + rank)
    poly_h = co.fill_scalars(field.id, "uniform", n, 2000 + rank)
    bases = ctx.register_bases(curve.id, bases_h, args.window_bits, True)  # resident SRS tables
    # (every host array goes to HBM through the library's staged upload -- Context.upload -- never through `tensor.cuda()` of a pageable array)
    d_scalars = ctx.upload(scalars_h)
    d_polys = [ctx.upload(poly_h) for _ in range(inflight)]
    # one output row per step: the commitment vector of this rank (gathered once, at the end)
    d_out_all = torch.zeros((args.warmup + args.steps + 1, 12), dtype=torch.int64, device="cuda")
    d_out = d_out_all[0:1]
    omega = field.encode(po.FIELDS[field.name].omega(log_n))
    torch.cuda.synchronize()
    counter = [0]

    def step():
        i = counter[0]
        k = i % inflight
        counter[0] += 1
        c = ctxs[k]
        c.msm_device(bases, d_scalars.data_ptr(), n, 1, d_out_all[i].data_ptr(), 0)   # the context's own stream
        c.ntt_device(field.id, d_polys[k].data_ptr(), log_n, omega, 1, 0)

    def gather_commitments(first, count):
        """RCCL all-gather of this job's commitment vector (north star: 'all-gather over xGMI for the
        final commitment vector'): every rank ends with all world x count commitments, in unit order."""
        for c in ctxs:
            c.synchronize()   # the collective runs on torch's stream: order it after the MSMs
        local = d_out_all[first:first + count]
        if world == 1:
            return local
        if args.dist_backend != "nccl":
            local = local.cpu()
        return sharding.all_gather_commitments(local, world * count, rank, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    pre = [0]

    def step_preheat():          # the timed region's own pattern (steps dealt round-robin to the contexts in flight), results to a spare row
        c = ctxs[pre[0] % inflight]
        c.msm_device(bases, d_scalars.data_ptr(), n, 1, d_out_all[args.warmup + args.steps].data_ptr(), 0)
        c.ntt_device(field.id, d_polys[pre[0] % inflight].data_ptr(), log_n, omega, 1, 0)
        pre[0] += 1
    note("steps: preheat, warm-up, timed region")
    import gc
    gc.collect(); gc.disable()      # (as timeit does: no collector pause of the interpreter inside the 30 ms region -- and none between the preheat and it)
    preheat_steps = preheat(step_preheat, lambda: [c.synchronize() for c in ctxs]) if PREHEAT_S > 0 else 0
    for _ in range(args.warmup):
        step()
    gather_commitments(0, max(args.warmup, 1))
    fence()
    for c in ctxs:
        c.timing_enable(True)
        c.timing_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    commitments = gather_commitments(args.warmup, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    assert commitments.shape == (world * args.steps, 12)
    elapsed = sharding.max_over_ranks(elapsed)
    for c in ctxs:
        c.timing_enable(False)

    # Not part of the metric: the same step with nothing else in flight, so that the per-kernel device
    # times are free of the overlap with the neighbouring steps' kernels (reported as single_stream).
    def collect(cs):
        return {kid: (sum(c.timing_get(kid)[0] for c in cs), sum(c.timing_get(kid)[1] for c in cs))
                for kid in (_lib.K_MSM_ACCUMULATE, _lib.K_MSM_SORT, _lib.K_MSM_REDUCE, _lib.K_NTT_PASS)}
    overlapped = collect(ctxs) if rank == 0 else None
    ss_steps = 0 if args.no_single_stream else min(10, args.steps)

    def one_step_alone():
        ctx.msm_device(bases, d_scalars.data_ptr(), n, 1, d_out_all[0].data_ptr(), 0)
        ctx.ntt_device(field.id, d_polys[0].data_ptr(), log_n, omega, 1, 0)
    if ss_steps:      # the clocks this (lighter) load settles at, as in the profiled one-step-at-a-time run the roofline's duration is compared with (profiles/README.md)
        preheat(one_step_alone, ctx.synchronize)
    ctx.timing_reset(); ctx.timing_enable(True)
    torch.cuda.synchronize()
    ts = time.perf_counter()
    for i in range(ss_steps):
        one_step_alone()
    ctx.synchronize()
    ss_ms = (time.perf_counter() - ts) * 1e3 / max(ss_steps, 1)
    ctx.timing_enable(False)
    single = collect([ctx]) if rank == 0 else None

    n_proofs = args.proofs if args.proofs >= 0 else 32 * world
    batch = None
    if args.proof_k > 0 and n_proofs > 0:
        batch = batch_proofs(pkg, ctx, args.proof_k, n_proofs, rank, world, args.dist_backend, local_rank, args.proofs_inflight, also_total=args.fixed_batch)

    if rank == 0:
        def tsum(kid):
            return overlapped[kid]
        acc_ms, acc_cnt = tsum(_lib.K_MSM_ACCUMULATE)
        sort_ms, sort_cnt = tsum(_lib.K_MSM_SORT)
        red_ms, red_cnt = tsum(_lib.K_MSM_REDUCE)
        ntt_ms, ntt_cnt = tsum(_lib.K_NTT_PASS)
        acc_avg_ms = acc_ms / max(acc_cnt, 1)
        c_bits, n_windows = bases.window_bits, bases.windows          # as dehalo_bases_register chose them (dehalo_bases_info)
        # The contract's roofline is the kernel doing its work ALONE: with several steps in flight a launch's duration is stretched by
        # the neighbouring steps' kernels sharing the chip (it can exceed ms_per_step) and varies with how the streams interleave.  So
        # the figure comes from the one-step-at-a-time pass (HIP events on the launching stream, same inputs, same launches); the
        # overlapped average of the timed region is printed beside it.
        acc_alone_ms = single[_lib.K_MSM_ACCUMULATE][0] / max(single[_lib.K_MSM_ACCUMULATE][1], 1) if ss_steps else 0.0
        roof_ms, roof_src = (acc_alone_ms, "one step at a time (single_stream pass, %d launches)" % ss_steps) if acc_alone_ms > 0 else \
                            (acc_avg_ms, "timed region, %d steps in flight (overlap-inflated)" % inflight)
        achieved = MSM_BYTES_PER_TERM * n / (roof_ms * 1e-3) / 1e9 if roof_ms > 0 else 0.0
        ceiling = measured_copy_ceiling()
        # HBM traffic cannot be read by the process itself: it comes from separate rocprofv3 --pmc passes over this same command
        # (profiles/README.md), summarised by tools/pmc_summary.py into profiles/pmc_traffic.json together with the source revision
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("log_n") == log_n and rec.get("curve") == args.curve and rec.get("window_bits") == c_bits and rec.get("kernel_rev") == KERNEL_REV:
                    traffic = rec.get("msm_accumulate_hbm_bytes_per_launch")
                    traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, kernel_rev %s)" % rec.get("kernel_rev")
            except Exception:
                traffic = None
        out = {
            "metric": "pallas_msm_mpoints_per_s_at_2^%d" % log_n if args.curve == "pallas" else "%s_msm_mpoints_per_s_at_2^%d" % (args.curve, log_n),
            "value": round(world * n * args.steps / elapsed / 1e6, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat": {"seconds": PREHEAT_S, "steps": preheat_steps,
                        "what": "untimed, before the W warm-up steps: the same steps, dealt to the same contexts, for --preheat-s seconds, so that the timed region runs at a busy "
                                "prover's clocks rather than ramping up from idle after the host-side setup (the proofs' and batch mode's regions are preceded "
                                "by the same amount of untimed proofs); --preheat-s 0 switches it off"},
            "attempts": int(os.environ.get("DEHALO_BENCH_ATTEMPT", "1")),      # 2: a first measuring process was killed by a signal (supervise())
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u256 (9 x 29-bit limbs, u64 multiply-add)",
            "data": "synthetic",
            "config": {"workload": "1 x MSM(2^%d, %s) + 1 x NTT(2^%d, %s) per step per GPU, %s scalars, SRS tables resident" % (log_n, args.curve, log_n, args.ntt_field, args.dist),
                       "configs_index": 1, "steps_in_flight": inflight,
                       "parallelism": "independent units per rank; RCCL all-gather of commitments" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_msm_accum0", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": MSM_BYTES_PER_TERM * n, "avg_kernel_ms": round(roof_ms, 4), "avg_kernel_ms_source": roof_src,
                         "avg_kernel_ms_overlapped": round(acc_avg_ms, 4), "measured_peak": ceiling,
                         "frac_of_measured_peak": round(achieved / ceiling["copy_GBps"], 5) if ceiling and ceiling.get("copy_GBps") else None,
                         "note": "MSM is integer-VALU-bound (group adds), not HBM-bound: a low HBM fraction is expected (SURVEY.md 8d)",
                         # the honest ceiling of this kernel: wide integer multiplies issued vs the measured v_mad_u64_u32 peak
                         "valu": {"mads_per_launch": MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows, "peak_tmad_per_s": VMAD_PEAK_TMADS,
                                  "achieved_tmad_per_s": round(MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows / (roof_ms * 1e-3) / 1e12, 2) if roof_ms > 0 else 0.0,
                                  "frac": round(MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows / (roof_ms * 1e-3) / 1e12 / VMAD_PEAK_TMADS, 4) if roof_ms > 0 else 0.0}},
            # per-kernel device time of a step from the one-step-at-a-time pass (the kernels doing their work alone: sums to <= single_stream.ms_per_step);
            # the same regions measured inside the timed region, stretched by the neighbouring steps' kernels, are printed as *_overlapped
            "breakdown_ms_per_step": None if ss_steps == 0 else {name: round(single[kid][0] / max(single[kid][1], 1), 4) for name, kid in
                                      (("msm_sort", _lib.K_MSM_SORT), ("msm_accumulate", _lib.K_MSM_ACCUMULATE), ("msm_reduce", _lib.K_MSM_REDUCE), ("ntt", _lib.K_NTT_PASS))},
            "breakdown_ms_per_step_overlapped": {"msm_sort": round(sort_ms / max(sort_cnt, 1), 4), "msm_accumulate": round(acc_avg_ms, 4),
                                                 "msm_reduce": round(red_ms / max(red_cnt, 1), 4), "ntt": round(ntt_ms / max(ntt_cnt, 1), 4),
                                                 "note": "the same regions inside the timed region, %d steps in flight: each is stretched by the other steps' kernels sharing the chip (they sum to more than a step)" % inflight},
            "single_stream": None if ss_steps == 0 else {"ms_per_step": round(ss_ms, 4), "steps": ss_steps,
                              "kernel_ms": {name: round(single[kid][0] / max(single[kid][1], 1), 4) for name, kid in
                                            (("msm_sort", _lib.K_MSM_SORT), ("msm_accumulate", _lib.K_MSM_ACCUMULATE), ("msm_reduce", _lib.K_MSM_REDUCE), ("ntt", _lib.K_NTT_PASS))},
                              "note": "same step, one at a time (not the metric): kernel times without overlap from the %d steps in flight" % inflight},
        }
        # NTT: like the MSM's roofline, from the transform running ALONE (the one-at-a-time pass: its passes back to back, nothing else on the chip);
        # the overlapped average of the timed region is printed beside it
        ntt_over_ms = ntt_ms / max(ntt_cnt, 1)
        ntt_alone_ms = single[_lib.K_NTT_PASS][0] / max(single[_lib.K_NTT_PASS][1], 1) if ss_steps else 0.0
        ntt_avg_ms, ntt_src = (ntt_alone_ms, "one step at a time (single_stream pass, %d transforms)" % ss_steps) if ntt_alone_ms > 0 else \
                              (ntt_over_ms, "timed region, %d steps in flight (overlap-inflated)" % inflight)
        passes = 1 if log_n <= 11 else (log_n + 7) // 8
        out["ntt_roofline"] = {"bound": "hbm", "kernel": "k_ntt_pass x %d" % passes, "achieved": round(NTT_BYTES_PER_ELEM * n / (ntt_avg_ms * 1e-3) / 1e9, 2) if ntt_avg_ms > 0 else 0.0,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_transform": NTT_BYTES_PER_ELEM * n,
                               "avg_transform_ms": round(ntt_avg_ms, 4), "avg_transform_ms_source": ntt_src, "avg_transform_ms_overlapped": round(ntt_over_ms, 4)}
        out["ntt_roofline"]["frac"] = round(out["ntt_roofline"]["achieved"] / HBM_PEAK_GBS, 5)
        # like the MSM, a 255-bit NTT on gfx950 is bound by wide-multiply issue, not by HBM: N/2 log2 N butterfly
        # multiplications + one output multiplication per element per pass (3 passes from 2^12 up to 2^24)
        ntt_muls = n * log_n // 2 + n * passes
        mads_per_mul = MADS_PER_FIELD_MUL.get(args.ntt_field, 0)
        out["ntt_roofline"]["valu"] = {"muls_per_launch": ntt_muls, "mads_per_mul": mads_per_mul, "peak_tmad_per_s": VMAD_PEAK_TMADS,
                                       "achieved_tmad_per_s": round(ntt_muls * mads_per_mul / (ntt_avg_ms * 1e-3) / 1e12, 2) if ntt_avg_ms > 0 else 0.0}
        out["ntt_roofline"]["valu"]["frac"] = round(out["ntt_roofline"]["valu"]["achieved_tmad_per_s"] / VMAD_PEAK_TMADS, 4)
        ss = out["single_stream"]["kernel_ms"] if out["single_stream"] else {"msm_accumulate": 0, "ntt": 0}
        if ss["msm_accumulate"] > 0 and ss["ntt"] > 0:
            out["single_stream"]["valu_frac"] = {"k_msm_accum0": round(MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows / (ss["msm_accumulate"] * 1e-3) / 1e12 / VMAD_PEAK_TMADS, 4),
                                                 "k_ntt_pass": round(ntt_muls * mads_per_mul / (ss["ntt"] * 1e-3) / 1e12 / VMAD_PEAK_TMADS, 4)}
        if os.environ.get("DEHALO_BENCH_FIRST_ATTEMPT"):      # supervise(): a first measuring process died by a signal; what is known about it travels in the line
            out["first_attempt"] = json.loads(os.environ["DEHALO_BENCH_FIRST_ATTEMPT"])
        out["rccl_world"] = rccl_world
        note("roofline inputs, cpu baseline, secondary numbers")
        if not args.no_cpu_baseline:
            # the CPU port on rank 0's own inputs: the reported baseline AND the parity check of the measured 2^20 configuration
            out["cpu_baseline"], want_msm, want_ntt = cpu_baseline(co, po, curve, field, log_n, bases_h, scalars_h, poly_h)
            got = ctx.to_affine(curve.id, ctx.download_tensor(d_out_all[args.warmup:args.warmup + args.steps].contiguous()))
            assert all(np.array_equal(g, want_msm) for g in got), "a timed step's MSM result differs from the CPU port's"
            chk = ctx.upload(poly_h)
            torch.cuda.synchronize()
            ctx.ntt_device(field.id, chk.data_ptr(), log_n, omega, 1, 0)
            ctx.synchronize()
            assert np.array_equal(ctx.download_tensor(chk), want_ntt), "the 2^%d NTT differs from the CPU port's" % log_n
            out["parity_of_timed_configuration"] = "all %d timed MSM results and the 2^%d NTT equal the CPU port's (checked after the timed region)" % (args.steps, log_n)
            if world == 1:
                out["secondary"] = secondary_numbers(pkg, co, ctx, curve, bases_h, scalars_h, log_n)
                # north star: "throughput on synthetic witnesses at k in {14, 17, 20} ... as fraction of the HBM roofline": three sizes x three scalar distributions x two NTT sizes
                note("by_k: MSM / NTT at k = 14, 17, 20, three scalar distributions")
                tables = {}
                for kk in (14, 17, 20):
                    if kk == log_n:
                        tables[kk] = (bases_h, bases)
                    else:
                        bh_k = co.synth_bases(curve.id, 1 << kk)
                        tables[kk] = (bh_k, ctx.register_bases(curve.id, bh_k, 0, True))
                out["by_k"] = by_k_numbers(pkg, po, co, ctx, curve, field, tables, out["cpu_baseline"]["cores"])
                for kk, (_, b) in tables.items():
                    if kk != log_n:
                        b.release()
        if args.proof_k > 0:
            with_cpu = not args.no_cpu_baseline
            verify = not args.no_verify
            out["proof"] = proof_numbers(pkg, co, po, ctx, args.proof_k, "delay_enc", with_cpu, verify)
        if args.proof_k > 0 and world == 1:      # (N > 1: rank 0 makes the headline proof only -- the other ranks wait at the final barrier meanwhile)
            out["proof_mod_pow"] = proof_numbers(pkg, co, po, ctx, 17, "mod_pow", with_cpu, verify)              # BASELINE configs[2]
            out["proof_pose_enc"] = proof_numbers(pkg, co, po, ctx, 11, "pose_enc", with_cpu, verify)            # BASELINE configs[0]
            # north star: k in {14, 17, 20}: each byte-compared with the CPU restatement's proof (k = 20: one CPU proof, about a minute) and pairing-checked
            out["proof_other_k"] = [proof_numbers(pkg, co, po, ctx, k, "delay_enc", with_cpu, verify, reps=3) for k in (14, 20) if k != args.proof_k]
        if batch is not None:
            out["batch_proofs"] = batch
        out["log_n"] = log_n
        if args.full_out:      # the verbose record, sentences included (scratch: copied into profiles/ by hand when it is the one to keep)
            try:
                os.makedirs(os.path.dirname(args.full_out) or ".", exist_ok=True)
                with open(args.full_out, "w") as fh:
                    json.dump(out, fh, indent=1)
            except OSError as e:
                note("verbose record not written: %s" % e)
        print(compact_line(out), flush=True)

    bases.release()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
