#!/usr/bin/env python3
"""bench.py -- one JSON line for BASELINE.json's metric on its configs[1]:
"Standalone 2^20 Pallas MSM + 2^20 pasta-Fp NTT on 1 MI355X".

A step = one pass of the hot path over one batch of synthetic input, inputs already resident
in HBM: one 2^20-term Pallas MSM (== best_multiexp / commit over a registered SRS) plus one
2^20-point NTT over pasta::Fp (== best_fft), then -- because the north star names it -- the
all-gather of the commitment vector across ranks (a no-op at N = 1).  One process per GPU;
N > 1 is launched by torch.distributed.run, units are independent per rank (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--no-cpu-baseline]
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
    ap.add_argument("--prover-k", type=int, default=17, help="also time the delay_enc-shaped MSM/NTT schedule at this k (0 = skip)")
    ap.add_argument("--prover-curve", default="bn254")
    ap.add_argument("--inflight", type=int, default=3, help="independent steps in flight, each on its own HIP stream / workspace")
    return ap.parse_args()


def cpu_baseline(co, po, curve, field, log_n, dist):
    """The oracle's C restatement of best_multiexp + best_fft (same chunk-per-thread split as
    upstream's rayon code) on this host's cores: a reported baseline, not the target."""
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
    n = 1 << log_n
    bases = co.synth_bases(curve.id, n)
    scalars = co.fill_scalars(curve.scalar.id, dist, n, 1)
    a = co.fill_scalars(field.id, "uniform", n, 2)
    omega = field.encode(po.FIELDS[field.name].omega(log_n))
    reps, t_msm, t_ntt = 0, 0.0, 0.0
    t_start = time.time()
    while reps < 3 and (reps == 0 or time.time() - t_start < 20.0):
        t0 = time.time()
        co.best_multiexp(curve.id, scalars, bases, cores)
        t1 = time.time()
        co.best_fft(field.id, a, omega, log_n, cores)
        t2 = time.time()
        t_msm += t1 - t0
        t_ntt += t2 - t1
        reps += 1
    step_s = (t_msm + t_ntt) / reps
    return {
        "value": round(n / step_s / 1e6, 4), "unit": "Mpoints/s", "cores": cores, "kind": "port",
        "sample": "%d x (2^%d-term MSM + 2^%d-point NTT), oracle/oracle.c best_multiexp+best_fft, %d threads" % (reps, log_n, log_n, cores),
        "msm_ms": round(1e3 * t_msm / reps, 2), "ntt_ms": round(1e3 * t_ntt / reps, 2),
    }


def prover_shape_numbers(pkg, co, po, ctx, k, curve_name, with_cpu, with_quotient=True):
    """The k~17 half of the metric: the MSM/NTT schedule of one delay_enc create_proof
    (31 MSM(n) + 24 iNTT(n) + 23 coset NTT(n -> 4n) + 1 iNTT(4n), phases separated by host
    syncs), synthetic device-resident columns; CPU = the same calls through oracle/oracle.c."""
    import numpy as np
    from dehalo2_amd import prover_shape as ps
    curve = pkg.fields.CURVES[curve_name]
    n = 1 << k
    g = co.synth_bases(curve.id, n)
    gl = g[::-1].copy()
    cols = ps.synthetic_columns(lambda fid, dist, m, seed: co.fill_scalars(fid, dist, m, seed), curve.scalar.id, k, 7)
    bg, bgl = ctx.register_bases(curve.id, g, 0, True), ctx.register_bases(curve.id, gl, 0, True)
    fill = lambda fid, dist, m, seed: co.fill_scalars(fid, dist, m, seed)
    if with_quotient:
        cols.update(ps.synthetic_proving_key(fill, curve.scalar, k, k + 2, 55))
    shape = ps.ProverShape(ctx, curve, k, bgl, bg, cols, with_quotient=with_quotient)
    shape.run()                                   # warm-up (twiddle tables, workspace)
    runs = [shape.run() for _ in range(5)]
    best = min(runs, key=lambda r: r.ms_total)
    overlapped = None
    if with_quotient:   # the same schedule with the NTTs on a second context (stream), overlapping the MSM phases
        ctx2 = pkg.Context(0)
        shape.run_overlapped(ctx2)
        overlapped = min(shape.run_overlapped(ctx2) for _ in range(5))
        ctx2.close()
    out = {"k": k, "curve": curve_name, "gpu_ms": round(best.ms_total, 3), "gpu_msm_ms": round(best.ms_msm, 3), "gpu_ntt_ms": round(best.ms_ntt, 3),
           "gpu_ms_ntt_overlapped": round(overlapped, 3) if overlapped is not None else None,
           "gpu_eval_h_ms": round(best.ms_eval_h, 3) if with_quotient else None,
           "gpu_arguments_ms": round(best.ms_arguments, 3) if with_quotient else None, "gpu_openings_ms": round(best.ms_openings, 3) if with_quotient else None,
           "schedule": "31 MSM(n) + 24 iNTT(n) + 23 coset-NTT(n->4n) + 1 iNTT(4n)%s; after every commit phase the commitments are converted to affine and copied to the host (transcript); columns resident in HBM"
                       % (" + 5 lookup permutations + 7 grand products + evaluate_h(4n: gates, 2 permutation sets, 5 lookups) + 72 eval_polynomial(n)" if with_quotient else "")}
    if with_cpu:
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
        f = curve.scalar
        d = shape.domain
        e = f.encode
        t0 = time.time()
        for name, cnt, _ in ps.MSM_PHASES:
            basis = gl if name in ("advice", "lookup_permuted", "grand_products") else g
            for i in range(cnt):
                co.best_multiexp(curve.id, cols[name][i], basis, cores)
        t1 = time.time()
        coeffs = [co.lagrange_to_coeff(f.id, cols["polys"][i], k, e(d.omega_inv), e(d.ifft_divisor), cores) for i in range(ps.N_INTT)]
        exts = [co.coeff_to_extended(f.id, coeffs[i], k, d.extended_k, e(d.extended_omega), e(d.g_coset), cores) for i in range(ps.N_COSET)]
        th = 0.0
        h = exts[0]
        if with_quotient:
            th0 = time.time()
            ch, ext_k, rs = cols["challenges"], d.extended_k, 1 << (d.extended_k - k)
            fixed, advice, L = [cols["pk_fixed"][i] for i in range(ps.N_FIXED)], exts[:5], cols["pk_l"]
            mg = ps.maingate_graph()
            h = co.graph_evaluate(f.id, f.encode_many(mg.constants), mg.rotations, mg.calculations, mg.num_intermediates, fixed, advice,
                                  [np.zeros((1 << ext_k, 4), dtype=np.uint64)], None, None, None, None, e(ch["y"]), ext_k, rs, None, cores)
            h = co.permutation_h(f.id, h, [exts[15], exts[16]], advice + [fixed[14]], [cols["pk_sigma"][i] for i in range(ps.N_SIGMA)], ps.PERM_CHUNK, ps.LAST_ROTATION,
                                 L[0], L[1], L[2], e(ch["beta"]), e(ch["gamma"]), e(ch["y"]), e(ch["delta"]), e(ch["beta"] * d.g_coset % f.p), e(d.extended_omega),
                                 ext_k, rs, cores)
            for i in range(ps.N_LOOKUPS):
                lg = ps.lookup_graph(i)
                tv = co.graph_evaluate(f.id, f.encode_many(lg.constants), lg.rotations, lg.calculations, lg.num_intermediates, fixed, advice, [], None, e(ch["beta"]),
                                       e(ch["gamma"]), e(ch["theta"]), None, ext_k, rs, None, cores)
                h = co.lookup_h(f.id, h, exts[17 + i], exts[5 + 2 * i], exts[6 + 2 * i], tv, L[0], L[1], L[2], e(ch["beta"]), e(ch["gamma"]), e(ch["y"]), ext_k, rs, cores)
            th = time.time() - th0
        co.extended_to_coeff(f.id, h, d.extended_k, e(d.extended_omega_inv), e(d.extended_ifft_divisor), e(d.g_coset), cores)
        t2 = time.time()
        tp = 0.0
        if with_quotient:
            tp0 = time.time()
            usable = n - 6
            for i in range(ps.N_LOOKUPS):
                co.permute_expression_pair(f.id, cols["lookup_inputs"][i], cols["lookup_table"], usable)
            for i in range(7):
                co.grand_product(f.id, cols["grand_products"][i], cols["lookup_permuted"][i])
            for _ in range(3):
                for i in range(ps.N_INTT):
                    co.eval_polynomial(f.id, coeffs[i], e(cols["challenges"]["y"]), cores)
            tp = time.time() - tp0
            t2 += tp
        out.update({"cpu_arguments_openings_ms": round(1e3 * tp, 1) if with_quotient else None, "cpu_ms": round(1e3 * (t2 - t0), 1), "cpu_msm_ms": round(1e3 * (t1 - t0), 1), "cpu_ntt_ms": round(1e3 * (t2 - t1 - th - tp), 1),
                    "cpu_eval_h_ms": round(1e3 * th, 1) if with_quotient else None, "cpu_cores": cores, "cpu_kind": "port (oracle/oracle.c)"})
    bg.release(); bgl.release()
    return out


def main():
    args = parse()
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend)  # "nccl" = RCCL over xGMI

    pkg = entry.load_package()
    po, co = entry.load_oracle()  # synthetic-input generators + the cpu_baseline leg only
    from dehalo2_amd import _lib, sharding

    curve = pkg.fields.CURVES[args.curve]
    field = pkg.fields.FIELDS[args.ntt_field]
    log_n, n = args.log_n, 1 << args.log_n

    # One context (= one HIP stream + one workspace) per step in flight: consecutive steps are
    # independent units (different columns / proofs), so step i+1's sort and bucket accumulation
    # overlap step i's latency-bound bucket reduction.  The SRS tables are shared.
    inflight = max(1, args.inflight)
    ctxs = [pkg.Context(local_rank) for _ in range(inflight)]
    ctx = ctxs[0]
    # synthetic SRS and witnesses; every rank gets its own scalar column
    bases_h = co.synth_bases(curve.id, n)
    scalars_h = co.fill_scalars(curve.scalar.id, args.dist, n, 1000 + rank)
    poly_h = co.fill_scalars(field.id, "uniform", n, 2000 + rank)
    bases = ctx.register_bases(curve.id, bases_h, args.window_bits, True)  # resident SRS tables
    d_scalars = torch.from_numpy(scalars_h.view(np.int64)).cuda()
    d_polys = [torch.from_numpy(poly_h.view(np.int64)).cuda() for _ in range(inflight)]
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

    for _ in range(args.warmup):
        step()
    gather_commitments(0, max(args.warmup, 1))
    # parity spot-check of the measured configuration, outside the timed region (rank 0, once)
    fence()
    if rank == 0 and log_n <= 14:
        got = ctx.to_affine(curve.id, d_out.cpu().numpy().view(np.uint64).reshape(1, 12))[0]
        want = co.to_affine(curve.id, co.best_multiexp(curve.id, scalars_h, bases_h, 4))
        assert np.array_equal(got, want), "bench MSM result differs from the oracle"

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
    ctx.timing_reset(); ctx.timing_enable(True)
    torch.cuda.synchronize()
    ts = time.perf_counter()
    for i in range(ss_steps):
        ctx.msm_device(bases, d_scalars.data_ptr(), n, 1, d_out_all[0].data_ptr(), 0)
        ctx.ntt_device(field.id, d_polys[0].data_ptr(), log_n, omega, 1, 0)
    ctx.synchronize()
    ss_ms = (time.perf_counter() - ts) * 1e3 / max(ss_steps, 1)
    ctx.timing_enable(False)
    single = collect([ctx]) if rank == 0 else None

    if rank == 0:
        def tsum(kid):
            return overlapped[kid]
        acc_ms, acc_cnt = tsum(_lib.K_MSM_ACCUMULATE)
        sort_ms, sort_cnt = tsum(_lib.K_MSM_SORT)
        red_ms, red_cnt = tsum(_lib.K_MSM_REDUCE)
        ntt_ms, ntt_cnt = tsum(_lib.K_NTT_PASS)
        acc_avg_ms = acc_ms / max(acc_cnt, 1)
        c_bits = args.window_bits or (16 if log_n >= 20 else 15 if log_n >= 17 else 13 if log_n >= 10 else max(6, log_n + 1))   # as dehalo_bases_register chooses it
        n_windows = -(-256 // c_bits)
        achieved = MSM_BYTES_PER_TERM * n / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written by tools/pmc_summary.py from rocprofv3 --pmc passes
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("log_n") == log_n and rec.get("curve") == args.curve:
                    traffic = rec.get("msm_accumulate_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "pallas_msm_mpoints_per_s_at_2^%d" % log_n if args.curve == "pallas" else "%s_msm_mpoints_per_s_at_2^%d" % (args.curve, log_n),
            "value": round(world * n * args.steps / elapsed / 1e6, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
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
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": MSM_BYTES_PER_TERM * n, "avg_kernel_ms": round(acc_avg_ms, 4),
                         "note": "MSM is integer-VALU-bound (group adds), not HBM-bound: a low HBM fraction is expected (SURVEY.md 8d)",
                         # the honest ceiling of this kernel: wide integer multiplies issued vs the measured v_mad_u64_u32 peak
                         "valu": {"mads_per_launch": MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows, "peak_tmad_per_s": VMAD_PEAK_TMADS,
                                  "achieved_tmad_per_s": round(MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows / (acc_avg_ms * 1e-3) / 1e12, 2) if acc_avg_ms > 0 else 0.0,
                                  "frac": round(MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows / (acc_avg_ms * 1e-3) / 1e12 / VMAD_PEAK_TMADS, 4) if acc_avg_ms > 0 else 0.0}},
            "breakdown_ms_per_step": {"msm_sort": round(sort_ms / max(sort_cnt, 1), 4), "msm_accumulate": round(acc_avg_ms, 4),
                                      "msm_reduce": round(red_ms / max(red_cnt, 1), 4), "ntt": round(ntt_ms / max(ntt_cnt, 1), 4)},
            "single_stream": None if ss_steps == 0 else {"ms_per_step": round(ss_ms, 4), "steps": ss_steps,
                              "kernel_ms": {name: round(single[kid][0] / max(single[kid][1], 1), 4) for name, kid in
                                            (("msm_sort", _lib.K_MSM_SORT), ("msm_accumulate", _lib.K_MSM_ACCUMULATE), ("msm_reduce", _lib.K_MSM_REDUCE), ("ntt", _lib.K_NTT_PASS))},
                              "note": "same step, one at a time (not the metric): kernel times without overlap from the %d steps in flight" % inflight},
            "ntt_roofline": {"bound": "hbm", "achieved": round(NTT_BYTES_PER_ELEM * n / (ntt_ms / max(ntt_cnt, 1) * 1e-3) / 1e9, 2) if ntt_ms > 0 else 0.0,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s"},
        }
        out["ntt_roofline"]["frac"] = round(out["ntt_roofline"]["achieved"] / HBM_PEAK_GBS, 5)
        # like the MSM, a 255-bit NTT on gfx950 is bound by wide-multiply issue, not by HBM: N/2 log2 N butterfly
        # multiplications + one output multiplication per element per pass (3 passes from 2^12 up to 2^24)
        ntt_avg_ms = ntt_ms / max(ntt_cnt, 1)
        passes = 1 if log_n <= 11 else (log_n + 7) // 8
        ntt_muls = n * log_n // 2 + n * passes
        mads_per_mul = MADS_PER_FIELD_MUL.get(args.ntt_field, 0)
        out["ntt_roofline"]["valu"] = {"muls_per_launch": ntt_muls, "mads_per_mul": mads_per_mul, "peak_tmad_per_s": VMAD_PEAK_TMADS,
                                       "achieved_tmad_per_s": round(ntt_muls * mads_per_mul / (ntt_avg_ms * 1e-3) / 1e12, 2) if ntt_avg_ms > 0 else 0.0}
        out["ntt_roofline"]["valu"]["frac"] = round(out["ntt_roofline"]["valu"]["achieved_tmad_per_s"] / VMAD_PEAK_TMADS, 4)
        ss = out["single_stream"]["kernel_ms"] if out["single_stream"] else {"msm_accumulate": 0, "ntt": 0}
        if ss["msm_accumulate"] > 0 and ss["ntt"] > 0:
            out["single_stream"]["valu_frac"] = {"k_msm_accum0": round(MADS_PER_MIXED_ADD.get(args.curve, 0) * n * n_windows / (ss["msm_accumulate"] * 1e-3) / 1e12 / VMAD_PEAK_TMADS, 4),
                                                 "k_ntt_pass": round(ntt_muls * mads_per_mul / (ss["ntt"] * 1e-3) / 1e12 / VMAD_PEAK_TMADS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(co, po, curve, field, log_n, args.dist)
        if world == 1 and args.prover_k > 0:
            out["prover_shape"] = prover_shape_numbers(pkg, co, po, ctx, args.prover_k, args.prover_curve, not args.no_cpu_baseline)
            # the other two sizes the north star names (GPU only: the CPU port takes ~1 min at k = 20)
            out["prover_shape_other_k"] = [prover_shape_numbers(pkg, co, po, ctx, k, args.prover_curve, False, with_quotient=(k < 20))
                                           for k in (14, 20) if k != args.prover_k]
        print(json.dumps(out), flush=True)

    bases.release()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
