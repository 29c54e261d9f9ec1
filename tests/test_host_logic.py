"""CPU suite: host-side mirror logic (constants, EvaluationDomain setup, argument checks)."""
import numpy as np
import pytest


def test_field_specs_match_oracle_constants(pkg, po):
    for name, spec in pkg.fields.FIELDS.items():
        f = po.FIELDS[name]
        assert spec.p == f.p and spec.id == po.FIELD_IDS[name]
        assert spec.two_adicity == f.S and spec.root_of_unity == f.root_of_unity
        assert spec.zeta == po.zeta(f)
        for x in (0, 1, f.p - 1, 12345678901234567890):
            assert po.from_limbs64(spec.encode(x)) == f.to_mont(x)
            assert spec.decode(spec.encode(x)) == x % f.p
    for name, c in pkg.fields.CURVES.items():
        oc = po.CURVES[name]
        assert c.id == po.CURVE_IDS[name] and c.base.p == oc.base.p and c.scalar.p == oc.scalar.p and c.b == oc.b


def test_upstream_zeta_literals(pkg):
    """F::ZETA pinned as the literal constants of halo2curves' bn256::{Fr,Fq} and pasta_curves' Fp / Fq [UPSTREAM; ADVICE r1]:
    bn256::Fr::ZETA is the 192-bit root 0xb3c4d79d...90dd (= 7^((r-1)/3)), not its square."""
    F = pkg.fields
    want = {"bn254_fr": 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD,
            "bn254_fq": 0x30644E72E131A0295E6DD9E7E0ACCCB0C28F069FBB966E3DE4BD44E5607CFD48,
            "pasta_fp": 0x12CCCA834ACDBA712CAAD5DC57AAB1B01D1F8BD237AD31491DAD5EBDFDFE4AB9,
            "pasta_fq": 0x06819A58283E528E511DB4D81CF70F5A0FED467D47C033AF2AA9D2E050AA0E4F}
    for name, z in want.items():
        f = F.FIELDS[name]
        assert f.zeta == z and z != 1 and pow(z, 3, f.p) == 1
    assert F.BN254_FR.zeta == pow(7, (F.BN254_FR.p - 1) // 3, F.BN254_FR.p)


@pytest.mark.parametrize("fname,j,k", [("bn254_fr", 5, 17), ("bn254_fr", 3, 11), ("pasta_fp", 5, 14), ("pasta_fq", 4, 6)])
def test_domain_setup_matches_oracle(pkg, po, fname, j, k):
    class NoCtx:  # the constructor performs no device work
        pass
    d = pkg.EvaluationDomain(NoCtx(), pkg.fields.FIELDS[fname], j, k)
    o = po.Domain(po.FIELDS[fname], k, j)
    assert (d.n, d.extended_k, d.quotient_poly_degree) == (o.n, o.extended_k, o.quotient_poly_degree)
    assert (d.omega, d.omega_inv, d.extended_omega, d.extended_omega_inv) == (o.omega, o.omega_inv, o.ext_omega, o.ext_omega_inv)
    assert (d.ifft_divisor, d.extended_ifft_divisor, d.g_coset, d.g_coset_inv) == (o.ifft_divisor, o.ext_ifft_divisor, o.g_coset, o.g_coset_inv)
    assert d.extended_len() == 1 << o.extended_k
    # delay_enc shape: degree 5 -> extended = 4n; pose_enc: degree 3 -> 2n
    if j == 5:
        assert d.extended_k == k + 2
    if j == 3:
        assert d.extended_k == k + 1


def test_domain_rejects_oversized_extension(pkg):
    class NoCtx:
        pass
    with pytest.raises(ValueError):
        pkg.EvaluationDomain(NoCtx(), pkg.fields.BN254_FR, 5, 27)  # 2^29 > two-adicity 28


def test_bench_replaces_a_measuring_process_killed_by_a_signal_once_only_when_asked():
    """bench.py measures in-process by default: a measuring process killed by a signal is an exit code the caller sees (no silent second attempt).  With --supervise
    it measures in a child process; a child that exits by itself is final (here: no GPU -> exit 1 with the reason), a child killed by a signal is
    replaced ONCE (DEHALO_BENCH_SELFTEST_KILL = the attempt that kills itself)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    run = lambda extra, *flags: subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", *flags], capture_output=True, text=True, timeout=600, env=dict(env, **extra))
    plain = run({})
    assert "one more" not in plain.stderr.lower()
    if plain.returncode == 0:
        pytest.skip("a GPU is present: the no-device exit path is not reachable")
    assert plain.returncode == 1 and "needs an MI355X" in plain.stderr
    # default: the signal ends the run (the shell's 128 + signal convention is subprocess's negative return code), nothing is started again
    died = run({"DEHALO_BENCH_SELFTEST_KILL": "1", "DEHALO_BENCH_ATTEMPT": "1"})
    assert died.returncode == -6 and "one more" not in died.stderr.lower() and died.stdout == ""
    once = run({"DEHALO_BENCH_SELFTEST_KILL": "1"}, "--supervise")
    assert once.returncode == 1 and "killed by signal 6" in once.stderr and "needs an MI355X" in once.stderr and once.stdout == ""


def test_bench_line_fits_the_drivers_tail_and_ends_with_the_metric():
    """The line on stdout is compact_line(record): at most 6,000 characters for a full record (tests/golden/bench_record_verbose.json: round 4's, 14.4 KB verbose),
    BASELINE's own metric -- the k = 17 delay_enc proof, the batch, attempts -- LAST, and its three scalars inside `config`."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = json.load(open(os.path.join(root, "tests", "golden", "bench_record_verbose.json")))
    rec["first_attempt"] = {"signal": 6, "section": "proof delay_enc k = 17: proving", "stderr_tail": ["x" * 300] * 6}
    rec["attempts"] = 2
    # the north star's per-k section (bench.by_k_numbers): three sizes x (three scalar distributions + two transform sizes), at most 700 characters
    rec["by_k"] = [{"k": k, "msm_mpts": {"u": 123.4, "w": 234.5, "l": 345.6}, "msm_frac": {"u": 0.00148, "w": 0.00281, "l": 0.00415}, "ntt_ms": [0.0123, 0.0456],
                    "ntt_frac": [0.0123, 0.0234], "ok": True} for k in (14, 17, 20)]
    rec["roofline"]["traffic"] = 1646269061
    text = bench.compact_line(rec)
    assert len(text) <= bench.LINE_LIMIT == 6000, len(text)
    line = json.loads(text)
    assert [e["k"] for e in line["roofline"]["by_k"]] == [14, 17, 20] and len(json.dumps(line["roofline"]["by_k"], separators=(",", ":"))) <= 700
    assert line["roofline"]["traffic_src"].startswith("pmc ")
    keys = list(line)
    assert keys[-4:] == ["proof", "batch_proofs", "attempts", "first_attempt"], keys[-6:]
    assert keys.index("proof_other_k") < keys.index("proof")
    assert line["proof"]["k"] == 17 and line["proof"]["gpu_ms"] == rec["proof"]["gpu_ms"] and line["proof"]["phase_ms"] == rec["proof"]["gpu_phase_ms"]
    cfg = line["config"]
    assert cfg["delay_enc_k17_ms"] == rec["proof"]["gpu_ms"] and cfg["batch_proofs_per_s"] == rec["batch_proofs"]["proofs_per_s"] and cfg["attempts"] == 2
    assert "model" not in cfg and cfg["workload"].startswith("1 x MSM(2^20, pallas)")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert {p["k"] for p in line["proof_other_k"]} == {14, 20}
    assert max(len(v) for v in _strings(line)) <= 110          # no sentences: the longest string is config.workload
    # a record whose sections grew still fits: optional sections are dropped, the metric's are not
    rec["proof_other_k"] = rec["proof_other_k"] * 4
    text = bench.compact_line(rec)
    assert len(text) <= 6000 and "\"proof\":{" in text and "batch_proofs" in text


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` (the driver's N = 1 command shape with N changed, no RANK in the environment) starts torch.distributed.run itself, as a fresh
    child process: here, without a GPU, both ranks get as far as the device check and the launcher's exit code comes back."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--force-device", "0", "--proofs", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    if r.returncode == 0:
        import json
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["rccl_world"]["world_size_seen_by_rank"] == [2, 2]
        return
    assert "launch N > 1 with" not in r.stderr
    # the ranks were started by the launcher: its failure report is there, and at least one rank's own message (the launcher ends the other rank as soon as the
    # first has failed -- on a loaded machine before it has printed)
    assert r.stderr.count("needs an MI355X") >= 1 and ("ChildFailedError" in r.stderr or "torch.distributed.elastic" in r.stderr), r.stderr[-2000:]


def test_bench_first_attempt_travels_to_the_second():
    """supervise(): what is known about a measuring process that died (signal, last section marker, stderr tail) is handed to the one that replaces it, which
    prints it in its JSON line as "first_attempt"."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    err = "[bench 0.1 s] steps: preheat\nnoise\n[bench 12.0 s] proof delay_enc k = 17: proving\nMemory access fault by GPU node-2\n"
    assert bench._section_of(err) == "proof delay_enc k = 17: proving"
    assert bench._section_of("") == ""
