"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full sizes --
directly plus through size-independent properties.  Bar: bit-exact (integer arithmetic).
MSM results are compared after to_affine(): the Jacobian representative legitimately depends
on summation order (as it does upstream with the rayon thread count)."""
import numpy as np
import pytest

from conftest import dec_point, enc_points, golden, random_graph as _random_graph

pytestmark = pytest.mark.gpu

FIELDS3 = ["bn254_fr", "pasta_fp", "pasta_fq"]


def test_native_library_is_the_loaded_one(pkg, ctx):
    import os
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(pkg.library_path()) in maps


# ---------------------------------------------------------------- field arithmetic
@pytest.mark.parametrize("fname", ["bn254_fr", "bn254_fq", "pasta_fp", "pasta_fq"])
def test_field_ops_vs_python(pkg, po, ctx, fname):
    f = po.FIELDS[fname]
    spec = pkg.fields.FIELDS[fname]
    p = f.p
    rng = po.Xoshiro(4321 + spec.id)
    vals = [0, 1, 2, p - 1, p - 2, (1 << 255) % p, f.R % p, (p - 1) // 2, (1 << 64) - 1, 1 << 128] + [rng.below(p) for _ in range(3000)]
    vb = list(reversed(vals))
    a, b = spec.encode_many(vals), spec.encode_many(vb)
    assert spec.decode_many(ctx.field_op(spec.id, "add", a, b)) == [(x + y) % p for x, y in zip(vals, vb)]
    assert spec.decode_many(ctx.field_op(spec.id, "sub", a, b)) == [(x - y) % p for x, y in zip(vals, vb)]
    assert spec.decode_many(ctx.field_op(spec.id, "mul", a, b)) == [(x * y) % p for x, y in zip(vals, vb)]
    assert spec.decode_many(ctx.field_op(spec.id, "mul", a, a)) == [(x * x) % p for x in vals]
    # the kernels' internal carry-free 9 x 29-bit representation (fp29.cuh), round-tripped through from_std / to_std
    assert np.array_equal(ctx.field_op(spec.id, "mul29", a, b), ctx.field_op(spec.id, "mul", a, b))
    assert np.array_equal(ctx.field_op(spec.id, "mul29", a, a), ctx.field_op(spec.id, "mul", a, a))
    nz = [x for x in vals if x][:200]
    assert spec.decode_many(ctx.field_op(spec.id, "inv", spec.encode_many(nz))) == [pow(x, -1, p) for x in nz]
    canon = np.stack([np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64) for x in vals])
    assert np.array_equal(ctx.field_op(spec.id, "to_mont", canon), a)
    assert np.array_equal(ctx.field_op(spec.id, "from_mont", a), canon)


def test_field_mul_vs_c_oracle_large(pkg, co, ctx):
    for fid in range(4):
        a = co.fill_scalars(fid, "uniform", 1 << 16, 21)
        b = co.fill_scalars(fid, "uniform", 1 << 16, 22)
        for op in ("mul", "add", "sub"):
            assert np.array_equal(ctx.field_op(fid, op, a, b), co.field_op(fid, op, a, b)), (fid, op)
        assert np.array_equal(ctx.field_op(fid, "mul29", a, b), co.field_op(fid, "mul", a, b)), (fid, "mul29")


def test_poseidon_kats_through_gpu_field_ops(pkg, po, ctx):
    """The reference's known-answer vectors (src/poseidon/permutation.rs:154-158,190-196),
    with every field mul/add executed by the HIP kernels: pins the device bn256::Fr arithmetic
    to the reference's own fixture."""
    f = po.BN254_FR
    spec = pkg.fields.BN254_FR
    mul = lambda a, b: spec.decode(ctx.field_op(spec.id, "mul", spec.encode(a).reshape(1, 4), spec.encode(b).reshape(1, 4))[0])
    add = lambda a, b: spec.decode(ctx.field_op(spec.id, "add", spec.encode(a).reshape(1, 4), spec.encode(b).reshape(1, 4))[0])
    kat = golden("poseidon_kat")[0]
    out = po.poseidon_permute_ref(f, kat["input"], kat["r_f"], kat["r_p"], mul=mul, add=add)
    assert out == [int(x) for x in kat["expected"]]
    # the same vectors through op 6: the carry-free 9 x 29-bit multiplier (fp29.cuh) that every hot loop uses
    mul29 = lambda a, b: spec.decode(ctx.field_op(spec.id, "mul29", spec.encode(a).reshape(1, 4), spec.encode(b).reshape(1, 4))[0])
    for kat in golden("poseidon_kat"):
        out = po.poseidon_permute_ref(f, kat["input"], kat["r_f"], kat["r_p"], mul=mul29, add=add)
        assert out == [int(x) for x in kat["expected"]]


# ---------------------------------------------------------------- NTT
def test_ntt_golden(pkg, ctx):
    for v in golden("ntt"):
        spec = pkg.fields.FIELDS[v["field"]]
        a = spec.encode_many([int(x, 16) for x in v["input"]])
        out = pkg.best_fft(ctx, spec, a, spec.encode(int(v["omega"], 16)), v["log_n"])
        assert spec.decode_many(out) == [int(x, 16) for x in v["output"]], (v["field"], v["log_n"])


def test_domain_golden(pkg, ctx):
    for v in golden("domain"):
        spec = pkg.fields.FIELDS[v["field"]]
        d = pkg.EvaluationDomain(ctx, spec, v["j"], v["k"])
        assert d.g_coset == int(v["zeta"], 16)
        coeffs = d.lagrange_to_coeff(spec.encode_many([int(x, 16) for x in v["lagrange"]]))
        assert spec.decode_many(coeffs) == [int(x, 16) for x in v["coeffs"]]
        ext = d.coeff_to_extended(coeffs)
        assert spec.decode_many(ext) == [int(x, 16) for x in v["extended"]]
        back = d.extended_to_coeff(ext)
        assert spec.decode_many(back) == [int(x, 16) for x in v["back"]]


@pytest.mark.parametrize("fname", FIELDS3)
@pytest.mark.parametrize("log_n", [0, 1, 2, 4, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
def test_ntt_vs_c_oracle(pkg, po, co, ctx, fname, log_n):
    spec = pkg.fields.FIELDS[fname]
    f = po.FIELDS[fname]
    a = co.fill_scalars(spec.id, "uniform", 1 << log_n, 100 + log_n)
    for w in (f.omega(log_n), f.inv(f.omega(log_n))):  # forward and inverse roots
        omega = spec.encode(w)
        got = pkg.best_fft(ctx, spec, a, omega, log_n)
        want = co.best_fft(spec.id, a, omega, log_n, 4)
        assert np.array_equal(got, want)


def test_ntt_full_twiddle_table(pkg, po, co, fname="bn254_fr"):
    """Transforms up to the context's "ntt_full_table_log" keep all N powers of omega (no negation of the upper half): switched on here
    at sizes the C restatement checks, forward and inverse roots, plus a padded coeff_to_extended."""
    spec = pkg.fields.FIELDS[fname]
    f = po.FIELDS[fname]
    ctx = pkg.Context(0)
    ctx.set_tuning("ntt_full_table_log", 24)
    for log_n in (12, 17):
        a = co.fill_scalars(spec.id, "uniform", 1 << log_n, 900 + log_n)
        for w in (f.omega(log_n), f.inv(f.omega(log_n))):
            omega = spec.encode(w)
            assert np.array_equal(pkg.best_fft(ctx, spec, a, omega, log_n), co.best_fft(spec.id, a, omega, log_n, 8))
    d = pkg.EvaluationDomain(ctx, spec, 5, 12)
    coeffs = co.fill_scalars(spec.id, "uniform", d.n, 77)
    e = spec.encode
    assert np.array_equal(d.coeff_to_extended(coeffs), co.coeff_to_extended(spec.id, coeffs, 12, d.extended_k, e(d.extended_omega), e(d.g_coset), 8))
    with pytest.raises(pkg.DehaloError):
        ctx.set_tuning("ntt_full_table_log", 31)


@pytest.mark.parametrize("fname,log_n", [("pasta_fp", 20), ("bn254_fr", 20), ("pasta_fq", 19), ("pasta_fp", 22), ("bn254_fr", 24)])
def test_ntt_full_size(pkg, po, co, ctx, fname, log_n):
    """BASELINE sizes: direct comparison with the C restatement (2^20 takes it ~1 s) and
    the round trip NTT(omega) then NTT(omega^-1) * n^-1 = id."""
    spec = pkg.fields.FIELDS[fname]
    f = po.FIELDS[fname]
    n = 1 << log_n
    a = co.fill_scalars(spec.id, "uniform", n, 7 + log_n)
    omega = spec.encode(f.omega(log_n))
    got = pkg.best_fft(ctx, spec, a, omega, log_n)
    if log_n <= 22:
        assert np.array_equal(got, co.best_fft(spec.id, a, omega, log_n, 16))
    else:
        # spot-check 3 outputs against the definition via Horner in the C field ops is too slow; use
        # linearity + round trip instead (properties that do not depend on n)
        b = co.fill_scalars(spec.id, "uniform", n, 1007)
        gb = pkg.best_fft(ctx, spec, b, omega, log_n)
        gsum = pkg.best_fft(ctx, spec, co.field_op(spec.id, "add", a, b), omega, log_n)
        assert np.array_equal(gsum, co.field_op(spec.id, "add", got, gb))
    back = ctx.intt_scaled(spec.id, got, log_n, spec.encode(f.inv(f.omega(log_n))), spec.encode(f.inv(n)))
    assert np.array_equal(back, a)


@pytest.mark.parametrize("fname,j,k", [("bn254_fr", 5, 10), ("bn254_fr", 3, 11), ("bn254_fr", 5, 14), ("pasta_fp", 5, 12), ("pasta_fq", 5, 9), ("bn254_fr", 5, 17)])
def test_domain_vs_c_oracle(pkg, po, co, ctx, fname, j, k):
    spec = pkg.fields.FIELDS[fname]
    d = pkg.EvaluationDomain(ctx, spec, j, k)
    e = spec.encode
    a = co.fill_scalars(spec.id, "uniform", d.n, 300 + k)
    coeffs = d.lagrange_to_coeff(a)
    assert np.array_equal(coeffs, co.lagrange_to_coeff(spec.id, a, k, e(d.omega_inv), e(d.ifft_divisor), 8))
    ext = d.coeff_to_extended(coeffs)
    assert np.array_equal(ext, co.coeff_to_extended(spec.id, coeffs, k, d.extended_k, e(d.extended_omega), e(d.g_coset), 8))
    back = d.extended_to_coeff(ext)
    want = co.extended_to_coeff(spec.id, ext, d.extended_k, e(d.extended_omega_inv), e(d.extended_ifft_divisor), e(d.g_coset), 8)
    assert np.array_equal(back, want[: d.n * d.quotient_poly_degree])
    assert np.array_equal(back[: d.n], coeffs) and not back[d.n:].any()


def test_ntt_argument_errors(pkg, ctx):
    spec = pkg.fields.BN254_FR
    with pytest.raises(ValueError):
        ctx.ntt(spec.id, np.zeros((3, 4), np.uint64), 2, spec.encode(1))
    with pytest.raises(pkg.DehaloError):   # log_n beyond the two-adicity of bn256::Fr (28)
        ctx.ntt_device(spec.id, 0x1000, 29, spec.encode(1))
    with pytest.raises(pkg.DehaloError):
        ctx.ntt_device(pkg.fields.BN254_FQ.id, 0x1000, 2, spec.encode(1))  # Fq has two-adicity 1


# ---------------------------------------------------------------- MSM
def _golden_case(pkg, v, bases_json):
    spec = pkg.fields.CURVES[v["curve"]]
    if "points" in v:
        pts = [None if P is None else (int(P[0], 16), int(P[1], 16)) for P in v["points"]]
    else:
        pts = [(int(P[0], 16), int(P[1], 16)) for P in bases_json[v["curve"]][: v["n"]]]
    bases = enc_points(spec.base, pts)
    scalars = spec.scalar.encode_many([int(x, 16) for x in v["scalars"]])
    want = None if v["result"] is None else (int(v["result"][0], 16), int(v["result"][1], 16))
    return spec, bases, scalars, want


@pytest.mark.parametrize("precompute", [True, False])
def test_msm_golden(pkg, ctx, precompute):
    bases_json = golden("bases")
    for v in golden("msm"):
        spec, bases, scalars, want = _golden_case(pkg, v, bases_json)
        for c in (0, 4, 7):
            h = ctx.register_bases(spec.id, bases, c, precompute)
            got = dec_point(spec.base, ctx.to_affine(spec.id, ctx.msm(h, scalars))[0])
            h.release()
            assert got == want, (v["curve"], v["n"], v["dist"], c, precompute)
        got = dec_point(spec.base, ctx.to_affine(spec.id, pkg.best_multiexp(ctx, spec, scalars, bases))[0])
        assert got == want


@pytest.mark.parametrize("cname", ["bn254", "pallas", "vesta"])
@pytest.mark.parametrize("n,dist", [(1, "uniform"), (5, "witness"), (63, "uniform"), (64, "lookup"), (65, "witness"), (1000, "uniform"),
                                    (1 << 12, "witness"), (1 << 14, "uniform"), (1 << 14, "witness"), (1 << 14, "lookup"), ((1 << 14) + 17, "uniform")])
def test_msm_vs_c_oracle(pkg, co, ctx, cname, n, dist):
    spec = pkg.fields.CURVES[cname]
    bases = co.synth_bases(spec.id, n)
    scalars = co.fill_scalars(spec.scalar.id, dist, n, 500 + n)
    want = co.to_affine(spec.id, co.best_multiexp(spec.id, scalars, bases, 8))
    for precompute in (True, False):
        h = ctx.register_bases(spec.id, bases, 0, precompute)
        got = ctx.to_affine(spec.id, ctx.msm(h, scalars))[0]
        h.release()
        assert np.array_equal(got, want), (cname, n, dist, precompute)


@pytest.mark.parametrize("c", [4, 5, 8, 11, 13, 16, 17])
def test_msm_every_window_size(pkg, co, ctx, c):
    spec = pkg.fields.PALLAS
    n = 3000
    bases = co.synth_bases(spec.id, n)
    scalars = co.fill_scalars(spec.scalar.id, "uniform", n, 900 + c)
    # force scalars that exercise the digit carry chain: r - 1, 2^255-ish patterns, all-ones windows
    r = spec.scalar.p
    special = [r - 1, r - 2, (1 << 254) - 1, (1 << 253), int("f" * 62, 16) % r, int("8" * 63, 16) % r, 1, 0]
    scalars[: len(special)] = spec.scalar.encode_many(special)
    want = co.to_affine(spec.id, co.best_multiexp(spec.id, scalars, bases, 8))
    for precompute in ((True, False) if c <= 16 else (True,)):      # (17 bits: 2^16 buckets, precomputed rows only -- 15 windows for the Pasta scalar fields)
        h = ctx.register_bases(spec.id, bases, c, precompute)
        got = ctx.to_affine(spec.id, ctx.msm(h, scalars))[0]
        h.release()
        assert np.array_equal(got, want), (c, precompute)
    if c == 17:
        with pytest.raises(pkg.DehaloError):
            ctx.register_bases(spec.id, bases, 17, False)


def test_msm_prefix_and_batch(pkg, co, ctx):
    """commit() uses a prefix of the SRS; msm_batch = one launch for a phase's columns."""
    spec = pkg.fields.BN254
    n = 1 << 12
    bases = co.synth_bases(spec.id, n)
    params = pkg.Params(ctx, spec, 12, bases, g_lagrange=bases[::-1].copy())
    cols = [co.fill_scalars(spec.scalar.id, d, n, 40 + i) for i, d in enumerate(["uniform", "witness", "lookup", "witness", "uniform"])]
    got = ctx.to_affine(spec.id, params.commit_many(cols))
    for i, col in enumerate(cols):
        want = co.to_affine(spec.id, co.best_multiexp(spec.id, col, bases, 4))
        assert np.array_equal(got[i], want)
        assert np.array_equal(ctx.to_affine(spec.id, params.commit(col))[0], want)
    m = 1000  # prefix
    want = co.to_affine(spec.id, co.best_multiexp(spec.id, cols[0][:m], bases[:m], 4))
    assert np.array_equal(ctx.to_affine(spec.id, params.commit(cols[0][:m]))[0], want)
    want = co.to_affine(spec.id, co.best_multiexp(spec.id, cols[1], bases[::-1].copy(), 4))
    assert np.array_equal(ctx.to_affine(spec.id, params.commit_lagrange(cols[1]))[0], want)
    with pytest.raises(pkg.DehaloError):   # more scalars than bases (upstream: assert_eq! panic)
        ctx.msm(params.g, np.zeros((n + 1, 4), np.uint64))
    with pytest.raises(ValueError):
        pkg.best_multiexp(ctx, spec, cols[0][:10], bases[:9])
    # empty input -> identity
    assert not ctx.msm(params.g, np.zeros((0, 4), np.uint64)).reshape(3, 4)[2].any()
    params.release()


def test_msm_skewed_buckets(pkg, co, ctx):
    """0/1 columns and constant columns: one bucket receives almost every point."""
    spec = pkg.fields.PALLAS
    n = 1 << 15
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 0, True)
    r = spec.scalar.p
    rng = np.random.default_rng(1)
    for name, vals in (("bits", rng.integers(0, 2, n).tolist()), ("const", [5] * n), ("minus_one", [r - 1] * n), ("zeros", [0] * n)):
        col = np.zeros((n, 4), np.uint64)
        uniq = {v: spec.scalar.encode(v) for v in set(vals)}
        for i, v in enumerate(vals):
            col[i] = uniq[v]
        want = co.to_affine(spec.id, co.best_multiexp(spec.id, col, bases, 8))
        got = ctx.to_affine(spec.id, ctx.msm(h, col))[0]
        assert np.array_equal(got, want), name
    h.release()


def test_msm_every_merge_class(pkg, co, ctx):
    """Repeated scalars put 1 .. 6000 points into single buckets: every class of the partial-sum merge (one lane / quad, 32 lanes,
    one wave, a whole block) and the single-record copy, alone (quad mode) and in a batch whose bucket count selects lane mode."""
    spec = pkg.fields.BN254
    n = 1 << 15
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 0, True)
    rng = np.random.default_rng(11)

    def column(seed_val, mults):
        col = np.zeros((n, 4), np.uint64)
        i, v = 0, seed_val
        for m in mults:
            col[i:i + m] = spec.scalar.encode(v)          # one small value: one non-zero digit, m points in its bucket
            i, v = i + m, v + 1
        return col[rng.permutation(n)]                     # positions do not matter to the sort, only multiplicities

    cols = [column(3, [6000, 2500, 1200, 700, 300, 200, 90, 60, 33, 17, 9, 5, 3, 2, 1, 1]),
            column(1000, [40] * 300 + [7] * 500), column(5000, [130] * 100), column(9000, [2] * 9000), column(1, [n]),
            co.fill_scalars(spec.scalar.id, "uniform", n, 5)]
    want = [co.to_affine(spec.id, co.best_multiexp(spec.id, c, bases, 8)) for c in cols]
    for c, w in zip(cols, want):
        assert np.array_equal(ctx.to_affine(spec.id, ctx.msm(h, c))[0], w)
    assert len(cols) * (1 << (h.window_bits - 1)) > 65536 or h.window_bits < 15     # the batch is past the quad-mode bound at c = 15
    got = ctx.to_affine(spec.id, ctx.msm_batch(h, cols))
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    h.release()


@pytest.mark.parametrize("log_n, batch", [(11, 1), (11, 5), (14, 1), (14, 2), (14, 3), (14, 6), (16, 1)])
def test_msm_small_launch_geometry_and_populous_merge_classes(pkg, co, ctx, log_n, batch):
    """Dense columns at the sizes where the launch geometry adapts (round 4): sort blocks of 256 scalars and k_msm_bucket blocks of fewer slices for launches too small
    to fill the chip, 128-bucket blocks in the bucket reduction up to 2^13 buckets, and the merge's 9 .. 64-record class taking 8 / 4 / 2 / 1 quads a bucket by how many
    buckets it holds (2^14 uniform columns: ~20 partial sums in every one of 4096 buckets a column -- one column 8 quads, two 4, three 2, six 1).  Also with the
    host-wait knob at both ends."""
    spec = pkg.fields.BN254
    n = 1 << log_n
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 0, True)
    cols = [co.fill_scalars(spec.scalar.id, "uniform", n, 300 + 7 * j + log_n) for j in range(batch)]
    want = [co.to_affine(spec.id, co.best_multiexp(spec.id, c, bases, 8)) for c in cols]
    for spin in (0, 400):
        ctx.set_tuning("host_wait_spin_us", spin)
        got = ctx.to_affine(spec.id, ctx.msm_batch(h, cols))
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    # a prefix of the registered bases (commit() of a shorter polynomial): another slice count over the same table
    m = n - n // 3
    got = ctx.to_affine(spec.id, ctx.msm(h, cols[0][:m]))[0]
    assert np.array_equal(got, co.to_affine(spec.id, co.best_multiexp(spec.id, cols[0][:m], bases[:m], 8)))
    h.release()


@pytest.mark.parametrize("cname", ["pallas", "bn254"])
def test_msm_full_size_2_20(pkg, co, ctx, cname):
    """BASELINE config #2: 2^20 points.  Direct comparison with the C restatement of
    best_multiexp (all host threads), plus linearity MSM(a) + MSM(b) = MSM(a + b)."""
    spec = pkg.fields.CURVES[cname]
    n = 1 << 20
    bases = co.synth_bases(spec.id, n)
    a = co.fill_scalars(spec.scalar.id, "uniform", n, 1)
    b = co.fill_scalars(spec.scalar.id, "witness", n, 2)
    h = ctx.register_bases(spec.id, bases, 0, True)
    ja, jb = ctx.msm(h, a), ctx.msm(h, b)
    jab = ctx.msm(h, co.field_op(spec.scalar.id, "add", a, b))
    h.release()
    want = co.to_affine(spec.id, co.best_multiexp(spec.id, a, bases, 16))
    assert np.array_equal(ctx.to_affine(spec.id, ja)[0], want)
    # linearity, checked with the oracle's group law on two-term MSMs: [1]A + [1]B
    one = spec.scalar.encode(1)
    pa, pb = ctx.to_affine(spec.id, ja)[0], ctx.to_affine(spec.id, jb)[0]
    s = co.to_affine(spec.id, co.best_multiexp(spec.id, np.stack([one, one]), np.stack([pa, pb]), 1))
    assert np.array_equal(ctx.to_affine(spec.id, jab)[0], s)


@pytest.mark.parametrize("cname", ["pallas", "bn254", "vesta"])
def test_msm_17_bit_windows_at_2_20(pkg, co, ctx, cname):
    """Precomputed tables with 17-bit windows: 15 rows instead of 16, 2^16 buckets -- the histogram as packed 16-bit counters (128 KiB), 256 partitions in the
    scatter, a full three-level bucket reduction.  2^20 points, uniform and skewed scalars, and a batch of two columns, against the CPU port."""
    spec = pkg.fields.CURVES[cname]
    n = 1 << 20
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 17, True)
    assert (h.window_bits, h.windows) == (17, 15)
    for dist, seed in (("uniform", 11), ("lookup", 12)):
        sc = co.fill_scalars(spec.scalar.id, dist, n, seed)
        want = co.to_affine(spec.id, co.best_multiexp(spec.id, sc, bases, 16))
        assert np.array_equal(ctx.to_affine(spec.id, ctx.msm(h, sc))[0], want), (cname, dist)
    # a prefix (commit uses the first len bases) and edge scalars: r - 1, the all-ones windows, one value everywhere
    m = (1 << 17) + 5
    sc = co.fill_scalars(spec.scalar.id, "witness", m, 13)
    r = spec.scalar.p
    sc[:4] = spec.scalar.encode_many([r - 1, (1 << 254) - 1, int("1ffff" * 12, 16) % r, 0])
    want = co.to_affine(spec.id, co.best_multiexp(spec.id, sc, bases[:m], 16))
    assert np.array_equal(ctx.to_affine(spec.id, ctx.msm(h, sc))[0], want)
    h.release()


def test_msm_2_21_points_with_the_librarys_window(pkg, co, ctx):
    """Beyond the bench's size: 2^21 points, the window the library chooses (17 bits: the per-slice histogram stays packed 16-bit by cutting more slices -- 481 instead
    of 256), uniform scalars and a long run of one value, against the CPU port."""
    spec = pkg.fields.PALLAS
    n = 1 << 21
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 0, True)
    assert (h.window_bits, h.windows) == (17, 15)
    sc = co.fill_scalars(spec.scalar.id, "uniform", n, 2121)
    sc[1000:700000] = spec.scalar.encode(int("00ff" * 16, 16) % spec.scalar.p)
    want = co.to_affine(spec.id, co.best_multiexp(spec.id, sc, bases, 16))
    assert np.array_equal(ctx.to_affine(spec.id, ctx.msm(h, sc))[0], want)
    h.release()


def test_msm_sort_in_512_thread_workgroups_with_packed_histogram(pkg, co):
    """dehalo_ctx_set_tuning("msm_sort_block", 512): the sort's scalar-decoding kernels in 512-thread workgroups -- k_msm_hist counts into two 16-bit counters a word
    (64 KiB of LDS) whenever a block's scalars times its windows stay below 2^16, k_msm_part stages on eight waves.  The same results as the CPU port for sizes on
    both sides of the packed / plain switch, for both table kinds, three distributions, and the one-value column whose counts come closest to a 16-bit counter's
    limit (every scalar of a block in ONE bucket of every window: 2048 x 16 = 2^15 per counter)."""
    c5 = pkg.Context(0)
    c5.set_tuning("msm_sort_block", 512)
    with pytest.raises(pkg.DehaloError):
        c5.set_tuning("msm_sort_block", 256)
    spec = pkg.fields.PALLAS
    for n, dist in ((1, "uniform"), (777, "witness"), (1 << 12, "lookup"), ((1 << 14) + 17, "uniform"), (1 << 17, "witness"), (1 << 20, "uniform")):
        bases = co.synth_bases(spec.id, n)
        scalars = co.fill_scalars(spec.scalar.id, dist, n, 4242 + n)
        if n == 1 << 17:      # one value everywhere, all 16 digits the same bucket: the densest counter a block can see
            scalars[:] = spec.scalar.encode(int("0001" * 16, 16) % spec.scalar.p)
        want = co.to_affine(spec.id, co.best_multiexp(spec.id, scalars, bases, 16))
        for precompute in ((True, False) if n <= 1 << 17 else (True,)):
            h = c5.register_bases(spec.id, bases, 16 if n == 1 << 17 else 0, precompute)
            got = c5.to_affine(spec.id, c5.msm(h, scalars))[0]
            h.release()
            assert np.array_equal(got, want), (n, dist, precompute)
    c5.close()


def test_two_contexts_concurrently(pkg, co, ctx):
    """bench.py keeps several steps in flight, one context (stream + workspace) each, sharing one SRS."""
    import torch
    spec = pkg.fields.PALLAS
    n = 1 << 13
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 0, True)
    other = pkg.Context(0)
    cols = [co.fill_scalars(spec.scalar.id, "uniform", n, 70 + i) for i in range(4)]
    d_cols = [ctx.upload(c) for c in cols]
    d_outs = [torch.zeros((1, 12), dtype=torch.int64, device="cuda") for _ in cols]
    torch.cuda.synchronize()
    for rep in range(3):
        for i, (dc, do) in enumerate(zip(d_cols, d_outs)):
            (ctx if i % 2 == 0 else other).msm_device(h, dc.data_ptr(), n, 1, do.data_ptr(), 0)
    torch.cuda.synchronize()
    for c, do in zip(cols, d_outs):
        want = co.to_affine(spec.id, co.best_multiexp(spec.id, c, bases, 2))
        assert np.array_equal(ctx.to_affine(spec.id, do.cpu().numpy().view(np.uint64))[0], want)
    other.close()
    h.release()


def test_msm_randomised_configurations(pkg, co, ctx):
    """Seeded sweep over (curve, n, window bits, table mode, batch, scalar distribution): every
    combination goes through sort -> accumulate -> size-classed merge -> reduction and must match
    the oracle bit for bit."""
    rng = np.random.default_rng(20261003)
    curves = [pkg.fields.BN254, pkg.fields.PALLAS, pkg.fields.VESTA]
    dists = ["uniform", "witness", "lookup"]
    for trial in range(24):
        spec = curves[trial % 3]
        n = int(rng.integers(1, 6000)) if trial % 4 else int(rng.integers(6000, 40000))
        c = int(rng.choice([0, 4, 6, 9, 12, 14, 16]))
        precompute = bool(rng.integers(0, 2))
        batch = int(rng.choice([1, 1, 2, 5]))
        bases = co.synth_bases(spec.id, n)
        h = ctx.register_bases(spec.id, bases, c, precompute)
        cols = [co.fill_scalars(spec.scalar.id, dists[int(rng.integers(0, 3))], n, 7000 + 10 * trial + j) for j in range(batch)]
        m = int(rng.integers(1, n + 1))          # commit() over a prefix of the SRS
        got = ctx.to_affine(spec.id, ctx.msm_batch(h, [col[:m] for col in cols]))
        h.release()
        for j, col in enumerate(cols):
            want = co.to_affine(spec.id, co.best_multiexp(spec.id, col[:m], bases[:m], 4))
            assert np.array_equal(got[j], want), (trial, spec.name, n, m, c, precompute, batch, j)


@pytest.mark.parametrize("precompute", [True, False])
def test_msm_device_affine_matches_msm_then_to_affine(pkg, co, ctx, precompute):
    """dehalo_msm_device_affine: the points the transcript absorbs, normalised by the kernel that finishes the MSM -- equal to
    msm_device + to_affine and to the oracle, identity columns and the Jacobian output included."""
    import torch
    spec = pkg.fields.BN254
    n = 1 << 12
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 0, precompute)
    cols = np.stack([co.fill_scalars(spec.scalar.id, "uniform", n, 77), np.zeros((n, 4), np.uint64), co.fill_scalars(spec.scalar.id, "witness", n, 78)])
    with ctx.torch_stream():
        d = ctx.upload(cols)
        jac = torch.zeros((3, 12), dtype=torch.int64, device="cuda")
        aff = torch.full((3, 8), -1, dtype=torch.int64, device="cuda")
        aff_only = torch.full((3, 8), -1, dtype=torch.int64, device="cuda")
        ctx.msm_device_affine(h, d.data_ptr(), n, 3, jac.data_ptr(), aff.data_ptr(), 0)
        ctx.msm_device_affine(h, d.data_ptr(), n, 3, 0, aff_only.data_ptr(), 0)
        ctx.synchronize()
        got, got_only, got_jac = aff.cpu().numpy().view(np.uint64), aff_only.cpu().numpy().view(np.uint64), jac.cpu().numpy().view(np.uint64)
    for j in range(3):
        want = co.to_affine(spec.id, co.best_multiexp(spec.id, cols[j], bases, 4))
        assert np.array_equal(got[j], want) and np.array_equal(got_only[j], want), j
        assert np.array_equal(co.to_affine(spec.id, got_jac[j]), want), j
    assert not got[1].any()                                   # the empty commitment: identity = (0, 0)
    h.release()


def test_msm_single_row_sort_blocks_cover_several_windows(pkg, co, ctx):
    """Unregistered-style (single-row) tables at 2^17: every window keeps its own buckets and a sort block covers eight of them;
    batch of three, prefix lengths that leave ragged last slices, and the one-shot best_multiexp entry point."""
    spec = pkg.fields.BN254
    n = 1 << 17
    bases = co.synth_bases(spec.id, n)
    h = ctx.register_bases(spec.id, bases, 0, False)
    assert not h.precomputed and h.windows > 8
    cols = [co.fill_scalars(spec.scalar.id, d, n, 31 + j) for j, d in enumerate(("uniform", "witness", "lookup"))]
    for m in (n, n - 1, 100001):
        got = ctx.to_affine(spec.id, ctx.msm_batch(h, [c[:m] for c in cols]))
        for j, c in enumerate(cols):
            assert np.array_equal(got[j], co.to_affine(spec.id, co.best_multiexp(spec.id, c[:m], bases[:m], 8))), (m, j)
    h.release()
    got = ctx.to_affine(spec.id, ctx.best_multiexp(spec.id, cols[0], bases))[0]
    assert np.array_equal(got, co.to_affine(spec.id, co.best_multiexp(spec.id, cols[0], bases, 8)))


def test_ntt_batched_device_entry_points(pkg, po, co, ctx):
    """ntt_device / intt_scaled_device with batch > 1 (the prover runs its columns batched)."""
    import torch
    f = pkg.fields.BN254_FR
    of = po.FIELDS[f.name]
    for k, batch in ((5, 3), (11, 4), (13, 7), (16, 2)):
        polys = np.stack([co.fill_scalars(f.id, "uniform", 1 << k, 900 + 7 * k + b) for b in range(batch)])
        d = ctx.upload(polys)
        omega = f.encode(of.omega(k))
        ctx.ntt_device(f.id, d.data_ptr(), k, omega, batch, 0)
        ctx.synchronize()
        got = d.cpu().numpy().view(np.uint64)
        for b in range(batch):
            assert np.array_equal(got[b], co.best_fft(f.id, polys[b], omega, k, 2)), (k, b)
        # out of place (the prover's lagrange_to_coeff beside the committed values): the source stays, the result equals the in-place one
        out = torch.zeros_like(d)
        before = d.clone()
        ctx.lagrange_to_coeff_device(f.id, d.data_ptr(), out.data_ptr(), k, f.encode(of.inv(of.omega(k))), f.encode(of.inv(1 << k)), batch, 0)
        ctx.synchronize()
        assert torch.equal(d, before) and np.array_equal(out.cpu().numpy().view(np.uint64), polys)
        ctx.intt_scaled_device(f.id, d.data_ptr(), k, f.encode(of.inv(of.omega(k))), f.encode(of.inv(1 << k)), batch, 0)
        ctx.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint64), polys)


# ---------------------------------------------------------------- field-vector primitives
def test_poly_golden(pkg, ctx):
    for v in golden("poly"):
        spec = pkg.fields.FIELDS[v["field"]]
        if v["op"] == "eval_polynomial":
            poly = spec.encode_many([int(c, 16) for c in v["poly"]]) if v["poly"] else np.zeros((0, 4), dtype=np.uint64)
            got = pkg.eval_polynomial(ctx, spec, poly, spec.encode(int(v["point"], 16)))
            assert spec.decode(got) == int(v["result"], 16), (v["field"], len(v["poly"]), v["point"])
        elif v["op"] == "batch_invert":
            got = pkg.batch_invert(ctx, spec, spec.encode_many([int(c, 16) for c in v["values"]]))
            assert spec.decode_many(got) == [int(c, 16) for c in v["result"]]
        else:
            got = pkg.grand_product(ctx, spec, spec.encode_many([int(c, 16) for c in v["num"]]), spec.encode_many([int(c, 16) for c in v["den"]]))
            assert spec.decode_many(got) == [int(c, 16) for c in v["result"]]


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_eval_polynomial_vs_c_oracle(pkg, co, ctx, fid):
    x = co.fill_scalars(fid, "uniform", 1, 900 + fid)[0]
    # tile edges (2048 coefficients per block), two and three tree levels
    for n in (1, 7, 8, 9, 2047, 2048, 2049, 5000, (1 << 16) + 3, 1 << 20, (1 << 22) + 5):
        c = co.fill_scalars(fid, "uniform", n, 901 + n % 97)
        assert np.array_equal(ctx.eval_polynomial(fid, c, x), co.eval_polynomial(fid, c, x, 16)), (fid, n)
    zero = np.zeros(4, dtype=np.uint64)
    c = co.fill_scalars(fid, "uniform", 3000, 5)
    assert np.array_equal(ctx.eval_polynomial(fid, c, zero), c[0])                 # p(0) = c_0
    assert not ctx.eval_polynomial(fid, np.zeros((0, 4), dtype=np.uint64), x).any()  # empty polynomial -> 0


def test_eval_polynomial_device_batch(pkg, co, ctx):
    import torch
    fid, n, stride, batch = 0, 70000, 70016, 5
    cols = np.zeros((batch, stride, 4), dtype=np.uint64)
    for b in range(batch):
        cols[b, :n] = co.fill_scalars(fid, "uniform", n, 40 + b)
        cols[b, n:] = co.fill_scalars(fid, "uniform", stride - n, 50 + b)      # must be ignored
    x = co.fill_scalars(fid, "uniform", 1, 60)[0]
    d = ctx.upload(cols)
    out = torch.zeros((batch, 4), dtype=torch.int64, device="cuda")
    ctx.eval_polynomial_device(fid, d.data_ptr(), n, stride, batch, x, out.data_ptr(), 0)
    ctx.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for b in range(batch):
        assert np.array_equal(got[b], co.eval_polynomial(fid, cols[b, :n], x, 8)), b


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_batch_invert_vs_c_oracle(pkg, co, ctx, fid):
    for n in (1, 3, 1023, 1024, 1025, 4096, (1 << 16) + 7):
        v = co.fill_scalars(fid, "uniform", n, 300 + n % 89)
        v[:: 5] = 0 if n > 3 else v[:: 5]                                       # zeros stay zero
        if n > 2000:
            v[1024:2048] = 0                                                     # a whole block of zeros
        assert np.array_equal(ctx.batch_invert(fid, v), co.batch_invert(fid, v)), (fid, n)
    assert ctx.batch_invert(fid, np.zeros((0, 4), dtype=np.uint64)).shape == (0, 4)


def test_batch_invert_full_size_property(pkg, co, ctx):
    """2^20 elements: v * v^-1 = 1 through the element-wise kernel, and spot rows against the oracle."""
    fid, n = 0, 1 << 20
    v = co.fill_scalars(fid, "uniform", n, 31337)
    inv = ctx.batch_invert(fid, v)
    prod = ctx.field_op(fid, "mul", v, inv)
    one = ctx.field_op(fid, "to_mont", np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
    assert (prod == one).all()
    assert np.array_equal(inv[:4096], co.field_op(fid, "inv", v[:4096]))


@pytest.mark.parametrize("fid", [0, 2, 3])
def test_grand_product_vs_c_oracle(pkg, co, ctx, fid):
    for n in (1, 4, 1023, 1024, 1025, 5000, (1 << 18) + 11, 1 << 20):
        num = co.fill_scalars(fid, "uniform", n, 400 + n % 83)
        den = co.fill_scalars(fid, "uniform", n, 500 + n % 83)
        assert np.array_equal(ctx.grand_product(fid, num, den), co.grand_product(fid, num, den)), (fid, n)
    # a permutation-argument-shaped case: num is a permutation of den -> the product telescopes back to 1
    n = 1 << 16
    den = co.fill_scalars(fid, "uniform", n, 7)
    num = den[np.random.default_rng(5).permutation(n)]
    z = ctx.grand_product(fid, num, den)
    last = ctx.field_op(fid, "mul", ctx.field_op(fid, "mul", z[-1:], num[-1:]), ctx.field_op(fid, "inv", den[-1:]))
    one = ctx.field_op(fid, "to_mont", np.array([[1, 0, 0, 0]], dtype=np.uint64))
    assert np.array_equal(z[0], one[0]) and np.array_equal(last, one)
    with pytest.raises(ValueError):
        pkg.grand_product(ctx, pkg.fields.FIELDS["bn254_fr"], num[:5], den[:4])


def test_prefix_product_device_in_place(pkg, co, ctx):
    import torch
    fid, n = 2, 100003
    v = co.fill_scalars(fid, "uniform", n, 8)
    d = ctx.upload(v)
    ctx.prefix_product_device(fid, d.data_ptr(), n, d.data_ptr(), 0)
    ctx.synchronize()
    ones = ctx.field_op(fid, "to_mont", np.tile(np.array([[1, 0, 0, 0]], dtype=np.uint64), (n, 1)))
    assert np.array_equal(d.cpu().numpy().view(np.uint64), co.grand_product(fid, v, ones))


# ---------------------------------------------------------------- quotient numerator
def _dev(spec, vals):
    """canonical ints -> device tensor (Montgomery limbs) through the library's staged upload (keygen.to_device: the per-device transfer context)"""
    from dehalo2_amd import keygen
    return keygen.to_device(spec.encode_many(list(vals)))


def _host(spec, t):
    return spec.decode_many(t.cpu().numpy().view(np.uint64))


def _run_graph(pkg, ctx, spec, g, env, log_rows, rot_scale, previous):
    import torch
    ev = pkg.evaluation
    ge = ev.GraphEvaluator(constants=list(g["constants"]), rotations=list(g["rotations"]), calculations=list(g["calcs"]), num_intermediates=g["num_intermediates"])
    cg = ge.compile(ctx, spec)
    cols = {k: [_dev(spec, c) for c in env[k]] for k in ("fixed", "advice", "instance")}
    prev = _dev(spec, previous) if previous is not None else None
    out = torch.zeros((1 << log_rows, 4), dtype=torch.int64, device="cuda")
    cg.evaluate_device([t.data_ptr() for t in cols["fixed"]], [t.data_ptr() for t in cols["advice"]], [t.data_ptr() for t in cols["instance"]], env["challenges"],
                       env["beta"], env["gamma"], env["theta"], env["y"], log_rows, rot_scale, prev.data_ptr() if prev is not None else 0, out.data_ptr())
    ctx.synchronize()
    res = _host(spec, out)
    if prev is not None:   # in place: previous == out (custom gates accumulate into `values`)
        cg.evaluate_device([t.data_ptr() for t in cols["fixed"]], [t.data_ptr() for t in cols["advice"]], [t.data_ptr() for t in cols["instance"]], env["challenges"],
                           env["beta"], env["gamma"], env["theta"], env["y"], log_rows, rot_scale, prev.data_ptr(), prev.data_ptr())
        ctx.synchronize()
        assert _host(spec, prev) == res
    cg.release()
    return res


@pytest.mark.parametrize("fname,seed,n_calcs,long_lived", [("bn254_fr", 1, 12, 0), ("bn254_fr", 2, 60, 0), ("pasta_fp", 3, 40, 30), ("pasta_fq", 4, 150, 20),
                                                           ("bn254_fq", 5, 25, 14)])
def test_graph_evaluate_random_programs(pkg, po, ctx, fname, seed, n_calcs, long_lived):
    f, spec = po.FIELDS[fname], pkg.fields.FIELDS[fname]
    rng = po.Xoshiro(0xE7A1 + seed)
    log_rows, rot_scale = 7, 4
    rows = 1 << log_rows
    nf, na, ni, nc = 3, 4, 1, 2
    col = lambda: [rng.below(f.p) for _ in range(rows)]
    env = {"fixed": [col() for _ in range(nf)], "advice": [col() for _ in range(na)], "instance": [col() for _ in range(ni)],
           "challenges": [rng.below(f.p) for _ in range(nc)], "beta": rng.below(f.p), "gamma": rng.below(f.p), "theta": rng.below(f.p), "y": rng.below(f.p)}
    g = _random_graph(po, f, rng, n_calcs, nf, na, ni, nc, long_lived)
    previous = col()
    assert _run_graph(pkg, ctx, spec, g, env, log_rows, rot_scale, previous) == po.graph_evaluate(f, g, env, rows, rot_scale, previous)
    assert _run_graph(pkg, ctx, spec, g, env, log_rows, rot_scale, None) == po.graph_evaluate(f, g, env, rows, rot_scale, None)


def test_graph_evaluate_maingate_shape(pkg, po, ctx):
    """The delay-encryption circuit's gate shape (5 advice a..e, selectors
    sa..se, s_mul_ab, s_mul_cd, s_next_e, s_constant, public input), built with upstream's
    de-duplicating builder and folded with y like Evaluator::new does for the gate polynomials."""
    ev = pkg.evaluation
    f, spec = po.BN254_FR, pkg.fields.BN254_FR
    rng = po.Xoshiro(77)
    g = ev.GraphEvaluator()
    adv = [g.column(ev.ADVICE, i) for i in range(5)]
    e_next = g.column(ev.ADVICE, 4, 1)
    fx = [g.column(ev.FIXED, i) for i in range(9)]
    inst = g.column(ev.INSTANCE, 0)
    terms = [g.add_calculation(ev.MUL, adv[i], fx[i]) for i in range(5)]
    terms.append(g.add_calculation(ev.MUL, g.add_calculation(ev.MUL, adv[0], adv[1]), fx[5]))
    terms.append(g.add_calculation(ev.MUL, g.add_calculation(ev.MUL, adv[2], adv[3]), fx[6]))
    terms.append(g.add_calculation(ev.MUL, e_next, fx[7]))
    terms.append(fx[8])
    terms.append(inst)
    acc = terms[0]
    for t in terms[1:]:
        acc = g.add_calculation(ev.ADD, acc, t)
    assert g.add_calculation(ev.MUL, adv[0], fx[0]) == terms[0]                  # de-duplicated like upstream
    g.add_calculation(ev.HORNER, (ev.PREVIOUS, 0, 0), (ev.Y, 0, 0), (acc,))      # values * y + gate
    log_rows, rot_scale = 10, 4
    rows = 1 << log_rows
    col = lambda: [rng.below(f.p) for _ in range(rows)]
    env = {"fixed": [col() for _ in range(9)], "advice": [col() for _ in range(5)], "instance": [col()], "challenges": [], "beta": None, "gamma": None,
           "theta": None, "y": rng.below(f.p)}
    og = {"constants": g.constants, "rotations": g.rotations, "calcs": g.calculations, "num_intermediates": g.num_intermediates}
    previous = col()
    oenv = dict(env, beta=0, gamma=0, theta=0)
    assert _run_graph(pkg, ctx, spec, og, env, log_rows, rot_scale, previous) == po.graph_evaluate(f, og, oenv, rows, rot_scale, previous)


def test_graph_create_rejects_bad_programs(pkg, po, ctx):
    ev = pkg.evaluation
    spec = pkg.fields.BN254_FR
    bad = [
        [(ev.ADD, (ev.CONSTANT, 9, 0), (ev.CONSTANT, 0, 0), (), 0)],               # constant out of range
        [(ev.ADD, (ev.INTERMEDIATE, 0, 0), (ev.CONSTANT, 0, 0), (), 0)],           # read before write
        [(ev.ADD, (ev.ADVICE, 0, 3), (ev.CONSTANT, 0, 0), (), 0)],                 # rotation index out of range
        [(9, (ev.CONSTANT, 0, 0), (ev.CONSTANT, 0, 0), (), 0)],                    # unknown op
        [(ev.STORE, (ev.CONSTANT, 0, 0), (ev.CONSTANT, 0, 0), (), 5)],             # target out of range
    ]
    for calcs in bad:
        g = ev.GraphEvaluator(calculations=calcs, num_intermediates=1, rotations=[0])
        with pytest.raises(pkg.DehaloError):
            g.compile(ctx, spec)
    # a program that reads advice column 2 cannot run with only one advice column supplied
    import torch
    g = ev.GraphEvaluator(calculations=[(ev.STORE, (ev.ADVICE, 2, 0), (ev.CONSTANT, 0, 0), (), 0)], num_intermediates=1, rotations=[0])
    cg = g.compile(ctx, spec)
    out = torch.zeros((16, 4), dtype=torch.int64, device="cuda")
    with pytest.raises(pkg.DehaloError):
        cg.evaluate_device([], [out.data_ptr()], [], [], None, None, None, None, 4, 1, 0, out.data_ptr())
    cg.release()


@pytest.mark.parametrize("fname,ncols,chunk_len", [("bn254_fr", 6, 3), ("pasta_fp", 5, 3), ("pasta_fq", 4, 4), ("bn254_fr", 1, 2)])
def test_permutation_h_vs_oracle(pkg, po, ctx, fname, ncols, chunk_len):
    f, spec = po.FIELDS[fname], pkg.fields.FIELDS[fname]
    rng = po.Xoshiro(0x9E21 + ncols)
    k, ext_k = 6, 8
    rows, rot_scale = 1 << ext_k, 1 << (ext_k - k)
    nsets = (ncols + chunk_len - 1) // chunk_len
    col = lambda: [rng.below(f.p) for _ in range(rows)]
    z, cols, sigma = [col() for _ in range(nsets)], [col() for _ in range(ncols)], [col() for _ in range(ncols)]
    l0, l_last, l_active, values = col(), col(), col(), col()
    beta, gamma, y, delta = (rng.below(f.p) for _ in range(4))
    zeta, w = po.zeta(f), f.omega(ext_k)
    want = po.permutation_h(f, values, z, cols, sigma, chunk_len, -6, l0, l_last, l_active, beta, gamma, y, delta, zeta, w, rot_scale)
    d = {n: _dev(spec, v) for n, v in (("l0", l0), ("l_last", l_last), ("l_active", l_active), ("values", values))}
    dz, dc, ds = [_dev(spec, c) for c in z], [_dev(spec, c) for c in cols], [_dev(spec, c) for c in sigma]
    pkg.evaluation.permutation_h_device(ctx, spec, [t.data_ptr() for t in dz], [t.data_ptr() for t in dc], [t.data_ptr() for t in ds], chunk_len, -6,
                                        d["l0"].data_ptr(), d["l_last"].data_ptr(), d["l_active"].data_ptr(), beta, gamma, y, delta, zeta, w, ext_k, rot_scale,
                                        d["values"].data_ptr())
    ctx.synchronize()
    assert _host(spec, d["values"]) == want


@pytest.mark.parametrize("fname", ["bn254_fr", "pasta_fp"])
def test_lookup_h_vs_oracle(pkg, po, ctx, fname):
    f, spec = po.FIELDS[fname], pkg.fields.FIELDS[fname]
    rng = po.Xoshiro(0x100C)
    k, ext_k = 5, 7
    rows, rot_scale = 1 << ext_k, 1 << (ext_k - k)
    col = lambda: [rng.below(f.p) for _ in range(rows)]
    names = ("product", "a", "s", "tv", "l0", "l_last", "l_active", "values")
    h = {n: col() for n in names}
    beta, gamma, y = (rng.below(f.p) for _ in range(3))
    want = po.lookup_h(f, h["values"], h["product"], h["a"], h["s"], h["tv"], h["l0"], h["l_last"], h["l_active"], beta, gamma, y, rot_scale)
    d = {n: _dev(spec, h[n]) for n in names}
    pkg.evaluation.lookup_h_device(ctx, spec, d["product"].data_ptr(), d["a"].data_ptr(), d["s"].data_ptr(), d["tv"].data_ptr(), d["l0"].data_ptr(),
                                   d["l_last"].data_ptr(), d["l_active"].data_ptr(), beta, gamma, y, ext_k, rot_scale, d["values"].data_ptr())
    ctx.synchronize()
    assert _host(spec, d["values"]) == want


@pytest.mark.parametrize("fname,count", [("bn254_fr", 5), ("pasta_fp", 8), ("bn254_fr", 1)])
def test_lookup_h_batch_vs_oracle(pkg, po, ctx, fname, count):
    """Several lookups folded in one pass (dehalo_lookup_h_batch_device) = the oracle's lookup_h applied lookup after lookup;
    more than eight lookups, or lookups that do not share the Lagrange columns, are refused."""
    f, spec = po.FIELDS[fname], pkg.fields.FIELDS[fname]
    rng = po.Xoshiro(0x100D + count)
    k, ext_k = 6, 8
    rows, rot_scale = 1 << ext_k, 1 << (ext_k - k)
    col = lambda: [rng.below(f.p) for _ in range(rows)]
    shared = {n: col() for n in ("l0", "l_last", "l_active", "values")}
    each = [{n: col() for n in ("product", "a", "s", "tv")} for _ in range(count)]
    beta, gamma, y = (rng.below(f.p) for _ in range(3))
    want = shared["values"]
    for h in each:
        want = po.lookup_h(f, want, h["product"], h["a"], h["s"], h["tv"], shared["l0"], shared["l_last"], shared["l_active"], beta, gamma, y, rot_scale)
    ds = {n: _dev(spec, v) for n, v in shared.items()}
    de = [{n: _dev(spec, v) for n, v in h.items()} for h in each]
    tuples = [(d["product"].data_ptr(), d["a"].data_ptr(), d["s"].data_ptr(), d["tv"].data_ptr()) for d in de]
    pkg.evaluation.lookup_h_batch_device(ctx, spec, tuples, ds["l0"].data_ptr(), ds["l_last"].data_ptr(), ds["l_active"].data_ptr(), beta, gamma, y, ext_k,
                                         rot_scale, ds["values"].data_ptr())
    ctx.synchronize()
    assert _host(spec, ds["values"]) == want
    with pytest.raises(pkg.DehaloError):
        pkg.evaluation.lookup_h_batch_device(ctx, spec, tuples * 9, ds["l0"].data_ptr(), ds["l_last"].data_ptr(), ds["l_active"].data_ptr(), beta, gamma, y,
                                             ext_k, rot_scale, ds["values"].data_ptr())


# ---------------------------------------------------------------- lookup permutation
@pytest.mark.parametrize("fid", [0, 2])
def test_permute_expression_pair_vs_c_oracle(pkg, po, co, ctx, fid):
    rng = np.random.default_rng(99 + fid)
    for n, tsize, spread in ((1, 1, 1), (7, 3, 3), (1000, 256, 256), (5000, 256, 9), (1 << 16, 1 << 12, 1 << 12), ((1 << 17) - 6, 1 << 16, 50000)):
        base = co.fill_scalars(fid, "uniform", tsize, 11 + n % 91)
        base[::3] = 0
        base[::3, 0] = np.arange(0, tsize, 3, dtype=np.uint64)        # small canonical-looking Montgomery words: distinct, sort across limbs
        table = np.concatenate([base, np.repeat(base[:1], n - tsize, axis=0)])
        inputs = base[rng.integers(0, spread, size=n)]
        want = co.permute_expression_pair(fid, inputs, table, n)
        assert want is not None
        pi, pt = ctx.permute_expression_pair(fid, inputs, table, n)
        assert np.array_equal(pi, want[0]) and np.array_equal(pt, want[1]), (fid, n)
    # an input value that is not in the table: upstream's Err(ConstraintSystemFailure)
    table = co.fill_scalars(fid, "uniform", 300, 5)
    inputs = table[rng.integers(0, 300, size=300)].copy()
    inputs[17] = co.fill_scalars(fid, "uniform", 1, 6)[0]
    assert co.permute_expression_pair(fid, inputs, table, 300) is None
    with pytest.raises(pkg.DehaloError) as e:
        ctx.permute_expression_pair(fid, inputs, table, 300)
    assert e.value.code == -6
    # the context stays usable after the error
    pi, pt = ctx.permute_expression_pair(fid, table, table, 300)
    want = co.permute_expression_pair(fid, table, table, 300)
    assert np.array_equal(pi, want[0]) and np.array_equal(pt, want[1])


def test_grand_product_batch_shares_one_inversion(pkg, co, ctx):
    import torch
    fid, n, stride, batch = 0, 5000, 5008, 7
    num = np.stack([np.concatenate([co.fill_scalars(fid, "uniform", n, 60 + b), np.zeros((stride - n, 4), dtype=np.uint64)]) for b in range(batch)])
    den = np.stack([np.concatenate([co.fill_scalars(fid, "uniform", n, 70 + b), np.zeros((stride - n, 4), dtype=np.uint64)]) for b in range(batch)])
    den[3, 100] = 0                                                  # a zero denominator in one column must not disturb the others
    dn, dd = ctx.upload(num), ctx.upload(den)
    dz = torch.zeros_like(dn)
    ctx.grand_product_batch_device(fid, dn.data_ptr(), dd.data_ptr(), n, batch, stride, dz.data_ptr(), 0)
    ctx.synchronize()
    z = dz.cpu().numpy().view(np.uint64)
    for b in range(batch):
        assert np.array_equal(z[b, :n], co.grand_product(fid, num[b, :n], den[b, :n])), b
        assert not z[b, n:].any()                                    # the padding between columns is untouched


def test_permute_expression_pair_shared_tables_and_odd_sizes(pkg, co, ctx):
    """dehalo_permute_expression_pair_ptrs_device: lookups given as pointer lists; those whose TABLE pointer is the same column share one
    table sort (the reference's five range lookups).  Sizes around the merge sort's 2048-key tiles, tables of all-distinct full-width
    values, more lookups than one launch group holds (16), and an error in ONE lookup of the batch."""
    import torch
    fid = 0
    rng = np.random.default_rng(7)
    dev = lambda a: ctx.upload(a)
    for n in (1, 2, 2047, 2048, 2049, 4097, 6000, 100003):
        t_shared = co.fill_scalars(fid, "uniform", n, 500 + n % 89)                       # n distinct full-width values
        t_other = t_shared.copy()
        t_other[n // 2:] = t_other[0]                                                     # half of it one repeated value
        B = 18 if n == 4097 else 5
        ins = [(t_shared if y % 3 != 2 else t_other)[rng.integers(0, max(1, (n // (y + 1))), size=n)] for y in range(B)]
        d_shared, d_other = dev(t_shared), dev(t_other)
        d_ins = [dev(a) for a in ins]
        outs_i = [torch.zeros((n, 4), dtype=torch.int64, device="cuda") for _ in range(B)]
        outs_t = [torch.zeros((n, 4), dtype=torch.int64, device="cuda") for _ in range(B)]
        torch.cuda.synchronize()
        tabs = [d_shared if y % 3 != 2 else d_other for y in range(B)]
        ctx.permute_expression_pair_ptrs_device(fid, [a.data_ptr() for a in d_ins], [t.data_ptr() for t in tabs], n, [o.data_ptr() for o in outs_i],
                                                [o.data_ptr() for o in outs_t], 0)
        ctx.synchronize()
        for y in range(B):
            want = co.permute_expression_pair(fid, ins[y], t_shared if y % 3 != 2 else t_other, n)
            assert want is not None
            assert np.array_equal(outs_i[y].cpu().numpy().view(np.uint64), want[0]), (n, y)
            assert np.array_equal(outs_t[y].cpu().numpy().view(np.uint64), want[1]), (n, y)
    # one lookup of the batch has an input that is not in its table
    n = 3000
    table = co.fill_scalars(fid, "uniform", n, 5)
    good = table[rng.integers(0, n, size=n)]
    bad = good.copy()
    bad[n - 1] = co.fill_scalars(fid, "uniform", 1, 6)[0]
    d_t, d_g, d_b = dev(table), dev(good), dev(bad)
    o = [torch.zeros((n, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    with pytest.raises(pkg.DehaloError) as e:
        ctx.permute_expression_pair_ptrs_device(fid, [d_g.data_ptr(), d_b.data_ptr()], [d_t.data_ptr(), d_t.data_ptr()], n, [o[0].data_ptr(), o[1].data_ptr()],
                                                [o[2].data_ptr(), o[3].data_ptr()], 0)
    assert e.value.code == -6
    # the same call without its synchronisation: one flag per lookup on the device, the good lookup's columns as they should be
    status = torch.full((2,), 7, dtype=torch.int32, device="cuda")
    ctx.permute_expression_pair_ptrs_deferred_device(fid, [d_g.data_ptr(), d_b.data_ptr()], [d_t.data_ptr(), d_t.data_ptr()], n, [o[0].data_ptr(), o[1].data_ptr()],
                                                     [o[2].data_ptr(), o[3].data_ptr()], status.data_ptr(), 0)
    ctx.synchronize()
    st = status.cpu().numpy()
    assert st[0] == 0 and st[1] != 0
    want = co.permute_expression_pair(fid, good, table, n)
    assert np.array_equal(o[0].cpu().numpy().view(np.uint64), want[0]) and np.array_equal(o[2].cpu().numpy().view(np.uint64), want[1])
    with pytest.raises(pkg.DehaloError):      # an output aliasing an input is refused
        ctx.permute_expression_pair_ptrs_device(fid, [d_g.data_ptr()], [d_t.data_ptr()], n, [d_g.data_ptr()], [o[2].data_ptr()], 0)


def test_permute_expression_pair_tables_given_as_distinct_rows(pkg, co, ctx):
    """dehalo_permute_expression_pair_distinct_device: a table of fixed columns handed over as one representative row per distinct value + multiplicities (what the
    prover computes once per proving key); only the distinct keys are sorted.  Same A' and S' as the CPU restatement on the full table: few distinct values among
    many rows (the range tables' shape), up to the 2048-key tile, two tables in one call, one table shared by several lookups, a table with more than 2048 distinct
    values (the general path), values that repeat with wildly different multiplicities, an input that is not in the table."""
    import torch
    fid = 0
    rng = np.random.default_rng(23)
    dev = lambda a: ctx.upload(a)
    dev32 = lambda a: ctx.upload(np.ascontiguousarray(a, dtype=np.uint32))

    def distinct_rows(table):
        _, first, counts = np.unique(table, axis=0, return_index=True, return_counts=True)
        order = rng.permutation(len(first))                      # representative rows in no particular order
        return first[order].astype(np.uint32), counts[order].astype(np.uint32)

    def table_of(n, d, skew):
        vals = co.fill_scalars(fid, "uniform", d, 900 + d)
        if skew:   # one value fills most of the table (the padding row of a range table), the rest appear once or a few times
            idx = np.concatenate([np.arange(d), rng.integers(0, min(d, 3), size=max(0, n - d - n // 2)), np.zeros(n // 2, dtype=np.int64)])[:n]
            idx[:d] = np.arange(d) if d <= n else idx[:d]
        else:
            idx = rng.integers(0, d, size=n)
            idx[:min(d, n)] = np.arange(min(d, n))
        return vals[rng.permutation(idx)]

    for n, d1, d2 in ((5000, 339, 7), (2048, 2048, 1), (100003, 1500, 2047), (6000, 3000, 40), (1, 1, 1), (70000, 2, 300)):
        d1, d2 = min(d1, n), min(d2, n)
        t1, t2 = table_of(n, d1, True), table_of(n, d2, False)
        B = 5
        tabs_h = [t1, t2, t1, t1, t2]
        ins = [tabs_h[y][rng.integers(0, max(1, n // (y + 1)), size=n)] for y in range(B)]
        r1, m1 = distinct_rows(t1)
        r2, m2 = distinct_rows(t2)
        assert int(m1.sum()) == n and int(m2.sum()) == n
        d_t1, d_t2 = dev(t1), dev(t2)
        d_r = {1: dev32(r1), 2: dev32(r2)}
        d_m = {1: dev32(m1), 2: dev32(m2)}
        which = [1, 2, 1, 1, 2]
        d_ins = [dev(a) for a in ins]
        outs_i = [torch.zeros((n, 4), dtype=torch.int64, device="cuda") for _ in range(B)]
        outs_t = [torch.zeros((n, 4), dtype=torch.int64, device="cuda") for _ in range(B)]
        status = torch.full((B,), 9, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.permute_expression_pair_distinct_device(fid, [a.data_ptr() for a in d_ins], [(d_t1 if w == 1 else d_t2).data_ptr() for w in which], n,
                                                    [o.data_ptr() for o in outs_i], [o.data_ptr() for o in outs_t], [d_r[w].data_ptr() for w in which],
                                                    [d_m[w].data_ptr() for w in which], [len(r1) if w == 1 else len(r2) for w in which], status.data_ptr(), 0)
        ctx.synchronize()
        assert not status.cpu().numpy().any()
        for y in range(B):
            want = co.permute_expression_pair(fid, ins[y], tabs_h[y], n)
            assert want is not None
            assert np.array_equal(outs_i[y].cpu().numpy().view(np.uint64), want[0]), (n, d1, d2, y)
            assert np.array_equal(outs_t[y].cpu().numpy().view(np.uint64), want[1]), (n, d1, d2, y)
    # an input value that is not in the table: the synchronous form of the call says so
    n = 4000
    table = table_of(n, 50, True)
    r, m = distinct_rows(table)
    good = table[rng.integers(0, n, size=n)]
    bad = good.copy()
    bad[17] = co.fill_scalars(fid, "uniform", 1, 77)[0]
    d_t, d_g, d_b, d_rr, d_mm = dev(table), dev(good), dev(bad), dev32(r), dev32(m)
    o = [torch.zeros((n, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    with pytest.raises(pkg.DehaloError) as e:
        ctx.permute_expression_pair_distinct_device(fid, [d_g.data_ptr(), d_b.data_ptr()], [d_t.data_ptr()] * 2, n, [o[0].data_ptr(), o[1].data_ptr()], [o[2].data_ptr(), o[3].data_ptr()],
                                                    [d_rr.data_ptr()] * 2, [d_mm.data_ptr()] * 2, [len(r)] * 2, 0, 0)
    assert e.value.code == -6
    with pytest.raises(pkg.DehaloError):      # lookups of one table must describe it alike
        ctx.permute_expression_pair_distinct_device(fid, [d_g.data_ptr(), d_g.data_ptr()], [d_t.data_ptr()] * 2, n, [o[0].data_ptr(), o[1].data_ptr()], [o[2].data_ptr(), o[3].data_ptr()],
                                                    [d_rr.data_ptr()] * 2, [d_mm.data_ptr()] * 2, [len(r), len(r) - 1], 0, 0)


def test_permute_expression_pair_small_values(pkg, co, ctx):
    """Range-table shaped columns (values < 2^16): the sort skips the limbs that are zero everywhere."""
    fid, n = 0, 40000
    small = np.zeros((1 << 12, 4), dtype=np.uint64)
    small[:, 0] = np.arange(1 << 12, dtype=np.uint64) * 13 % 65521
    table_c = np.concatenate([small, np.repeat(small[:1], n - (1 << 12), axis=0)])
    inputs_c = small[np.random.default_rng(3).integers(0, 1 << 12, size=n)]
    table, inputs = ctx.field_op(fid, "to_mont", table_c), ctx.field_op(fid, "to_mont", inputs_c)
    want = co.permute_expression_pair(fid, inputs, table, n)
    pi, pt = ctx.permute_expression_pair(fid, inputs, table, n)
    assert np.array_equal(pi, want[0]) and np.array_equal(pt, want[1])


# ---------------------------------------------------------------- lifecycle
def test_context_lifecycle_releases_device_memory(pkg, co):
    """Contexts, SRS tables, compiled programs and every grow-only workspace are freed on destroy."""
    import torch
    curve = pkg.fields.CURVES["pallas"]
    fid = curve.scalar.id
    n = 1 << 12
    g = co.synth_bases(curve.id, n)
    sc = co.fill_scalars(fid, "uniform", n, 1)
    om = curve.scalar.encode(__import__("__graft_entry__").load_oracle()[0].FIELDS[curve.scalar.name].omega(12))
    ev = pkg.evaluation

    def cycle():
        with pkg.Context(0) as c:
            b = c.register_bases(curve.id, g, 0, True)
            c.msm(b, sc)
            c.ntt(fid, sc, 12, om)
            c.batch_invert(fid, sc)
            c.grand_product(fid, sc, sc)
            c.eval_polynomial(fid, sc, sc[0])
            c.permute_expression_pair(fid, sc, sc, n)
            gr = ev.GraphEvaluator()
            gr.add_calculation(ev.MUL, gr.column(ev.ADVICE, 0), gr.column(ev.ADVICE, 0, 1))
            cg = gr.compile(c, curve.scalar)
            d = c.upload(sc)
            o = torch.zeros_like(d)
            cg.evaluate_device([], [d.data_ptr()], [], [], None, None, None, None, 12, 1, 0, o.data_ptr())
            c.synchronize()
            cg.release()
            b.release()

    cycle()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(8):
        cycle()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0, free1)      # nothing accumulates across eight create/use/destroy cycles


def test_permute_expression_pair_batch(pkg, co, ctx):
    """Five lookups in one call (shared sort passes) against the oracle column by column; a miss in one of them is reported."""
    import torch
    fid, n, stride, B = 0, 3000, 3072, 5
    rng = np.random.default_rng(12)
    ins, tabs = np.zeros((B, stride, 4), dtype=np.uint64), np.zeros((B, stride, 4), dtype=np.uint64)
    for y in range(B):
        tsize = 100 * (y + 1)
        base = co.fill_scalars(fid, "uniform", tsize, 200 + y) if y % 2 else ctx.field_op(fid, "to_mont", np.pad(np.arange(tsize, dtype=np.uint64)[:, None], ((0, 0), (0, 3))))
        tabs[y, :n] = np.concatenate([base, np.repeat(base[:1], n - tsize, axis=0)])
        ins[y, :n] = base[rng.integers(0, tsize, size=n)]
        tabs[y, n:], ins[y, n:] = co.fill_scalars(fid, "uniform", stride - n, 300 + y), co.fill_scalars(fid, "uniform", stride - n, 400 + y)   # beyond usable rows: ignored
    di, dt = ctx.upload(ins), ctx.upload(tabs)
    oi, ot = torch.zeros_like(di), torch.zeros_like(dt)
    ctx.permute_expression_pair_batch_device(fid, di.data_ptr(), dt.data_ptr(), n, B, stride, oi.data_ptr(), ot.data_ptr(), 0)
    gi, gt = oi.cpu().numpy().view(np.uint64), ot.cpu().numpy().view(np.uint64)
    for y in range(B):
        want = co.permute_expression_pair(fid, ins[y], tabs[y], n)
        assert np.array_equal(gi[y, :n], want[0]) and np.array_equal(gt[y, :n], want[1]), y
        assert not gi[y, n:].any() and not gt[y, n:].any()
    ins[3, 5] = co.fill_scalars(fid, "uniform", 1, 999)[0]
    di = ctx.upload(ins)
    with pytest.raises(pkg.DehaloError) as e:
        ctx.permute_expression_pair_batch_device(fid, di.data_ptr(), dt.data_ptr(), n, B, stride, oi.data_ptr(), ot.data_ptr(), 0)
    assert e.value.code == -6 and "lookup 3" in str(e.value)


def test_internal_form_round_trip_and_flags(pkg, po, co, ctx):
    """The optional device-internal element form: convert there and back, coset NTT emitting / consuming it, and the three
    evaluate_h entry points with every column and value in it -- all equal to the standard-form results."""
    import torch
    ev = pkg.evaluation
    f, spec = po.BN254_FR, pkg.fields.BN254_FR
    fid = spec.id
    rng = po.Xoshiro(0x1F0)
    k, ext_k = 6, 8
    rows, rot_scale = 1 << ext_k, 1 << (ext_k - k)
    x = co.fill_scalars(fid, "uniform", 5000, 3)
    dx = ctx.upload(x)
    di = torch.zeros_like(dx)
    ctx.convert_form_device(fid, dx.data_ptr(), di.data_ptr(), 5000, True)
    back = torch.zeros_like(dx)
    ctx.convert_form_device(fid, di.data_ptr(), back.data_ptr(), 5000, False)
    ctx.synchronize()
    assert np.array_equal(back.cpu().numpy().view(np.uint64), x)
    # internal = value * 2^261 mod p (canonical): check one element against Python integers
    got = po.from_limbs64(di[7].cpu().numpy().view(np.uint64))
    assert got == spec.decode(x[7]) * pow(2, 261, f.p) % f.p
    to_int = lambda t: (ctx.convert_form_device(fid, t.data_ptr(), t.data_ptr(), t.numel() // 4, True), t)[1]
    to_std = lambda t: (ctx.convert_form_device(fid, t.data_ptr(), t.data_ptr(), t.numel() // 4, False), ctx.synchronize(), t)[2]   # the context's stream is not torch's

    # coset NTT out-internal, then coset iNTT in-internal == identity on the coefficients
    dom = pkg.EvaluationDomain(ctx, spec, 5, k)
    e = spec.encode
    coeffs = co.fill_scalars(fid, "uniform", 1 << k, 9)
    dc = ctx.upload(coeffs)
    dext = torch.zeros((1 << dom.extended_k, 4), dtype=torch.int64, device="cuda")
    ctx.coset_ntt_form_device(fid, dc.data_ptr(), k, dext.data_ptr(), dom.extended_k, e(dom.extended_omega), e(dom.g_coset), 1, ev.FORM_OUT_INTERNAL)
    want_ext = co.coeff_to_extended(fid, coeffs, k, dom.extended_k, e(dom.extended_omega), e(dom.g_coset), 2)
    chk = dext.clone(); to_std(chk); ctx.synchronize()
    assert np.array_equal(chk.cpu().numpy().view(np.uint64), want_ext)
    ctx.coset_intt_form_device(fid, dext.data_ptr(), dom.extended_k, e(dom.extended_omega_inv), e(dom.extended_ifft_divisor), e(dom.g_coset), 1, ev.FORM_IN_INTERNAL)
    ctx.synchronize()
    got = dext.cpu().numpy().view(np.uint64)
    assert np.array_equal(got[: 1 << k], coeffs) and not got[1 << k:].any()

    # evaluate_h entry points with internal columns and values
    col = lambda: [rng.below(f.p) for _ in range(rows)]
    env = {"fixed": [col() for _ in range(3)], "advice": [col() for _ in range(4)], "instance": [col()], "challenges": [rng.below(f.p) for _ in range(2)],
           "beta": rng.below(f.p), "gamma": rng.below(f.p), "theta": rng.below(f.p), "y": rng.below(f.p)}
    g = _random_graph(po, f, rng, 40, 3, 4, 1, 2, 20)
    previous = col()
    want = po.graph_evaluate(f, g, env, rows, rot_scale, previous)
    ge = ev.GraphEvaluator(constants=list(g["constants"]), rotations=list(g["rotations"]), calculations=list(g["calcs"]), num_intermediates=g["num_intermediates"])
    cg = ge.compile(ctx, spec)
    cols = {kk: [to_int(_dev(spec, c)) for c in env[kk]] for kk in ("fixed", "advice", "instance")}
    prev = to_int(_dev(spec, previous))
    cg.evaluate_device([t.data_ptr() for t in cols["fixed"]], [t.data_ptr() for t in cols["advice"]], [t.data_ptr() for t in cols["instance"]], env["challenges"],
                       env["beta"], env["gamma"], env["theta"], env["y"], ext_k, rot_scale, prev.data_ptr(), prev.data_ptr(), 0, ev.COLUMNS_INTERNAL | ev.VALUES_INTERNAL)
    assert _host(spec, to_std(prev)) == want
    cg.release()
    z, pc, sg = [col() for _ in range(2)], [col() for _ in range(5)], [col() for _ in range(5)]
    l0, l_last, l_active, values = col(), col(), col(), col()
    beta, gamma, y, delta = (rng.below(f.p) for _ in range(4))
    zeta, w = po.zeta(f), f.omega(ext_k)
    want = po.permutation_h(f, values, z, pc, sg, 3, -6, l0, l_last, l_active, beta, gamma, y, delta, zeta, w, rot_scale)
    D = lambda c: to_int(_dev(spec, c))
    dz, dcols, dsg, dl, dv = [D(c) for c in z], [D(c) for c in pc], [D(c) for c in sg], [D(l0), D(l_last), D(l_active)], D(values)
    ev.permutation_h_device(ctx, spec, [t.data_ptr() for t in dz], [t.data_ptr() for t in dcols], [t.data_ptr() for t in dsg], 3, -6, dl[0].data_ptr(), dl[1].data_ptr(),
                            dl[2].data_ptr(), beta, gamma, y, delta, zeta, w, ext_k, rot_scale, dv.data_ptr(), 0, ev.COLUMNS_INTERNAL | ev.VALUES_INTERNAL)
    assert _host(spec, to_std(dv)) == want
    prod, a, s_, tv = col(), col(), col(), col()
    want = po.lookup_h(f, values, prod, a, s_, tv, l0, l_last, l_active, beta, gamma, y, rot_scale)
    dv = D(values)
    dp, da, ds, dtv = D(prod), D(a), D(s_), D(tv)
    ev.lookup_h_device(ctx, spec, dp.data_ptr(), da.data_ptr(), ds.data_ptr(), dtv.data_ptr(), dl[0].data_ptr(), dl[1].data_ptr(), dl[2].data_ptr(), beta, gamma, y, ext_k,
                       rot_scale, dv.data_ptr(), 0, ev.COLUMNS_INTERNAL | ev.VALUES_INTERNAL)
    assert _host(spec, to_std(dv)) == want
    # mixed: standard columns, internal values
    dv = D(values)
    ev.lookup_h_device(ctx, spec, (tp := _dev(spec, prod)).data_ptr(), (ta := _dev(spec, a)).data_ptr(), (ts := _dev(spec, s_)).data_ptr(), dtv.data_ptr(),
                       (t0 := _dev(spec, l0)).data_ptr(), (t1 := _dev(spec, l_last)).data_ptr(), (t2 := _dev(spec, l_active)).data_ptr(), beta, gamma, y, ext_k, rot_scale,
                       dv.data_ptr(), 0, ev.VALUES_INTERNAL)
    assert _host(spec, to_std(dv)) == want


def test_evalh_golden(pkg, po, ctx):
    """The committed vectors of tests/golden/evalh.json through the device kernels."""
    import torch
    ev = pkg.evaluation
    ints = lambda l: [int(x, 16) for x in l]
    for v in golden("evalh"):
        spec = pkg.fields.FIELDS[v["field"]]
        D = lambda c: _dev(spec, ints(c))
        if v["op"] == "graph":
            g = v["graph"]
            ge = ev.GraphEvaluator(constants=ints(g["constants"]), rotations=list(g["rotations"]), num_intermediates=g["num_intermediates"],
                                   calculations=[(op, tuple(a), tuple(b), tuple(tuple(q) for q in parts), t) for op, a, b, parts, t in g["calcs"]])
            cg = ge.compile(ctx, spec)
            cols = {k: [D(c) for c in v["env"][k]] for k in ("fixed", "advice", "instance")}
            prev = D(v["previous"])
            cg.evaluate_device([t.data_ptr() for t in cols["fixed"]], [t.data_ptr() for t in cols["advice"]], [t.data_ptr() for t in cols["instance"]],
                               ints(v["env"]["challenges"]), *[int(v["env"][k], 16) for k in ("beta", "gamma", "theta", "y")], v["ext_k"], v["rot_scale"],
                               prev.data_ptr(), prev.data_ptr())
            ctx.synchronize()
            assert _host(spec, prev) == ints(v["result"])
            cg.release()
        elif v["op"] == "permutation":
            z, cols, sigma = ([D(c) for c in v[k]] for k in ("z", "columns", "sigma"))
            l0, l_last, l_active, values = (D(v[k]) for k in ("l0", "l_last", "l_active", "values"))
            sc = {k: int(v[k], 16) for k in ("beta", "gamma", "y", "delta", "zeta", "extended_omega")}
            ev.permutation_h_device(ctx, spec, [t.data_ptr() for t in z], [t.data_ptr() for t in cols], [t.data_ptr() for t in sigma], v["chunk_len"], v["last_rotation"],
                                    l0.data_ptr(), l_last.data_ptr(), l_active.data_ptr(), sc["beta"], sc["gamma"], sc["y"], sc["delta"], sc["zeta"], sc["extended_omega"],
                                    v["ext_k"], v["rot_scale"], values.data_ptr())
            ctx.synchronize()
            assert _host(spec, values) == ints(v["result"])
        elif v["op"] == "lookup":
            t = {k: D(v[k]) for k in ("product", "permuted_input", "permuted_table", "table_value", "l0", "l_last", "l_active", "values")}
            ev.lookup_h_device(ctx, spec, t["product"].data_ptr(), t["permuted_input"].data_ptr(), t["permuted_table"].data_ptr(), t["table_value"].data_ptr(),
                               t["l0"].data_ptr(), t["l_last"].data_ptr(), t["l_active"].data_ptr(), int(v["beta"], 16), int(v["gamma"], 16), int(v["y"], 16), v["ext_k"],
                               v["rot_scale"], t["values"].data_ptr())
            ctx.synchronize()
            assert _host(spec, t["values"]) == ints(v["result"])
        else:
            pi, pt = pkg.lookup.permute_expression_pair(ctx, spec, spec.encode_many(ints(v["input"])), spec.encode_many(ints(v["table"])), v["usable"])
            assert spec.decode_many(pi) == ints(v["permuted_input"]) and spec.decode_many(pt) == ints(v["permuted_table"])
    with pytest.raises(pkg.lookup.ConstraintSystemFailure):
        spec = pkg.fields.BN254_FR
        pkg.lookup.permute_expression_pair(ctx, spec, spec.encode_many([1, 2, 77]), spec.encode_many([1, 2, 3]), 3)


def test_msm_split_by_point_range(pkg, co, ctx):
    """One MSM split over four 'ranks' by point range (each registers its slice of the bases), partial results summed on
    the device: equals the undivided MSM.  (The all-gather between real ranks is covered by the gloo test.)"""
    import torch
    from dehalo2_amd import sharding
    curve = pkg.fields.CURVES["bn254"]
    n, world = 5003, 4
    g = co.synth_bases(curve.id, n)
    sc = co.fill_scalars(curve.scalar.id, "witness", n, 77)
    parts = torch.zeros((world + 1, 12), dtype=torch.int64, device="cuda")      # + one identity entry
    handles = []
    for r in range(world):
        lo, hi = sharding.point_range_for_rank(n, r, world)
        h = ctx.register_bases(curve.id, g[lo:hi], 0, True); handles.append(h)
        d = ctx.upload(sc[lo:hi])
        ctx.msm_device(h, d.data_ptr(), hi - lo, 1, parts[r].data_ptr(), 0)
        ctx.synchronize()
    assert [sharding.point_range_for_rank(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    out = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
    ctx.point_sum_device(curve.id, parts.data_ptr(), world + 1, out.data_ptr(), 0)
    ctx.synchronize()
    got = ctx.to_affine(curve.id, out.cpu().numpy().view(np.uint64))[0]
    want = co.to_affine(curve.id, co.best_multiexp(curve.id, sc, g, 4))
    assert np.array_equal(got, want)
    # through the sharding helper at world size 1, and the empty sum
    one = sharding.combine_partial_msm(ctx, curve.id, out, 0, 1)
    assert np.array_equal(ctx.to_affine(curve.id, one.cpu().numpy().view(np.uint64))[0], want)
    ctx.point_sum_device(curve.id, parts.data_ptr(), 0, out.data_ptr(), 0)
    ctx.synchronize()
    assert not out.cpu().numpy().view(np.uint64)[0, 8:].any()                    # z = 0: identity
    for h in handles:
        h.release()


@pytest.mark.gpu
@pytest.mark.parametrize("cname", ["pallas", "vesta"])
@pytest.mark.parametrize("k", [4, 10, 13])
def test_params_ipa_commit_vs_oracle(pkg, co, ctx, cname, k):
    """ParamsIPA::{commit, commit_lagrange}(poly, r) = <poly, g> + r * w (n + 1 terms; reference: the IPA variant kept at
    benches/delay_enc.rs:42,51-52) against the C oracle's best_multiexp over g || w."""
    curve = pkg.fields.CURVES[cname]
    n = 1 << k
    pts = co.synth_bases(curve.id, 2 * n + 1)
    g, gl, w = pts[:n], pts[n:2 * n], pts[2 * n:]
    params = pkg.ParamsIPA(ctx, curve, k, g, gl, w)
    for dist, seed in (("uniform", 1), ("witness", 2)):
        poly = co.fill_scalars(curve.scalar.id, dist, n, seed)
        r = co.fill_scalars(curve.scalar.id, "uniform", 1, 100 + seed)
        for fn, basis in ((params.commit, g), (params.commit_lagrange, gl)):
            got = ctx.to_affine(curve.id, fn(poly, r))[0]
            want = co.to_affine(curve.id, co.best_multiexp(curve.id, np.concatenate([poly, r]), np.concatenate([basis, w]), 3))
            assert np.array_equal(got, want)
    zero = np.zeros((1, 4), dtype=np.uint64)
    poly = co.fill_scalars(curve.scalar.id, "uniform", n, 9)
    assert np.array_equal(ctx.to_affine(curve.id, params.commit(poly, zero))[0], co.to_affine(curve.id, co.best_multiexp(curve.id, poly, g, 2)))
    with pytest.raises(ValueError):
        params.commit(poly[:-1], zero)
    params.release()


@pytest.mark.gpu
def test_permute_expression_pair_leading_bit_ties_fall_back(pkg, po, co, ctx):
    """The lookup sort's fast path orders on the leading 48 bits of the top limb and verifies the full order: values that agree
    on those bits and differ below must take the full sort and still match upstream's order."""
    f, of = pkg.fields.BN254_FR, po.BN254_FR
    n = 3000
    rng = po.Xoshiro(77)
    high = 0x1234567890ABCDEF << 190                       # every value shares its leading bits
    table_vals = [(high + ((i * 7919) % 4096) * (1 << (64 * (i % 3)))) % f.p for i in range(512)]
    table = [table_vals[i % 512] for i in range(n)]
    inputs = [table_vals[rng.below(512)] for _ in range(n)]
    ti, tt = f.encode_many(inputs), f.encode_many(table)
    got = ctx.permute_expression_pair(f.id, ti, tt, n)
    want = co.permute_expression_pair(f.id, ti, tt, n)
    assert want is not None and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.gpu
@pytest.mark.parametrize("fname", ["bn254_fr", "pasta_fq"])
def test_product_terms_vs_python_integers(pkg, co, ctx, fname):
    """dehalo_product_terms_device: the grand products' per-row numerators / denominators [UPSTREAM plonk/permutation/prover.rs commit,
    plonk/lookup/prover.rs commit_product] against a direct evaluation on Python integers: 7 columns in chunks of 3 (three sets, the last
    with one column), two lookups, a row count that is not a multiple of the block size."""
    import torch
    from dehalo2_amd.keygen import delta_of

    f = pkg.fields.FIELDS[fname]
    n, stride, ncols, chunk, nl = 333, 340, 7, 3, 2
    dev = lambda a: ctx.upload(a)
    cols = [co.fill_scalars(f.id, "uniform", n, 40 + j) for j in range(ncols)]
    sig = [co.fill_scalars(f.id, "uniform", n, 60 + j) for j in range(ncols)]
    lk = [[co.fill_scalars(f.id, "witness" if t % 2 else "uniform", n, 80 + 4 * l + t) for t in range(4)] for l in range(nl)]
    omega = pow(f.root_of_unity, 1 << (f.two_adicity - 9), f.p)
    om = f.encode_many([pow(omega, i, f.p) for i in range(n)])
    beta, gamma, delta = 0x1234567 * 0x9E3779B97F4A7C15 % f.p, 0xABCDEF123 * 0xC2B2AE3D27D4EB4F % f.p, delta_of(f)
    sets = (ncols + chunk - 1) // chunk
    d_cols, d_sig, d_om = [dev(c) for c in cols], [dev(c) for c in sig], dev(om)
    d_lk = [[dev(c) for c in four] for four in lk]
    num = torch.zeros((sets + nl, stride, 4), dtype=torch.int64, device="cuda")
    den = torch.zeros_like(num)
    torch.cuda.synchronize()
    ctx.product_terms_device(f.id, [c.data_ptr() for c in d_cols], [c.data_ptr() for c in d_sig], chunk, d_om.data_ptr(), f.encode(beta), f.encode(gamma), f.encode(delta),
                             f.encode_many([beta * pow(delta, chunk * s, f.p) % f.p for s in range(sets)]), [tuple(c.data_ptr() for c in four) for four in d_lk], n,
                             num.data_ptr(), den.data_ptr(), stride)
    ctx.synchronize()
    got_n, got_d = num.cpu().numpy().view(np.uint64), den.cpu().numpy().view(np.uint64)
    C, S, LK = [f.decode_many(c) for c in cols], [f.decode_many(c) for c in sig], [[f.decode_many(c) for c in four] for four in lk]
    for s in range(sets):
        wn, wd = [], []
        for i in range(n):
            a = b = 1
            for j in range(s * chunk, min(ncols, (s + 1) * chunk)):
                b = b * (C[j][i] + beta * S[j][i] + gamma) % f.p
                a = a * (C[j][i] + pow(delta, j, f.p) * beta % f.p * pow(omega, i, f.p) + gamma) % f.p
            wn.append(a); wd.append(b)
        assert f.decode_many(got_n[s, :n]) == wn and f.decode_many(got_d[s, :n]) == wd, s
    for l in range(nl):
        A, Sx, a_, s_ = LK[l]
        assert f.decode_many(got_n[sets + l, :n]) == [(A[i] + beta) * (Sx[i] + gamma) % f.p for i in range(n)]
        assert f.decode_many(got_d[sets + l, :n]) == [(a_[i] + beta) * (s_[i] + gamma) % f.p for i in range(n)]
    assert not got_n[:, n:].any() and not got_d[:, n:].any()      # the padding between columns is untouched


@pytest.mark.gpu
def test_staged_upload_and_download_of_caller_memory(pkg, ctx):
    """dehalo_upload / dehalo_download: every byte between caller memory and the device goes through the context's page-locked staging chunks (4 MiB, two of
    them) or moves by DMA from pages the library pins for the call -- sizes around the thresholds (64 KiB direct, 4 MiB pin), not multiples of a chunk, an
    unaligned source, a source the caller page-locked itself."""
    import torch
    rng = np.random.default_rng(7)
    for words in (1, 8191, 8192, 8193, (4 << 20) // 8 - 1, (4 << 20) // 8 + 5, 3 * (4 << 20) // 8 + 12345, (9 << 20) // 8):
        a = rng.integers(0, 1 << 63, size=words + 1, dtype=np.int64).view(np.uint64)
        for src in (a[:words], a[1:]):                       # 8-byte aligned, and 8 bytes into the allocation
            src = np.ascontiguousarray(src) if src.base is None else src
            d = ctx.upload(src)
            assert np.array_equal(ctx.download_tensor(d), src)
            assert np.array_equal(d.cpu().numpy().view(np.uint64), src)
    pinned = torch.empty((3 << 20,), dtype=torch.int64).pin_memory()
    pinned.copy_(torch.from_numpy(rng.integers(0, 1 << 62, size=3 << 20, dtype=np.int64)))
    host = pinned.numpy().view(np.uint64)
    d = ctx.upload(host)
    assert np.array_equal(ctx.download_tensor(d), host)
    # the same staging chunks serve two threads at once (one context, serialised inside)
    import threading
    errs = []
    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            for _ in range(4):
                x = r.integers(0, 1 << 63, size=(1 << 20) + seed, dtype=np.int64).view(np.uint64)
                if not np.array_equal(ctx.download_tensor(ctx.upload(x)), x):
                    errs.append("mismatch in thread %d" % seed)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(s,)) for s in (1, 2, 3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
