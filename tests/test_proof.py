"""create_proof as a real data flow (reference call sites benches/delay_enc.rs:84-165):

CPU suite  -- the oracle's own chain: setup -> keygen -> create_proof -> verify_proof accepts (the reference's one
              end-to-end check, `assert!(accept)`), rejects a tampered proof; pairing bilinearity; transcript and
              point-compression known answers; the constraint-system shapes of the two circuits.
GPU suite  -- the device prover against that oracle: identical verifying key, byte-identical proof, accepted by the
              verifier, at k = 6 / 9 and at the full sizes k = 11 (pose_enc shape) and k = 17 (delay_enc shape);
              formats round-trip; the new element-wise entry points against the C oracle.
"""
import hashlib
import io
import threading

import numpy as np
import pytest

import fine_grained_prover as fgp      # the Python driver over the fine-grained entry points: test infrastructure (was dehalo2_amd/prover.py's Prover)

S_TOXIC = 0x1234567890ABCDEF1234567890ABCDEF


@pytest.fixture(scope="session")
def chain(pkg, po, co):
    """(k, range_lookups) -> circuit, SRS, oracle key, proof material; cached for the session."""
    import plonk_oracle as PO
    import pairing as pr
    from dehalo2_amd import circuits

    cache = {}

    def get(k, rl, threads=8):
        if (k, rl) not in cache:
            curve = po.BN254
            circ = circuits.synthesize(curve.scalar.p, k, rl, seed=3)
            import shapes
            desc = shapes.maingate_description(rl)                 # the checker's own statement of the shape ...
            assert desc == circ.cs.description()                   # ... which the product's must equal
            srs = PO.setup_srs(curve, k, S_TOXIC, threads)
            key = PO.keygen(curve, srs, desc, k, circ.fixed, circ.assembly.mapping, threads)
            rep = PO.transcript_repr(curve, key, circ.selectors)
            F = PO.Fld(curve.scalar)
            adv = np.stack([co.field_op(F.id, "to_mont", circ.advice[i]) for i in range(5)])
            cache[(k, rl)] = dict(circ=circ, desc=desc, srs=srs, key=key, rep=rep, adv=adv, s_g2=pr.g2_mul(S_TOXIC, pr.G2))
        return cache[(k, rl)]

    return get


def oracle_proof(po, c, seed=7, threads=8):
    import plonk_oracle as PO
    from dehalo2_amd import prover

    return PO.create_proof(po.BN254, c["srs"], c["key"], c["adv"], [[]], PO.ScalarStream(seed), c["rep"], threads)


def oracle_verify(po, c, proof, k):
    import pairing as pr
    import verifier as V

    return V.verify_proof(po.BN254, c["desc"], k, c["key"]["fixed_commitments"], c["key"]["perm_commitments"], c["rep"], (1, 2), pr.G2, c["s_g2"], [[]], proof)


def arith_row(c) -> int:
    """first row whose s_mul_ab selector is set (and whose a is non-zero)"""
    from dehalo2_amd import plonk

    fx, adv = c["circ"].fixed, c["circ"].advice
    return next(r for r in range(fx.shape[1]) if fx[plonk.MG_MUL_AB, r].any() and adv[0, r].any())


# ---------------------------------------------------------------- CPU suite
def test_circuit_shapes(pkg):
    """delay_enc / mod_pow = 5 advice, 15 fixed, 5 lookups, degree 5, 2 permutation sets, 5 blinding factors;
    pose_enc = MainGate only, degree 3, 6 permutation sets."""
    from dehalo2_amd import plonk

    d, q = plonk.maingate_cs(True), plonk.maingate_cs(False)
    assert (d.num_advice, d.num_fixed, d.num_instance, len(d.lookups), d.degree(), d.num_permutation_sets(), d.blinding_factors()) == (5, 15, 1, 5, 5, 2, 5)
    assert (q.num_advice, q.num_fixed, len(q.lookups), q.degree(), q.num_permutation_sets(), q.blinding_factors()) == (5, 9, 0, 3, 6, 5)
    assert d.advice_queries == [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (4, 1)] and len(d.permutation_columns) == 6
    assert len(plonk.range_table()) == 1 + 256 + 16 + 2 + 64


def test_pairing_bilinear(po):
    import pairing as pr

    assert pr.g2_on_curve(pr.G2) and pr.g2_mul(pr.R, pr.G2) is None
    e1 = pr.pairing(pr.G2, pr.G1)
    a, b = 0x1F3A5, 0x7C0FFEE
    assert pr.pairing(pr.g2_mul(b, pr.G2), po.ec_mul(po.BN254, a, (1, 2))) == pr.f12_pow(e1, a * b)
    assert e1 != pr.F12_ONE and pr.f12_pow(e1, pr.R) == pr.F12_ONE
    assert pr.g2_from_raw(pr.g2_to_raw(pr.g2_mul(5, pr.G2))) == pr.g2_mul(5, pr.G2)


def test_transcript_known_answers(pkg, po):
    """Blake2b-512, personalisation "Halo2-Transcript", prefix bytes 0 / 1 / 2, 64-byte digest reduced mod r; the package's
    transcript and the oracle's agree with a direct hashlib restatement."""
    import plonk_oracle as PO
    from dehalo2_amd import transcript

    curve = pkg.fields.BN254
    P = po.ec_mul(po.BN254, 12345, (1, 2))
    t, o = transcript.Blake2bWrite(curve), PO.Transcript(po.BN254)
    for tr in (t, o):
        tr.common_scalar(7)
        tr.write_point(P)
        tr.write_scalar(curve.scalar.p - 1)
    c1 = t.squeeze_challenge_scalar()
    assert c1 == o.challenge()
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
    h.update(b"\x02" + (7).to_bytes(32, "little") + b"\x01" + P[0].to_bytes(32, "little") + P[1].to_bytes(32, "little") + b"\x02" +
             (curve.scalar.p - 1).to_bytes(32, "little") + b"\x00")
    assert c1 == int.from_bytes(h.digest(), "little") % curve.scalar.p
    assert t.finalize() == bytes(o.proof) and len(t.finalize()) == 64
    # a second challenge continues the same state (the prefix byte stays absorbed)
    assert t.squeeze_challenge_scalar() == o.challenge() != c1
    r = transcript.Blake2bRead(curve, t.finalize())
    r.common_scalar(7)
    assert r.read_point() == P and r.read_scalar() == curve.scalar.p - 1 and r.squeeze_challenge_scalar() == c1
    with pytest.raises(ValueError):
        t.write_point(None)


@pytest.mark.parametrize("cname", ["bn254", "pallas", "vesta"])
def test_point_compression_roundtrip(pkg, po, cname):
    from dehalo2_amd import transcript

    curve, oc = pkg.fields.CURVES[cname], po.CURVES[cname]
    for kmul in (1, 2, 3, 0xDEADBEEF):
        P = po.ec_mul(oc, kmul, (oc.gx, oc.gy))
        for Q in (P, po.ec_neg(oc, P)):
            b = transcript.compress(curve, Q)
            assert len(b) == 32 and transcript.decompress(curve, b) == Q
    assert transcript.compress(curve, None) == bytes(32) and transcript.decompress(curve, bytes(32)) is None
    with pytest.raises(ValueError):
        transcript.decompress(curve, b"\xff" * 32)


@pytest.mark.parametrize("k,rl", [(6, False), (9, True)])
def test_oracle_proof_is_accepted(po, chain, k, rl):
    c = chain(k, rl)
    proof, trace = oracle_proof(po, c)
    assert len(trace["commitments"]) == (31 if rl else 17)            # ~31 / ~17 commitments per proof
    assert oracle_verify(po, c, proof, k)
    again, _ = oracle_proof(po, c)
    assert again == proof                                             # seeded SRS + seeded blinding: reproducible bytes
    other, _ = oracle_proof(po, c, seed=8)
    assert other != proof and oracle_verify(po, c, other, k)
    for pos in (5, 32 * 20 + 3, len(proof) - 1):                      # a commitment, a later item, the last quotient
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert not oracle_verify(po, c, bytes(bad), k)
    assert not oracle_verify(po, c, proof[:-32], k) and not oracle_verify(po, c, proof + bytes(32), k)


def test_x_last_queries_run_from_the_second_last_set_down(po, chain):
    """[UPSTREAM plonk/permutation/prover.rs Evaluated::open, verifier.rs Evaluated::queries: `sets.iter().rev().skip(1)`] -- the
    openings at omega^last x are queued from set S - 2 down to set 0, so z_{S-2} takes v^0 in that point's batch; the evaluations
    themselves are WRITTEN in forward order.  pose_enc has six sets (the device prover is pinned to the same order by the
    byte comparison with this oracle at K = 11)."""
    c = chain(6, False)
    proof, trace = oracle_proof(po, c)
    last = trace["perm_last_evals"]
    assert len(last) == 5 and len(set(last)) == 5
    assert trace["x_last_group"] == list(reversed(last))
    assert oracle_verify(po, c, proof, 6)


def test_oracle_rejects_unsatisfied_witness(po, co, chain):
    """One wrong advice cell: t(X) no longer divides the numerator, h(x) (x^n - 1) differs from the folded expressions, rejected."""
    import plonk_oracle as PO
    from dehalo2_amd import prover

    c = chain(6, False)
    adv = c["adv"].copy()
    adv[1, arith_row(c)] = PO.Fld(po.BN254.scalar).m(123456789)      # b of a row whose a * b term is switched on
    proof, trace = PO.create_proof(po.BN254, c["srs"], c["key"], adv, [[]], PO.ScalarStream(7), c["rep"], 4)
    assert not oracle_verify(po, c, proof, 6)


def test_kate_division_and_lincomb_oracle(po, co):
    f = po.BN254_FR
    fid = po.FIELD_IDS[f.name]
    import plonk_oracle as PO
    F = PO.Fld(f)
    rng = po.Xoshiro(99)
    a = po.scalars_uniform(f, 37, rng)
    z = 0xABCDEF
    q = F.un_many(co.kate_division(fid, F.many(a), F.m(z)))
    az = po.eval_polynomial(f, a, z)
    # (X - z) q(X) + a(z) == a(X)
    back = [(-z * q[0] + az) % f.p] + [(q[i - 1] - z * q[i]) % f.p for i in range(1, 36)] + [q[35]]
    assert back == a
    cols = [po.scalars_uniform(f, 9, rng) for _ in range(3)]
    cf = [5, f.p - 2, 77]
    got = F.un_many(co.lincomb(fid, [F.many(c) for c in cols], F.many(cf), F.m(11)))
    want = [sum(c * col[i] for c, col in zip(cf, cols)) % f.p for i in range(9)]
    want[0] = (want[0] - 11) % f.p
    assert got == want


# ---------------------------------------------------------------- GPU suite
@pytest.fixture(scope="session")
def device_chain(pkg, ctx, chain):
    import pairing as pr
    from dehalo2_amd import keygen, prover

    cache = {}

    def get(k, rl):
        if (k, rl) not in cache:
            c = chain(k, rl, threads=16)
            params = keygen.ParamsKZG(ctx, pkg.fields.BN254, k, c["srs"]["g"], c["srs"]["g_lagrange"], pr.g2_to_raw(pr.G2), pr.g2_to_raw(c["s_g2"]))
            pk = keygen.keygen(ctx, params, c["circ"].cs, c["circ"].fixed, c["circ"].assembly, c["circ"].selectors)
            cache[(k, rl)] = dict(params=params, pk=pk, prover=fgp.Prover(params, pk))
        return cache[(k, rl)]

    yield get
    for v in cache.values():
        v["params"].release()


@pytest.mark.gpu
@pytest.mark.parametrize("k,rl", [(6, False), (9, True), (11, False), (17, True)])
def test_device_proof_matches_oracle_and_verifies(pkg, po, ctx, chain, device_chain, k, rl):
    """configs[0] shape at k = 11 and configs[3] shape at k = 17 included: the verifying key, every commitment, evaluation
    and opening quotient -- the whole proof, byte for byte -- equal the CPU restatement's, and the verifier accepts it."""
    import plonk_oracle as PO
    from dehalo2_amd import keygen, prover, transcript

    c, d = chain(k, rl, threads=16), device_chain(k, rl)
    pk = d["pk"]
    assert keygen.decode_points(pkg.fields.BN254, pk.vk.fixed_commitments) == c["key"]["fixed_commitments"]
    assert keygen.decode_points(pkg.fields.BN254, pk.vk.permutation_commitments) == c["key"]["perm_commitments"]
    buf = io.BytesIO()
    pk.vk.write(buf)
    assert buf.getvalue() == PO.vk_bytes(po.BN254, c["key"], c["circ"].selectors) and pk.vk.transcript_repr == c["rep"]
    tr = transcript.Blake2bWrite(pkg.fields.BN254)
    d["prover"].create_proof(c["adv"], [[]], prover.SeededRng(7), tr)
    proof = tr.finalize()
    want, trace = oracle_proof(po, c, threads=16)
    assert len(proof) == len(want)
    diff = [i // 32 for i in range(0, len(want), 32) if proof[i:i + 32] != want[i:i + 32]]
    assert not diff, "proof items differ from the oracle's: %r" % diff[:8]
    assert oracle_verify(po, c, proof, k)
    # the same prover object again (buffers reused), other blinding: still a valid, different proof
    tr2 = transcript.Blake2bWrite(pkg.fields.BN254)
    d["prover"].create_proof(c["adv"], [[]], prover.SeededRng(8), tr2)
    assert tr2.finalize() != proof
    if k <= 11:
        assert oracle_verify(po, c, tr2.finalize(), k)


@pytest.mark.gpu
@pytest.mark.parametrize("k,rl", [(9, True), (11, False)])
def test_side_context_prover_makes_the_same_proof(pkg, po, ctx, chain, device_chain, k, rl):
    """Prover(side_ctx=...): the NTTs of each phase's columns and the random polynomial's commitment run on a second context
    beside the commitment phases; the proof bytes do not change."""
    from dehalo2_amd import prover, transcript

    c, d = chain(k, rl), device_chain(k, rl)
    want, _ = oracle_proof(po, c)
    side = pkg.Context(0)
    P2 = fgp.Prover(d["params"], d["pk"], ctx, side)
    for _ in range(3):
        tr = transcript.Blake2bWrite(pkg.fields.BN254)
        P2.create_proof(c["adv"], [[]], prover.SeededRng(7), tr)
        assert tr.finalize() == want
    side.close()


@pytest.mark.gpu
def test_device_proof_unsatisfied_witness_is_rejected(pkg, po, ctx, chain, device_chain):
    from dehalo2_amd import prover, transcript

    c, d = chain(6, False), device_chain(6, False)
    adv = c["adv"].copy()
    adv[1, arith_row(c)] = pkg.fields.BN254_FR.encode(987654321)
    tr = transcript.Blake2bWrite(pkg.fields.BN254)
    d["prover"].create_proof(adv, [[]], prover.SeededRng(7), tr)
    assert not oracle_verify(po, c, tr.finalize(), 6)


@pytest.mark.gpu
def test_formats_roundtrip(pkg, po, ctx, chain, device_chain):
    """ParamsKZG / VerifyingKey / ProvingKey RawBytes: write -> read -> write is the identity, sizes follow upstream's layout
    (pk = 3 E + (2 + E)(F + P) columns), and a proof made from the re-read objects is the same proof."""
    from dehalo2_amd import keygen, prover, transcript

    k, rl = 9, True
    c, d = chain(k, rl), device_chain(k, rl)
    n, curve = 1 << k, pkg.fields.BN254
    b = io.BytesIO()
    d["params"].write(b)
    raw = b.getvalue()
    assert len(raw) == 4 + 2 * n * 64 + 256 and raw[:4] == (k).to_bytes(4, "little")
    p2 = keygen.ParamsKZG.read(ctx, curve, io.BytesIO(raw))
    b2 = io.BytesIO()
    p2.write(b2)
    assert b2.getvalue() == raw
    b = io.BytesIO()
    d["pk"].write(b)
    pkraw = b.getvalue()
    cs = c["circ"].cs
    F, P, E = cs.num_fixed, len(cs.permutation_columns), 4
    vk_len = 8 + 64 * (F + P) + 2 * ((n + 7) // 8)
    assert len(pkraw) == vk_len + 3 * (4 + 32 * E * n) + 6 * 4 + (F + P) * (2 * (4 + 32 * n) + 4 + 32 * E * n)
    pk2 = keygen.ProvingKey.read(ctx, curve, cs, io.BytesIO(pkraw), num_selectors=2)
    b2 = io.BytesIO()
    pk2.write(b2)
    assert b2.getvalue() == pkraw and pk2.vk.transcript_repr == d["pk"].vk.transcript_repr
    vk2 = keygen.VerifyingKey.read(curve, cs, io.BytesIO(pkraw[:vk_len]), num_selectors=2)
    assert vk2.transcript_repr == d["pk"].vk.transcript_repr
    with pytest.raises(ValueError):
        keygen.ProvingKey.read(ctx, curve, cs, io.BytesIO(pkraw[:-5]), num_selectors=2)
    tr = transcript.Blake2bWrite(curve)
    fgp.Prover(p2, pk2).create_proof(c["adv"], [[]], prover.SeededRng(7), tr)
    want, _ = oracle_proof(po, c)
    assert tr.finalize() == want
    p2.release()


@pytest.mark.gpu
@pytest.mark.parametrize("fname", ["bn254_fr", "pasta_fp"])
@pytest.mark.parametrize("n", [1, 2, 7, 2048, 2049, 5000, (1 << 17), (1 << 20) + 3])
def test_kate_division_vs_oracle(pkg, co, ctx, fname, n):
    f = pkg.fields.FIELDS[fname]
    a = co.fill_scalars(f.id, "uniform", n, 31 + n)
    z = f.encode(0x123456789ABCDEF0FEDCBA9876543210 + n)
    got = ctx.kate_division(f.id, a, z)
    assert np.array_equal(got, co.kate_division(f.id, a, z))


@pytest.mark.gpu
@pytest.mark.parametrize("fname", ["bn254_fr", "pasta_fq"])
@pytest.mark.parametrize("n", [1, 9, 2048, 2049, 70000, 1 << 17])
def test_eval_polynomial_multi_vs_oracle(pkg, co, ctx, n, fname):
    import torch

    f = pkg.fields.FIELDS[fname]
    batch = 5
    cols = np.stack([co.fill_scalars(f.id, "uniform", n, 300 + i) for i in range(batch)])
    pts = co.fill_scalars(f.id, "uniform", 4, 17)
    d = ctx.upload(cols)
    for npts in (1, 3, 4):
        out = torch.zeros((npts, batch, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.eval_polynomial_multi_device(f.id, [d[b].data_ptr() for b in range(batch)], n, pts[:npts], out.data_ptr())
        ctx.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        for i in range(npts):
            for b in range(batch):
                assert np.array_equal(got[i, b], co.eval_polynomial(f.id, cols[b], pts[i], 2)), (n, npts, i, b)
        # only some (polynomial, point) pairs wanted (a proof's case): those are the same values, the others come back as zero
        wanted = [(0b1011, 0b0000, 0b0100, 0b1111, 0b0001)[b] & ((1 << npts) - 1) for b in range(batch)]
        out2 = torch.full((npts, batch, 4), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.eval_polynomial_multi_device(f.id, [d[b].data_ptr() for b in range(batch)], n, pts[:npts], out2.data_ptr(), wanted=wanted)
        ctx.synchronize()
        got2 = out2.cpu().numpy().view(np.uint64)
        for i in range(npts):
            for b in range(batch):
                assert np.array_equal(got2[i, b], got[i, b] if (wanted[b] >> i) & 1 else np.zeros(4, np.uint64)), (n, npts, i, b)


@pytest.mark.gpu
@pytest.mark.parametrize("fname", ["bn254_fr", "pasta_fp"])
@pytest.mark.parametrize("n", [2, 2048, 2049, 1 << 17, (1 << 18) + 5])
def test_kate_division_batch_vs_oracle(pkg, co, ctx, n, fname):
    import torch

    f = pkg.fields.FIELDS[fname]
    cnt = 4
    a = np.stack([co.fill_scalars(f.id, "uniform", n, 900 + i) for i in range(cnt)])
    pts = co.fill_scalars(f.id, "uniform", cnt, 23)
    da = ctx.upload(a)
    dq = torch.zeros((cnt, n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.kate_division_batch_device(f.id, [da[i].data_ptr() for i in range(cnt)], n, pts, [dq[i].data_ptr() for i in range(cnt)])
    ctx.synchronize()
    got = dq.cpu().numpy().view(np.uint64)
    for i in range(cnt):
        assert np.array_equal(got[i, :n - 1], co.kate_division(f.id, a[i], pts[i])), (n, i)
        assert not got[i, n - 1].any()


@pytest.mark.gpu
@pytest.mark.parametrize("fname", ["bn254_fr", "pasta_fp"])
def test_lincomb_and_scale_vs_oracle(pkg, co, ctx, fname):
    import torch

    f = pkg.fields.FIELDS[fname]
    n = 5000
    for count in (1, 3, 40, 41, 97):
        cols = [co.fill_scalars(f.id, "uniform", n, 100 + i) for i in range(count)]
        coefs = co.fill_scalars(f.id, "uniform", count, 5)
        sub = f.encode(424242)
        d = [ctx.upload(c) for c in cols]
        out = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.lincomb_device(f.id, [t.data_ptr() for t in d], coefs, n, out.data_ptr(), sub)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), co.lincomb(f.id, cols, coefs, sub))
    a = co.fill_scalars(f.id, "uniform", 4096 + 5, 9)
    for period in (1, 2, 4, 8):
        pat = co.fill_scalars(f.id, "uniform", period, 70 + period)
        d = ctx.upload(a)
        fac = ctx.upload(co.fill_scalars(f.id, "uniform", 1, 3))
        torch.cuda.synchronize()
        ctx.scale_device(f.id, d.data_ptr(), a.shape[0], pat, 0)
        ctx.synchronize()
        want = co.scale_periodic(f.id, a, pat)
        assert np.array_equal(d.cpu().numpy().view(np.uint64), want)
        ctx.scale_device(f.id, d.data_ptr(), a.shape[0], None, fac.data_ptr())
        ctx.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint64), co.scale_periodic(f.id, want, fac.cpu().numpy().view(np.uint64)))


@pytest.mark.gpu
def test_two_threads_share_one_context(pkg, co, ctx):
    """host-buffer entry points hold the context for the whole call (staging, kernels, download): two threads on one
    context get their own results."""
    curve, f = pkg.fields.PALLAS, pkg.fields.PASTA_FP
    n = 1 << 12
    bases = co.synth_bases(curve.id, n)
    reg = ctx.register_bases(curve.id, bases, 0, True)
    jobs = []
    for t in range(2):
        s = co.fill_scalars(curve.scalar.id, "uniform", n, 500 + t)
        a = co.fill_scalars(f.id, "uniform", n, 600 + t)
        jobs.append((s, a))
    omega = f.encode(pow(f.root_of_unity, 1 << (f.two_adicity - 12), f.p))
    want = [(co.to_affine(curve.id, co.best_multiexp(curve.id, s, bases, 2)), co.best_fft(f.id, a, omega, 12, 2)) for s, a in jobs]
    errors = []

    def work(idx):
        try:
            s, a = jobs[idx]
            for _ in range(20):
                got = ctx.to_affine(curve.id, ctx.msm(reg, s))[0]
                assert np.array_equal(got, want[idx][0]), "msm result of another thread"
                assert np.array_equal(ctx.ntt(f.id, a, 12, omega), want[idx][1]), "ntt result of another thread"
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    reg.release()
    assert not errors, errors


# ---------------------------------------------------------------- sizes the reference publishes
def _mac_ls_h(nbytes: int) -> str:
    """`ls -lh` on the reference authors' machine (MacBook Pro, Apple M1 Pro: benches/README.md:30-33): BSD humanize_number -- powers of 1024,
    rounded to nearest, one decimal below 10 units."""
    if nbytes < 1000:
        return "%dB" % nbytes
    for unit in "KMGT":
        nbytes /= 1024.0
        if nbytes < 999.5:
            return ("%.1f%s" % (nbytes, unit)) if nbytes < 9.95 else ("%d%s" % (int(nbytes + 0.5), unit))
    raise ValueError


def test_sizes_match_reference_readme(pkg):
    """The reference publishes |vk|, |pk| and the size of its proof file per k (benches/README.md:56-60 delay_enc, :64-78 mod_pow,
    :84-96 pose_enc); its bench appends one proof per criterion iteration to one transcript and writes that (benches/delay_enc.rs:
    120-134).  This repository's RawBytes layouts and proof layout must reproduce every one of them."""
    from dehalo2_amd import keygen, plonk, prover

    f = pkg.fields.BN254_FR
    cs = plonk.maingate_cs(True)                                     # delay_enc / mod_pow: MainGate + RangeChip, two selectors
    for k, vk_h, pk_h in ((15, "9.3K", "138M"), (16, "17K", "276M"), (17, "33K", "552M"), (18, "65K", "1.1G"), (19, "129K", "2.2G")):
        assert _mac_ls_h(keygen.vk_size(cs, k, 2)) == vk_h, k
        assert _mac_ls_h(keygen.pk_size(cs, k, 2, f)) == pk_h, k
    assert keygen.vk_size(cs, 15, 2) == 9544 and keygen.vk_size(cs, 17, 2) == 34120
    head, evals = prover.proof_layout(cs)
    per_proof = 32 * (head + evals + 4)                              # + one opening quotient per distinct point (x, wx, w^-1 x, w^last x)
    assert per_proof == 2848
    assert _mac_ls_h(101 * per_proof) == "281K" and _mac_ls_h(103 * per_proof) == "286K"      # k >= 16 / k = 15 rows of the README
    # pose_enc: MainGate only, no selectors -- |vk| is exact to the byte in the README
    q = plonk.maingate_cs(False)
    assert keygen.vk_size(q, 11, 0) == 968 and _mac_ls_h(968) == "968B"
    for k, pk_h in ((11, "4.1M"), (12, "8.3M"), (13, "17M")):
        assert _mac_ls_h(keygen.pk_size(q, k, 0, f)) == pk_h, k
    head, evals = prover.proof_layout(q)
    per_proof = 32 * (head + evals + 3)                              # x, wx, w^last x (no lookups: no w^-1 x)
    assert per_proof == 1792
    assert _mac_ls_h(131 * per_proof) == "229K" and _mac_ls_h(115 * per_proof) == "201K"      # K = 11 / K >= 12 rows


@pytest.mark.gpu
def test_written_keys_have_the_published_sizes(pkg, ctx, chain, device_chain):
    """The writers themselves (not only the size formulas): vk and pk of the delay_enc shape at k = 9 and of pose_enc at k = 11."""
    from dehalo2_amd import keygen

    for k, rl, nsel in ((9, True, 2), (11, False, 0)):
        d, c = device_chain(k, rl), chain(k, rl)
        buf = io.BytesIO()
        d["pk"].vk.write(buf)
        assert len(buf.getvalue()) == keygen.vk_size(c["circ"].cs, k, nsel)
        buf = io.BytesIO()
        d["pk"].write(buf)
        assert len(buf.getvalue()) == keygen.pk_size(c["circ"].cs, k, nsel, pkg.fields.BN254_FR)


@pytest.mark.gpu
def test_prover_intermediate_buffers_vs_oracle(pkg, po, co, ctx, chain, device_chain):
    """What a proof leaves in the prover's device buffers, phase by phase, against the C oracle (the assertions of round 1's schedule
    replay, now on the real prover): every committed column's MSM, its lagrange_to_coeff and coeff_to_extended, and the quotient's
    coefficients [UPSTREAM ParamsKZG::commit_lagrange, EvaluationDomain::{lagrange_to_coeff, coeff_to_extended, extended_to_coeff}]."""
    from dehalo2_amd import keygen, prover, transcript

    k, rl = 9, True
    c, d = chain(k, rl), device_chain(k, rl)
    f = pkg.fields.BN254_FR
    side = pkg.Context(0)
    P = fgp.Prover(d["params"], d["pk"], ctx, side)
    tr = transcript.Blake2bWrite(pkg.fields.BN254)
    P.create_proof(c["adv"], [[]], prover.SeededRng(7), tr)
    ctx.synchronize(); side.synchronize()
    want, trace = oracle_proof(po, c)
    assert tr.finalize() == want
    dom, e = d["pk"].domain, f.encode
    cols = keygen.to_host(P.cols)                      # Lagrange values with their blinding rows (the random polynomial lives in polys only)
    polys = keygen.to_host(P.polys)
    std = P.ext.clone()
    ctx.convert_form_device(f.id, std.data_ptr(), std.data_ptr(), std.numel() // 4, False, 0)
    ctx.synchronize()
    ext = keygen.to_host(std)
    nco = P.NC - 1
    commitments = trace["commitments"]
    assert len(commitments) == P.NC + dom.quotient_poly_degree + 4
    for col in range(nco):                             # advice | permuted (input, table)... | permutation z | lookup z: transcript order
        pt = co.to_affine(0, co.best_multiexp(0, cols[col], c["srs"]["g_lagrange"], 4))
        assert keygen.decode_points(pkg.fields.BN254, pt.reshape(1, 8))[0] == commitments[col], col
        coeff = co.lagrange_to_coeff(f.id, cols[col], k, e(dom.omega_inv), e(dom.ifft_divisor), 2)
        assert np.array_equal(polys[col], coeff), col
        if col % 4 == 0:
            assert np.array_equal(ext[col], co.coeff_to_extended(f.id, coeff, k, dom.extended_k, e(dom.extended_omega), e(dom.g_coset), 2)), col
    pt = co.to_affine(0, co.best_multiexp(0, polys[nco], c["srs"]["g"], 4))                     # the random polynomial: commit, not commit_lagrange
    assert keygen.decode_points(pkg.fields.BN254, pt.reshape(1, 8))[0] == commitments[nco]
    h = keygen.to_host(P.h)[: dom.n * dom.quotient_poly_degree]
    assert np.array_equal(h, np.asarray(trace["h_coeffs"])[: len(h)])
    side.close()
