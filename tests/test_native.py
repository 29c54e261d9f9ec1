"""The whole call behind the C ABI (include/dehalo.h "the whole call"; csrc/prover.hip): dehalo_params_*, dehalo_keygen, dehalo_pk_*,
dehalo_transcript_*, dehalo_create_proof(s) -- the reference's objects and call at benches/delay_enc.rs:41-54, 84-115, 120-134.

CPU suite -- what needs no device: the transcript against the Python mirror and hashlib, the three random-scalar sources, the
             constraint-system descriptor.
GPU suite -- native keygen == Python keygen (vk / pk bytes), native proofs == the CPU restatement's proofs byte for byte and accepted
             by the verifier (k = 6 / 9 / 11 / 17, with and without a side context), batch mode without the interpreter, formats,
             upstream's error cases.
"""
import ctypes as C
import hashlib
import io
import threading

import numpy as np
import pytest

from test_proof import chain, oracle_proof, oracle_verify, S_TOXIC      # noqa: F401  (fixtures)


# ---------------------------------------------------------------- CPU suite
def test_native_transcript_matches_the_python_mirror_and_hashlib(pkg, po):
    from dehalo2_amd import native, transcript

    curve = pkg.fields.BN254
    P, Q = po.ec_mul(po.BN254, 12345, (1, 2)), po.ec_mul(po.BN254, 0xDEADBEEF, (1, 2))
    t, m = native.Blake2bWrite(curve), transcript.Blake2bWrite(curve)
    for tr in (t, m):
        tr.common_scalar(7)
        tr.write_point(P)
        tr.write_scalar(curve.scalar.p - 1)
    c1 = t.squeeze_challenge_scalar()
    assert c1 == m.squeeze_challenge_scalar()
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
    h.update(b"\x02" + (7).to_bytes(32, "little") + b"\x01" + P[0].to_bytes(32, "little") + P[1].to_bytes(32, "little") + b"\x02" +
             (curve.scalar.p - 1).to_bytes(32, "little") + b"\x00")
    assert c1 == int.from_bytes(h.digest(), "little") % curve.scalar.p
    # long inputs cross Blake2b's 128-byte block boundary many times; challenges keep chaining
    for i in range(40):
        for tr in (t, m):
            tr.write_point(Q if i % 3 else P)
            tr.write_scalar(i * 0x1234567)
            if i % 7 == 0:
                assert tr.squeeze_challenge_scalar() is not None
    assert t.squeeze_challenge_scalar() == m.squeeze_challenge_scalar()
    assert t.finalize() == m.finalize() and len(t.finalize()) == 64 + 40 * 64
    with pytest.raises(ValueError):
        t.write_point(None)


def test_blake2b_block_boundaries(pkg):
    """The library's Blake2b against hashlib at every length around the 128-byte block size (scalars are absorbed as 33 bytes)."""
    from dehalo2_amd import native

    curve = pkg.fields.BN254
    p = curve.scalar.p
    for count in list(range(0, 12)) + [31, 32, 64, 127, 128, 129]:
        t = native.Blake2bWrite(curve)
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        for i in range(count):
            v = (i * 0x9E3779B97F4A7C15 + 5) % p
            t.common_scalar(v)
            h.update(b"\x02" + v.to_bytes(32, "little"))
        h.update(b"\x00")
        assert t.squeeze_challenge_scalar() == int.from_bytes(h.digest(), "little") % p, count


def test_native_pcg64_is_numpys_stream(pkg):
    """DEHALO_RNG_PCG64 must hand out exactly prover.SeededRng's scalars (numpy PCG64, four outputs per scalar, top word masked to 61
    bits), including after a skip (the helper thread's fork)."""
    from dehalo2_amd import native, prover
    from dehalo2_amd._lib import load_library

    lib = load_library()
    for seed in (0, 7, 0xDEADBEEF):
        want = prover.SeededRng(seed).scalars(50)
        r = native.rng_struct(prover.SeededRng(seed))
        out = np.zeros((50, 4), dtype=np.uint64)
        assert lib.dehalo_rng_scalars(C.byref(r), 0, 0, out.ctypes.data, 50) == 0
        assert np.array_equal(out, want)
        out2 = np.zeros((20, 4), dtype=np.uint64)
        assert lib.dehalo_rng_scalars(C.byref(r), 0, 30, out2.ctypes.data, 20) == 0
        assert np.array_equal(out2, want[30:])
        forked = prover.SeededRng(seed).fork(123457).scalars(3)
        out3 = np.zeros((3, 4), dtype=np.uint64)
        assert lib.dehalo_rng_scalars(C.byref(r), 0, 123457, out3.ctypes.data, 3) == 0
        assert np.array_equal(out3, forked)


@pytest.mark.parametrize("fname", ["bn254_fr", "pasta_fp", "pasta_fq"])
def test_native_os_rng_is_uniform_below_p(pkg, fname):
    """DEHALO_RNG_OS (the default: NULL): every scalar < p, no two draws alike, the top bits are not stuck (the whole field is covered,
    not only 253-bit values), and two calls give different streams."""
    from dehalo2_amd._lib import load_library
    from dehalo2_amd.keygen import array_to_ints

    f = pkg.fields.FIELDS[fname]
    lib = load_library()
    a, b = np.zeros((4096, 4), dtype=np.uint64), np.zeros((4096, 4), dtype=np.uint64)
    assert lib.dehalo_rng_scalars(None, f.id, 0, a.ctypes.data, 4096) == 0
    assert lib.dehalo_rng_scalars(None, f.id, 0, b.ctypes.data, 4096) == 0
    va, vb = array_to_ints(a), array_to_ints(b)
    assert all(v < f.p for v in va + vb)
    assert len(set(va + vb)) == 8192
    top = f.p.bit_length() - 1
    assert 0.2 < sum(v >> top for v in va) / 4096 < 0.8 if f.p >> (top - 1) == 3 else True      # bn254: p ~ 1.51 * 2^253
    assert max(va) > (f.p * 7) // 8 and min(va) < f.p // 8


def test_native_callback_rng(pkg):
    from dehalo2_amd import native, prover
    from dehalo2_amd._lib import load_library

    class Wrapped:      # not a SeededRng: goes through the callback
        def __init__(self):
            self.inner = prover.SeededRng(11)

        def scalars(self, count):
            return self.inner.scalars(count)

    r = native.rng_struct(Wrapped())
    assert r.kind == native.RNG_CALLBACK
    out = np.zeros((9, 4), dtype=np.uint64)
    assert load_library().dehalo_rng_scalars(C.byref(r), 0, 0, out.ctypes.data, 9) == 0
    assert np.array_equal(out, prover.SeededRng(11).scalars(9))


def test_constraint_system_descriptor(pkg):
    from dehalo2_amd import native, plonk

    f = pkg.fields.BN254_FR
    for rl in (True, False):
        cs = plonk.maingate_cs(rl)
        d = native.ConstraintSystemDescriptor(cs, f).struct
        assert (d.num_advice, d.num_fixed, d.num_instance) == (5, 15 if rl else 9, 1)
        assert d.num_gates == 1 and d.num_lookups == (5 if rl else 0) and d.num_permutation_columns == 6
        assert d.num_advice_queries == 6 and d.num_fixed_queries == (15 if rl else 9)
        nodes = [d.nodes[i] for i in range(d.num_nodes)]
        for i, nd in enumerate(nodes):      # children precede parents
            if nd.kind in (4, 7):
                assert nd.a < i
            if nd.kind in (5, 6):
                assert nd.a < i and nd.b < i
        assert [d.lookup_lens[i] for i in range(d.num_lookups)] == [2] * d.num_lookups


# ---------------------------------------------------------------- GPU suite
@pytest.fixture(scope="module")
def native_chain(pkg, ctx, chain):
    import pairing as pr
    from dehalo2_amd import native

    cache = {}

    def get(k, rl):
        if (k, rl) not in cache:
            c = chain(k, rl, threads=16)
            params = native.ParamsKZG.create(ctx, pkg.fields.BN254, k, c["srs"]["g"], c["srs"]["g_lagrange"], pr.g2_to_raw(pr.G2), pr.g2_to_raw(c["s_g2"]))
            pk = native.ProvingKey.keygen(ctx, params, c["circ"].cs, c["circ"].fixed, c["circ"].assembly, c["circ"].selectors)
            pk.transcript_repr = c["rep"]
            cache[(k, rl)] = dict(params=params, pk=pk)
        return cache[(k, rl)]

    yield get
    for v in cache.values():
        v["pk"].release()
        v["params"].release()


@pytest.mark.gpu
@pytest.mark.parametrize("k,rl", [(6, False), (9, True), (11, False), (14, True), (17, True)])
def test_native_proof_matches_oracle_and_verifies(pkg, po, ctx, chain, native_chain, k, rl):
    """dehalo_keygen + dehalo_create_proof: the verifying key's bytes and the whole proof equal the CPU restatement's, the verifier accepts
    -- configs[0] shape at k = 11, configs[3] shape at k = 17 and the north star's k = 14 included (k = 20: the next test) -- with and without a side context."""
    import plonk_oracle as PO
    from dehalo2_amd import native, prover

    c, d = chain(k, rl, threads=16), native_chain(k, rl)
    assert d["pk"].vk_bytes() == PO.vk_bytes(po.BN254, c["key"], c["circ"].selectors)
    want, _ = oracle_proof(po, c, threads=16)
    P = native.Prover(d["params"], d["pk"])
    rng = prover.SeededRng(7)
    proof = P.create_proof(c["adv"], [[]], rng).finalize()
    assert len(proof) == len(want)
    diff = [i // 32 for i in range(0, len(want), 32) if proof[i:i + 32] != want[i:i + 32]]
    assert not diff, "proof items differ from the oracle's: %r" % diff[:8]
    assert oracle_verify(po, c, proof, k)
    # the caller's generator has moved past the proof's draws, exactly as far as the CPU restatement moves it
    ref = PO.ScalarStream(7)
    oracle_proof_rng = PO.create_proof(po.BN254, c["srs"], c["key"], c["adv"], [[]], ref, c["rep"], 16)[0]
    assert oracle_proof_rng == want and np.array_equal(rng.scalars(2), ref.scalars(2))
    side = pkg.Context(0)
    P2 = native.Prover(d["params"], d["pk"], ctx, side)
    for _ in range(3):
        assert P2.create_proof(c["adv"], [[]], prover.SeededRng(7)).finalize() == want
    t = P2.last_timings()
    assert t["total"] > 0 and abs(sum(t[p] for p in native.PHASES[:-1]) - t["total"]) < 0.5 * t["total"] + 0.5
    # OS entropy (rng = None): a different, valid proof every time
    p1, p2 = P2.create_proof(c["adv"], [[]]).finalize(), P2.create_proof(c["adv"], [[]]).finalize()
    assert p1 != p2 != want and len(p1) == len(want)
    p3 = P.create_proof(c["adv"], [[]]).finalize()                     # no side context: the random polynomial's kernel runs on the main stream
    assert p3 != p1 and len(p3) == len(want)
    if k <= 11:
        assert oracle_verify(po, c, p1, k) and oracle_verify(po, c, p2, k) and oracle_verify(po, c, p3, k)
    P.release(); P2.release(); side.close()


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_native_proof_k20_matches_oracle_and_verifies(pkg, po, co, ctx):
    """The north star's largest size, k = 20 (delay_enc shape): the only size whose proof runs c = 16 windows, 32,768-bucket reduction groups and the
    2^20-row distinct-table lookup path.  Verifying key and proof equal the CPU restatement's byte for byte (its SRS, keygen and proof take a few minutes of
    host time), the pairing check accepts, with and without a side context.  Nothing of this size is kept for the other tests."""
    import pairing as pr
    import plonk_oracle as PO
    import shapes
    import verifier as V
    from dehalo2_amd import circuits, native, prover

    k, threads, curve = 20, 16, po.BN254
    circ = circuits.synthesize(curve.scalar.p, k, True, seed=3)
    desc = shapes.maingate_description(True)
    assert desc == circ.cs.description()
    srs = PO.setup_srs(curve, k, S_TOXIC, threads)
    s_g2 = pr.g2_mul(S_TOXIC, pr.G2)
    # the device's own ParamsKZG::setup from the same secret: same bytes as the CPU restatement's SRS at the size bench.py reports
    params = native.ParamsKZG.setup(ctx, pkg.fields.BN254, k, S_TOXIC)
    raw, n = params.write(), 1 << k
    assert raw[4:4 + 64 * n] == np.ascontiguousarray(srs["g"]).tobytes() and raw[4 + 64 * n:4 + 128 * n] == np.ascontiguousarray(srs["g_lagrange"]).tobytes()
    del raw
    key = PO.keygen(curve, srs, desc, k, circ.fixed, circ.assembly.mapping, threads)
    rep = PO.transcript_repr(curve, key, circ.selectors)
    adv = np.stack([co.field_op(PO.Fld(curve.scalar).id, "to_mont", circ.advice[i]) for i in range(5)])
    pk = native.ProvingKey.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    assert pk.vk_bytes() == PO.vk_bytes(curve, key, circ.selectors)
    pk.transcript_repr = rep
    want, _ = PO.create_proof(curve, srs, key, adv, [[]], PO.ScalarStream(7), rep, threads)
    P = native.Prover(params, pk)
    proof = P.create_proof(adv, [[]], prover.SeededRng(7)).finalize()
    diff = [i // 32 for i in range(0, len(want), 32) if proof[i:i + 32] != want[i:i + 32]]
    assert len(proof) == len(want) and not diff, "k = 20 proof items differ from the oracle's: %r" % diff[:8]
    assert V.verify_proof(curve, desc, k, key["fixed_commitments"], key["perm_commitments"], rep, (1, 2), pr.G2, s_g2, [[]], proof)
    side = pkg.Context(0)
    P2 = native.Prover(params, pk, ctx, side)
    for _ in range(2):
        assert P2.create_proof(adv, [[]], prover.SeededRng(7)).finalize() == want
    P.release(); P2.release(); side.close(); pk.release(); params.release()


@pytest.mark.gpu
def test_native_keygen_equals_python_keygen(pkg, ctx, chain, native_chain):
    """Same vk and pk bytes as keygen.py's (the fine-grained path): RawBytes layouts, commitments, cosets."""
    import pairing as pr
    from dehalo2_amd import keygen

    for k, rl in ((9, True), (6, False)):
        c, d = chain(k, rl), native_chain(k, rl)
        params = keygen.ParamsKZG(ctx, pkg.fields.BN254, k, c["srs"]["g"], c["srs"]["g_lagrange"], pr.g2_to_raw(pr.G2), pr.g2_to_raw(c["s_g2"]))
        pk = keygen.keygen(ctx, params, c["circ"].cs, c["circ"].fixed, c["circ"].assembly, c["circ"].selectors)
        b = io.BytesIO(); pk.write(b)
        assert d["pk"].write() == b.getvalue()
        b = io.BytesIO(); params.write(b)
        assert d["params"].write() == b.getvalue()
        assert len(d["pk"].write()) == keygen.pk_size(c["circ"].cs, k, 2 if rl else 0, pkg.fields.BN254_FR)
        params.release()


@pytest.mark.gpu
def test_native_formats_roundtrip_and_prove(pkg, po, ctx, chain, native_chain):
    """dehalo_params_read / dehalo_pk_read of what the writers produced: same bytes out again; a proof from the re-read objects is the same
    proof; truncated or foreign input is refused."""
    from dehalo2_amd import native, prover

    k, rl = 9, True
    c, d = chain(k, rl), native_chain(k, rl)
    curve = pkg.fields.BN254
    raw_params, raw_pk = d["params"].write(), d["pk"].write()
    p2 = native.ParamsKZG.read(ctx, curve, raw_params)
    pk2 = native.ProvingKey.read(ctx, curve, c["circ"].cs, raw_pk, num_selectors=2)
    assert p2.write() == raw_params and pk2.write() == raw_pk
    # the substitute transcript_repr is a function of the key and the circuit only
    fresh = native.ProvingKey.read(ctx, curve, c["circ"].cs, raw_pk, num_selectors=2)
    assert fresh.transcript_repr == pk2.transcript_repr != 0
    pk2.transcript_repr = c["rep"]
    want, _ = oracle_proof(po, c)
    P = native.Prover(p2, pk2)
    assert P.create_proof(c["adv"], [[]], prover.SeededRng(7)).finalize() == want
    with pytest.raises(pkg.DehaloError):
        native.ProvingKey.read(ctx, curve, c["circ"].cs, raw_pk[:-5], num_selectors=2)
    with pytest.raises(pkg.DehaloError):
        native.ProvingKey.read(ctx, curve, c["circ"].cs, raw_pk, num_selectors=0)
    with pytest.raises(pkg.DehaloError):
        native.ParamsKZG.read(ctx, curve, raw_params[:-1])
    P.release(); fresh.release(); pk2.release(); p2.release()


@pytest.mark.gpu
def test_native_transcript_continues_across_proofs(pkg, po, ctx, chain, native_chain):
    """The reference's bench appends every criterion iteration's proof to ONE transcript (benches/delay_enc.rs:120-134): the second proof's
    challenges depend on the first; the first 1792 bytes are still the single proof."""
    from dehalo2_amd import native, prover

    c, d = chain(6, False), native_chain(6, False)
    want, _ = oracle_proof(po, c)
    P = native.Prover(d["params"], d["pk"])
    tr = native.Blake2bWrite(pkg.fields.BN254)
    P.create_proof(c["adv"], [[]], prover.SeededRng(7), tr)
    P.create_proof(c["adv"], [[]], prover.SeededRng(7), tr)
    both = tr.finalize()
    assert len(both) == 2 * len(want) and both[:len(want)] == want
    assert both[len(want):len(want) + 5 * 32] == want[:5 * 32]               # same advice commitments (same blinding) ...
    assert both[len(want) + 5 * 32:] != want[5 * 32:]                        # ... but theta and everything after it differ
    P.release()


def plonk_mod(pkg):
    from dehalo2_amd import plonk
    return plonk


@pytest.mark.gpu
def test_native_device_advice_and_instances(pkg, po, ctx, chain, native_chain):
    """advice as a device tensor; upstream's InvalidInstances / InstanceTooLarge; one wrong witness cell -> rejected proof."""
    import torch
    from dehalo2_amd import native, prover
    from test_proof import arith_row

    c, d = chain(6, False), native_chain(6, False)
    want, _ = oracle_proof(po, c)
    P = native.Prover(d["params"], d["pk"])
    dev = ctx.upload(c["adv"])
    assert P.create_proof(dev, [[]], prover.SeededRng(7)).finalize() == want
    with pytest.raises(ValueError):
        P.create_proof(c["adv"], [], prover.SeededRng(7))
    with pytest.raises(ValueError):
        P.create_proof(c["adv"], [[1] * 64], prover.SeededRng(7))
    assert P.create_proof(c["adv"], [[]], prover.SeededRng(7)).finalize() == want      # the prover is usable after an error
    adv = c["adv"].copy()
    adv[1, arith_row(c)] = pkg.fields.BN254_FR.encode(987654321)
    assert not oracle_verify(po, c, P.create_proof(adv, [[]], prover.SeededRng(7)).finalize(), 6)
    P.release()
    # a range-checked cell holding a value that is not in the table: upstream's permute_expression_pair returns Err(ConstraintSystemFailure) -- here
    # the flag travels back with the permuted columns' commitments (no synchronisation of its own) and fails the call just the same
    c, d = chain(9, True), native_chain(9, True)
    P = native.Prover(d["params"], d["pk"])
    fixed = c["circ"].fixed
    row = next(r for r in range(c["circ"].used_rows) if np.any(fixed[plonk_mod(pkg).RC_S_COMPOSITION][r]))
    adv = c["adv"].copy()
    adv[0, row] = pkg.fields.BN254_FR.encode(0x1234567)
    with pytest.raises(pkg.DehaloError) as e:
        P.create_proof(adv, [[]], prover.SeededRng(7))
    assert e.value.code == -6 and "not in the table" in str(e.value)
    want, _ = oracle_proof(po, c)
    assert P.create_proof(c["adv"], [[]], prover.SeededRng(7)).finalize() == want      # and the prover is usable afterwards
    P.release()


@pytest.mark.gpu
def test_native_batch_mode(pkg, po, ctx, chain, native_chain):
    """dehalo_create_proofs (BASELINE configs[4] on one GPU): four provers on four contexts, one library thread each, no interpreter in
    the loop; every proof equals the proof a lone prover with a side context makes from the same seed."""
    from dehalo2_amd import native, prover

    k, rl = 9, True
    c, d = chain(k, rl), native_chain(k, rl)
    side = pkg.Context(0)
    lone = native.Prover(d["params"], d["pk"], ctx, side)
    seeds = list(range(200, 232))
    alone = [lone.create_proof(c["adv"], [[]], prover.SeededRng(s)).finalize() for s in seeds]
    assert len(set(alone)) == len(seeds)
    want, _ = oracle_proof(po, c, seed=seeds[5])
    assert alone[5] == want
    ctxs = [pkg.Context(0) for _ in range(4)]
    provers = [native.Prover(d["params"], d["pk"], cx) for cx in ctxs]
    for rep in range(2):
        got = native.create_proofs(provers, c["adv"], [prover.SeededRng(s) for s in seeds])
        bad = [s for s, a, g in zip(seeds, alone, got) if a != g]
        assert not bad, "batch-mode proofs differ from the lone prover's for seeds %r" % bad
    assert oracle_verify(po, c, got[-1], k)
    # the same from Python threads, one create_proof call each
    res = {}

    def work(j):
        for i in range(j, len(seeds), 4):
            res[i] = provers[j].create_proof(c["adv"], [[]], prover.SeededRng(seeds[i])).finalize()
    th = [threading.Thread(target=work, args=(j,)) for j in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    assert [res[i] for i in range(len(seeds))] == alone
    for p in provers:
        p.release()
    lone.release(); side.close()
    for cx in ctxs:
        cx.close()


def test_library_field_constants_are_upstreams(pkg):
    """dehalo_field_info: the constants the C++ prover derives its domain from -- modulus, R, ROOT_OF_UNITY = g^((p-1) >> S), ZETA (the
    crates' literal: it fixes which coset the proving key's extended columns are evaluated on), DELTA = g^(2^S)."""
    from dehalo2_amd._lib import load_library
    from dehalo2_amd.keygen import array_to_ints, delta_of

    lib = load_library()
    for f in pkg.fields.FIELDS.values():
        out = np.zeros((6, 4), dtype=np.uint64)
        assert lib.dehalo_field_info(f.id, out.ctypes.data) == 0
        v = array_to_ints(out)
        rinv = pow(1 << 256, -1, f.p)
        assert v[0] == f.p and v[1] == (1 << 256) % f.p
        assert v[2] * rinv % f.p == f.root_of_unity and v[3] * rinv % f.p == f.zeta and v[4] * rinv % f.p == delta_of(f) and v[5] * rinv % f.p == f.gen
        assert pow(f.zeta, 3, f.p) == 1 and f.zeta != 1


@pytest.mark.gpu
def test_cpp_host_example_proves_without_an_interpreter(pkg, po, ctx, chain, tmp_path):
    """host/example.cpp: ParamsKZG::read -> keygen -> ProvingKey::write -> ProvingKey::read -> create_proof, all from C++ over the C ABI
    (the reference's bench flow, benches/pose_enc.rs:41-135).  The files it writes: pk / vk bytes equal the Python keygen's, the proof
    equals the CPU restatement's and is accepted by the verifier."""
    import os
    import subprocess

    import pairing as pr
    import plonk_oracle as PO
    from dehalo2_amd import keygen, prover
    from conftest import ROOT

    exe = os.path.join(ROOT, "delay-encryption-in-halo2_amd", "host", "example")
    assert os.path.exists(exe), "host example not built (make host_example)"
    k = 6
    c = chain(k, False)
    f = pkg.fields.BN254_FR
    n = 1 << k
    # no params.bin: the program makes the SRS itself (ParamsKZG::setup on the device) from the secret and caches it, as the reference's bench does
    want_params = (k.to_bytes(4, "little") + np.ascontiguousarray(c["srs"]["g"]).tobytes() + np.ascontiguousarray(c["srs"]["g_lagrange"]).tobytes() +
                   pr.g2_to_raw(pr.G2) + pr.g2_to_raw(c["s_g2"]))
    (tmp_path / "secret.bin").write_bytes(f.encode(S_TOXIC).tobytes())
    fixed_m = np.stack([ctx.field_op(f.id, "to_mont", c["circ"].fixed[i]) for i in range(9)])
    st = prover.SeededRng(7).gen.bit_generator.state["state"]
    (tmp_path / "circuit.bin").write_bytes(k.to_bytes(4, "little") + fixed_m.tobytes() + np.ascontiguousarray(c["circ"].assembly.mapping, dtype=np.uint64).tobytes() +
                                           np.ascontiguousarray(c["adv"]).tobytes() + f.encode(c["rep"]).tobytes() + st["state"].to_bytes(16, "little") +
                                           st["inc"].to_bytes(16, "little"))
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    want, _ = oracle_proof(po, c)
    proof = (tmp_path / "proof.bin").read_bytes()
    assert (tmp_path / "params.bin").read_bytes() == want_params, "the SRS the program made differs from the CPU restatement's"
    assert proof == want and oracle_verify(po, c, proof, k)
    out2 = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)      # second run: params.bin is read, same proof
    assert out2.returncode == 0 and (tmp_path / "proof.bin").read_bytes() == want
    assert (tmp_path / "vk.bin").read_bytes() == PO.vk_bytes(po.BN254, c["key"], c["circ"].selectors)
    assert len((tmp_path / "pk.bin").read_bytes()) == keygen.pk_size(c["circ"].cs, k, 0, f)


@pytest.mark.gpu
def test_native_witness_to_proof_end_to_end(pkg, po, co, ctx):
    """dehalo_synthesize -> dehalo_keygen -> dehalo_create_proof(DEHALO_PROOF_ADVICE_CANONICAL): the reference's whole timed call
    (synthesize + prove, benches/delay_enc.rs:123-131 over src/lib.rs:164-318) without Python integers anywhere; the proof equals the
    CPU restatement's proof of witness.py's (identical) columns and is accepted by the verifier."""
    import json
    import os

    import pairing as pr
    import plonk_oracle as PO
    import verifier as V
    from conftest import ROOT
    from dehalo2_amd import native, plonk, prover

    v = json.load(open(os.path.join(ROOT, "tests", "golden", "rsa_vectors.json")))[1]
    n, x, k = int(v["n"]), int(v["signature"]), 16
    nat = native.synthesize(native.CIRCUIT_DELAY_ENC, k, n_big=n, e=0b10011, x=x, exp_bits=5, message=[0, 0], keygen=True)
    assert nat["rsa_result"] == pow(x, 0b10011, n)
    cs = plonk.maingate_cs(True)
    asm = plonk.Assembly(6, 1 << k)
    asm.mapping = nat["mapping"].astype(np.int64)
    s = 0x5EED5EED5EED5EED
    srs = PO.setup_srs(po.BN254, k, s, 16)
    import shapes
    assert shapes.maingate_description(True) == cs.description()
    key = PO.keygen(po.BN254, srs, shapes.maingate_description(True), k, nat["fixed"], asm.mapping, 16)
    rep = PO.transcript_repr(po.BN254, key, nat["selectors"])
    params = native.ParamsKZG.create(ctx, pkg.fields.BN254, k, srs["g"], srs["g_lagrange"])
    pk = native.ProvingKey.keygen(ctx, params, cs, nat["fixed"], asm, nat["selectors"])
    assert pk.vk_bytes() == PO.vk_bytes(po.BN254, key, nat["selectors"])
    pk.transcript_repr = rep
    side = pkg.Context(0)
    P = native.Prover(params, pk, ctx, side)
    proof = P.create_proof(nat["advice"], [[]], prover.SeededRng(5), canonical=True).finalize()
    adv_m = np.stack([co.field_op(0, "to_mont", nat["advice"][i]) for i in range(5)])
    want, _ = PO.create_proof(po.BN254, srs, key, adv_m, [[]], PO.ScalarStream(5), rep, 16)
    assert proof == want
    assert V.verify_proof(po.BN254, shapes.maingate_description(True), k, key["fixed_commitments"], key["perm_commitments"], rep, (1, 2), pr.G2, pr.g2_mul(s, pr.G2), [[]], proof)
    assert P.create_proof(adv_m, [[]], prover.SeededRng(5)).finalize() == want              # Montgomery input, same proof
    # the reference's own call shape -- create_proof(&params, &pk, &[circuit], ...) synthesizes inside (dehalo_create_proof_circuit): same proof, same
    # summary, with and without a side context, twice in a row (the prover's page-locked advice buffer is reused), and the caller's generator moves as far
    inputs = dict(n_big=n, e=0b10011, x=x, exp_bits=5, message=[0, 0])
    rng, ref = prover.SeededRng(5), PO.ScalarStream(5)
    PO.create_proof(po.BN254, srs, key, adv_m, [[]], ref, rep, 16)
    for rep_i in range(2):
        tr, info = P.create_proof_circuit(native.CIRCUIT_DELAY_ENC, [[]], rng if rep_i == 0 else prover.SeededRng(5), **inputs)
        assert tr.finalize() == want and info["rsa_result"] == pow(x, 0b10011, n) and info["rows"] == nat["rows"] and info["cipher"] == nat["cipher"]
    assert np.array_equal(rng.scalars(2), ref.scalars(2))
    lone = native.Prover(params, pk, ctx, None)
    assert lone.create_proof_circuit(native.CIRCUIT_DELAY_ENC, [[]], prover.SeededRng(5), **inputs)[0].finalize() == want
    with pytest.raises(pkg.DehaloError):                                                     # a circuit that does not fit the key's 2^k rows
        lone.create_proof_circuit(native.CIRCUIT_DELAY_ENC, [[]], prover.SeededRng(5), **dict(inputs, e=(1 << 14) | 1, exp_bits=15))
    assert lone.create_proof_circuit(native.CIRCUIT_DELAY_ENC, [[]], prover.SeededRng(5), **inputs)[0].finalize() == want      # and the prover is usable afterwards
    lone.release()
    # four provers, four library threads, ONE host witness array of 10 MiB: each call page-locks it for its upload (HostPin); the registration is shared
    # and released by the last one out
    ctxs = [pkg.Context(0, priority=(1, 0, -1)[i % 3]) for i in range(4)]
    provers4 = [native.Prover(params, pk, cx) for cx in ctxs]
    for _ in range(2):
        got = native.create_proofs(provers4, adv_m, [prover.SeededRng(5) for _ in range(12)])
        assert all(g == want for g in got)
    # the same with every proof's circuit synthesized inside its call; proof 3 has another exponent (its own witness), all others the one above
    ins = [dict(inputs) for _ in range(8)]
    ins[3] = dict(inputs, e=0b10101)
    got = native.create_proofs_circuit(provers4, native.CIRCUIT_DELAY_ENC, ins, [prover.SeededRng(5) for _ in range(8)])
    assert all(g == want for i, g in enumerate(got) if i != 3) and got[3] != want
    assert got[3] == P.create_proof_circuit(native.CIRCUIT_DELAY_ENC, [[]], prover.SeededRng(5), **ins[3])[0].finalize()
    for q in provers4:
        q.release()
    for cx in ctxs:
        cx.close()
    P.release(); pk.release(); params.release(); side.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [6, 11, 17])
def test_params_setup_on_the_device_equals_the_cpu_restatement(pkg, po, ctx, k):
    """ParamsKZG::setup (benches/delay_enc.rs:43) through dehalo_params_setup: g[i] = [s^i] G and g_lagrange[i] = [L_i(s)] G by fixed-base table multiplication
    on the device, g2 / s_g2 on the host.  The RawBytes serialisation must equal, byte for byte, the one assembled from oracle/plonk_oracle.setup_srs and
    oracle/pairing.py (the generator of G2 and [s] G2) for the same s; and a commitment made with the device-made tables equals the CPU's."""
    import time
    import pairing as pr
    import plonk_oracle as PO
    from dehalo2_amd import native

    curve = pkg.fields.BN254
    s = 0x64656C6179656E63 * 0x9E3779B97F4A7C15 % curve.scalar.p
    t0 = time.perf_counter()
    params = native.ParamsKZG.setup(ctx, curve, k, s)
    dt = time.perf_counter() - t0
    got = params.write()
    srs = PO.setup_srs(po.BN254, k, s, 8)
    want = (k.to_bytes(4, "little") + np.ascontiguousarray(srs["g"]).tobytes() + np.ascontiguousarray(srs["g_lagrange"]).tobytes() +
            pr.g2_to_raw(pr.G2) + pr.g2_to_raw(pr.g2_mul(s, pr.G2)))
    assert len(got) == len(want) == 4 + 128 * (1 << k) + 256
    assert got[:4] == want[:4]
    n = 1 << k
    assert got[4:4 + 64 * n] == want[4:4 + 64 * n], "g differs"
    assert got[4 + 64 * n:4 + 128 * n] == want[4 + 64 * n:4 + 128 * n], "g_lagrange differs"
    assert got[4 + 128 * n:] == want[4 + 128 * n:], "g2 / s_g2 differ"
    print("dehalo_params_setup k = %d: %.1f ms" % (k, 1e3 * dt))
    params.release()


@pytest.mark.gpu
def test_params_setup_k20_and_its_limits(pkg, ctx):
    """The largest north-star size: 2 x 2^20 fixed-base multiplications, the tables of both vectors and the download of the points for write() (the time is
    printed, not asserted: bench.py reports it as one_time_setup_s.params_setup_gpu); k = 26, whose precomputed tables would pass the 2^30 index limit, is refused
    before any device work; a curve without a pairing is refused."""
    import time
    from dehalo2_amd import native

    curve = pkg.fields.BN254
    native.ParamsKZG.setup(ctx, curve, 10, 12345).release()          # (first use: code objects, workspace)
    t0 = time.perf_counter()
    params = native.ParamsKZG.setup(ctx, curve, 20, 0x1234567890ABCDEF)
    dt = time.perf_counter() - t0
    print("dehalo_params_setup k = 20: %.1f ms" % (1e3 * dt))
    assert len(params.write()) == 4 + 128 * (1 << 20) + 256
    params.release()
    t0 = time.perf_counter()
    with pytest.raises(pkg.DehaloError, match="too large"):
        native.ParamsKZG.setup(ctx, curve, 26, 0x1234567890ABCDEF)
    assert time.perf_counter() - t0 < 5.0, "k = 26 must be refused before the 2^27 fixed-base multiplications, not after"      # (those take tens of seconds)
    with pytest.raises(Exception):
        native.ParamsKZG.setup(ctx, pkg.fields.CURVES["pallas"], 6, 3)      # no pairing: not a KZG curve
