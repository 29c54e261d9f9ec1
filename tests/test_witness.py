"""Witness generation: the reference's native helpers pinned by the reference's own vectors, the
laid-out rows checked against the constraint system, and proofs made from the real witness."""
import json
import os

import numpy as np
import pytest

import fine_grained_prover as fgp      # the Python driver over the fine-grained entry points: test infrastructure (was dehalo2_amd/prover.py's Prover)

from conftest import golden

HERE = os.path.dirname(os.path.abspath(__file__))


def rsa_vectors():
    return json.load(open(os.path.join(HERE, "golden", "rsa_vectors.json")))


def test_big_pow_mod_on_reference_rsa_vectors(pkg):
    """src/rsa/chip.rs:706-716, 751-761: signature^65537 mod n is the PKCS#1 v1.5 encoding of the SHA-256 digest -- exactly the
    limb values RSAChip::verify_pkcs1v15_signature compares against (src/rsa/chip.rs:140-203)."""
    from dehalo2_amd import witness as W

    for v in rsa_vectors():
        n, s, d = int(v["n"]), int(v["signature"]), int(v["sha256_digest"])
        em = W.big_pow_mod(s, v["e"], n)
        limbs = W.limbs_of(em)
        assert limbs[:4] == W.limbs_of(d, 4)
        assert limbs[4] == 217300885422736416 and limbs[5] == 938447882527703397           # DigestInfo prefix, src/rsa/chip.rs:152-155
        assert limbs[6] == (0xFFFFFFFF << 32) | 3158320 and all(l == (1 << 64) - 1 for l in limbs[7:31]) and limbs[31] == 562949953421311
    assert W.big_pow_mod(5, 0, 7) == 1 and W.big_pow_mod(3, 5, 1000) == 243


def test_poseidon_spec_reproduces_reference_kats(pkg):
    """src/poseidon/permutation.rs:154-158, 190-196 through the package's own Grain / MDS / permutation restatement."""
    from dehalo2_amd import witness as W

    p = pkg.fields.BN254_FR.p
    num = lambda x: int(x, 0) if isinstance(x, str) else int(x)
    for kat in golden("poseidon_kat"):
        assert W.PoseidonSpec(p, kat["t"], kat["r_f"], kat["r_p"]).permute([num(x) for x in kat["input"]]) == [num(x) for x in kat["expected"]]


@pytest.fixture(scope="module")
def small_delay_witness(pkg):
    from dehalo2_amd import witness as W

    v = rsa_vectors()[0]
    p = pkg.fields.BN254_FR.p
    return W.delay_enc_witness(p, 14, int(v["n"]), 0b1, int(v["signature"]), 1, [0, 0]), int(v["n"]), int(v["signature"])


def test_delay_enc_rows_satisfy_the_constraint_system(pkg, small_delay_witness):
    from dehalo2_amd import witness as W

    (circ, info), n, x = small_delay_witness
    p = pkg.fields.BN254_FR.p
    assert info.rsa_result == pow(x, 1, n) and info.total_rows == circ.used_rows
    assert W.check_rows(circ, p) == info.total_rows
    # the native cipher of the same key gives the in-circuit ciphertext (src/lib.rs:270-312)
    assert len(info.cipher) == 3
    # 3,974 rows per mul_mod, two of them and 32 selects per exponent bit (benches/README.md:77-78: 7,981 rows per bit), 1,860 rows around them
    assert info.rsa_rows == W.rsa_region_rows(1) == 1860 + 7981 + 1 and W.mul_mod_rows() == 3974
    assert info.total_rows == info.rsa_rows + W.hash_region_rows() + W.cipher_region_rows(2, True) == info.rsa_rows + 2182 + 1453
    # value classes of the witness: mostly small values
    from dehalo2_amd.keygen import array_to_ints
    vals = [v for c in range(5) for v in array_to_ints(circ.advice[c])[:circ.used_rows]]
    small = sum(1 for v in vals if v < (1 << 8)) / len(vals)
    wide = sum(1 for v in vals if v >= (1 << 134)) / len(vals)
    assert small > 0.4 and wide < 0.35            # (the 1-bit exponent of this test leaves the Poseidon rows a large share)


def test_pose_enc_witness_fits_the_reference_k(pkg):
    """benches/pose_enc.rs:184 runs PoseidonEncCircuit at K = 11: the cipher region alone fits 2^11 rows, satisfies the
    MainGate-only constraint system, and carries the native ciphertext."""
    from dehalo2_amd import witness as W

    p = pkg.fields.BN254_FR.p
    circ, info = W.pose_enc_witness(p, 11, [5, 7], [0, 0])
    assert circ.cs.num_fixed == 9 and not circ.cs.lookups and info.total_rows < (1 << 11) - 6
    assert W.check_rows(circ, p) == info.total_rows
    assert info.cipher == W.NativeCipher(W.PoseidonSpec(p, 5, 8, 57), [5, 7]).encrypt([0, 0], 1)
    # the in-circuit cipher adds the message twice, the native one never to the state it permutes (witness.py's header): a non-zero message is
    # unsatisfiable in the reference's own circuit, and refused here as MockProver would refuse it
    with pytest.raises(W.NotSatisfied):
        W.pose_enc_witness(p, 11, [5, 7], [11, 22])


def test_a_broken_witness_is_caught(pkg, small_delay_witness):
    from dehalo2_amd import witness as W

    (circ, info), _, _ = small_delay_witness
    bad = W.SyntheticCircuit(circ.cs, circ.k, circ.fixed, circ.advice.copy(), circ.assembly, circ.selectors, circ.used_rows)
    bad.advice[3, 1000, 0] ^= np.uint64(1)
    with pytest.raises(AssertionError):
        W.check_rows(bad, pkg.fields.BN254_FR.p)


def _oracle_chain(po, co, circ, k, threads):
    import plonk_oracle as PO
    import pairing as pr

    s = 0x5EED5EED5EED5EED
    srs = PO.setup_srs(po.BN254, k, s, threads)
    import shapes
    desc = shapes.maingate_description(bool(circ.cs.lookups))      # the checker's own statement of the shape; the product's must equal it
    assert desc == circ.cs.description()
    key = PO.keygen(po.BN254, srs, desc, k, circ.fixed, circ.assembly.mapping, threads)
    rep = PO.transcript_repr(po.BN254, key, circ.selectors)
    adv = np.stack([co.field_op(0, "to_mont", circ.advice[i]) for i in range(5)])
    return dict(srs=srs, desc=desc, key=key, rep=rep, adv=adv, s_g2=pr.g2_mul(s, pr.G2), s=s)


def test_oracle_proof_of_the_real_witness_is_accepted(pkg, po, co, small_delay_witness):
    import pairing as pr
    import plonk_oracle as PO
    import verifier as V
    from dehalo2_amd import prover

    (circ, info), _, _ = small_delay_witness
    c = _oracle_chain(po, co, circ, 14, 8)
    proof, _ = PO.create_proof(po.BN254, c["srs"], c["key"], c["adv"], [[]], PO.ScalarStream(5), c["rep"], 8)
    assert V.verify_proof(po.BN254, c["desc"], 14, c["key"]["fixed_commitments"], c["key"]["perm_commitments"], c["rep"], (1, 2), pr.G2, c["s_g2"], [[]], proof)


@pytest.mark.gpu
def test_device_proof_of_the_real_pose_enc_witness(pkg, po, co, ctx):
    """BASELINE configs[0]: PoseidonEncCircuit at K = 11 (benches/pose_enc.rs:127-135, 184)."""
    import pairing as pr
    import plonk_oracle as PO
    import verifier as V
    from dehalo2_amd import keygen, prover, transcript, witness as W

    circ, info = W.pose_enc_witness(pkg.fields.BN254_FR.p, 11, [0xABCDEF, 0x123456], [0, 0])
    c = _oracle_chain(po, co, circ, 11, 8)
    params = keygen.ParamsKZG(ctx, pkg.fields.BN254, 11, c["srs"]["g"], c["srs"]["g_lagrange"])
    pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    tr = transcript.Blake2bWrite(pkg.fields.BN254)
    fgp.Prover(params, pk).create_proof(c["adv"], [[]], prover.SeededRng(5), tr)
    want, _ = PO.create_proof(po.BN254, c["srs"], c["key"], c["adv"], [[]], PO.ScalarStream(5), c["rep"], 8)
    assert tr.finalize() == want
    assert V.verify_proof(po.BN254, c["desc"], 11, c["key"]["fixed_commitments"], c["key"]["perm_commitments"], c["rep"], (1, 2), pr.G2, c["s_g2"], [[]], want)
    params.release()


@pytest.mark.gpu
def test_device_proof_of_the_real_delay_enc_witness(pkg, po, co, ctx):
    """DelayEncryptCircuit with the reference's checked-in parameters (2048-bit modulus, EXP_LIMB_BITS = 5 -> k = 16,
    src/lib.rs:122-124, benches/delay_enc.rs:181): real witness -> device proof == CPU restatement's proof, verifier accepts."""
    import pairing as pr
    import plonk_oracle as PO
    import verifier as V
    from dehalo2_amd import keygen, prover, transcript, witness as W

    v = rsa_vectors()[1]
    n, x = int(v["n"]), int(v["signature"])
    circ, info = W.delay_enc_witness(pkg.fields.BN254_FR.p, 16, n, 0b10011, x, 5, [0, 0])
    assert info.rsa_result == pow(x, 0b10011, n)
    c = _oracle_chain(po, co, circ, 16, 16)
    params = keygen.ParamsKZG(ctx, pkg.fields.BN254, 16, c["srs"]["g"], c["srs"]["g_lagrange"])
    pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    tr = transcript.Blake2bWrite(pkg.fields.BN254)
    fgp.Prover(params, pk).create_proof(c["adv"], [[]], prover.SeededRng(5), tr)
    want, _ = PO.create_proof(po.BN254, c["srs"], c["key"], c["adv"], [[]], PO.ScalarStream(5), c["rep"], 16)
    assert tr.finalize() == want
    assert V.verify_proof(po.BN254, c["desc"], 16, c["key"]["fixed_commitments"], c["key"]["perm_commitments"], c["rep"], (1, 2), pr.G2, c["s_g2"], [[]], want)
    params.release()


@pytest.mark.gpu
def test_device_proof_of_the_metrics_own_witness_k17(pkg, po, co, ctx):
    """BASELINE's metric configuration itself (configs[3]; benches/delay_enc.rs:123-131, benches/README.md:60): DelayEncryptCircuit, 2048-bit modulus, 15-bit
    exponent, k = 17 -- 125,214 rows in halo2wrong's layout of the checked-in source (RSA region 121,579 = the published 121,578 + 1).  The witness is
    synthesized by the library (dehalo_synthesize), checked by the checker's own MockProver, proved through the reference's call shape
    (dehalo_create_proof_circuit): proof bytes equal the CPU restatement's, the verifier accepts."""
    import pairing as pr
    import plonk_oracle as PO
    import rowcheck
    import shapes
    import verifier as V
    from dehalo2_amd import native, plonk, prover

    v = rsa_vectors()[1]
    n, x, k, bits = int(v["n"]), int(v["signature"]), 17, 15
    e = 0b101101110010111
    inputs = dict(n_big=n, e=e, x=x, exp_bits=bits, message=[0, 0])
    nat = native.synthesize(native.CIRCUIT_DELAY_ENC, k, keygen=True, **inputs)
    assert nat["rsa_result"] == pow(x, e, n) and (nat["rsa_rows"], nat["rows"]) == (121579, 125214)
    desc = shapes.maingate_description(True)
    p = pkg.fields.BN254_FR.p
    rowcheck.verify(desc, k, p, nat["fixed"], nat["advice"], nat["mapping"], nat["rows"])
    rowcheck.verify_range_table(nat["fixed"], k)
    cs = plonk.maingate_cs(True)
    assert desc == cs.description()
    asm = plonk.Assembly(6, 1 << k)
    asm.mapping = nat["mapping"].astype(np.int64)
    s = 0x5EED5EED5EED5EED
    srs = PO.setup_srs(po.BN254, k, s, 16)
    key = PO.keygen(po.BN254, srs, desc, k, nat["fixed"], asm.mapping, 16)
    rep = PO.transcript_repr(po.BN254, key, nat["selectors"])
    params = native.ParamsKZG.setup(ctx, pkg.fields.BN254, k, s)
    pk = native.ProvingKey.keygen(ctx, params, cs, nat["fixed"], asm, nat["selectors"])
    assert pk.vk_bytes() == PO.vk_bytes(po.BN254, key, nat["selectors"])
    pk.transcript_repr = rep
    side = pkg.Context(0)
    P = native.Prover(params, pk, ctx, side)
    tr, info = P.create_proof_circuit(native.CIRCUIT_DELAY_ENC, [[]], prover.SeededRng(5), **inputs)
    proof = tr.finalize()
    adv_m = np.stack([co.field_op(0, "to_mont", nat["advice"][i]) for i in range(5)])
    want, _ = PO.create_proof(po.BN254, srs, key, adv_m, [[]], PO.ScalarStream(5), rep, 16)
    assert len(proof) == 2848 and proof == want and info["rows"] == nat["rows"] and info["cipher"] == nat["cipher"]
    assert V.verify_proof(po.BN254, desc, k, key["fixed_commitments"], key["perm_commitments"], rep, (1, 2), pr.G2, pr.g2_mul(s, pr.G2), [[]], proof)
    P.release(); pk.release(); params.release(); side.close()


@pytest.mark.gpu
def test_device_proof_of_a_1024_bit_mod_pow_witness(pkg, po, co, ctx):
    """BASELINE configs[2] as BASELINE.json words it -- a 1024-bit RSA mod-exp circuit (the checked-in bench's constant is 2048: test_device_proof_of_the_real_mod_pow_witness):
    16 limbs, the RangeChip table compute_range_lens(16) gives (5-bit overflow limbs), 5-bit exponent, k = 14: synthesized inside the call, proof bytes equal the CPU
    restatement's, verifier accepts."""
    import random
    import pairing as pr
    import plonk_oracle as PO
    import shapes
    import verifier as V
    from dehalo2_amd import native, plonk, prover

    rnd = random.Random(1024)
    n16, x16, e, k = rnd.getrandbits(1024) | (1 << 1023) | 1, rnd.getrandbits(1000), 0b10111, 14
    inputs = dict(n_big=n16, e=e, x=x16, exp_bits=5, bits_len=1024)
    nat = native.synthesize(native.CIRCUIT_MOD_POW, k, keygen=True, **inputs)
    assert nat["rsa_result"] == pow(x16, e, n16) and nat["rows"] == 15671
    desc = shapes.maingate_description(True)
    cs = plonk.maingate_cs(True)
    asm = plonk.Assembly(6, 1 << k)
    asm.mapping = nat["mapping"].astype(np.int64)
    s = 0x5EED5EED5EED5EED
    srs = PO.setup_srs(po.BN254, k, s, 16)
    key = PO.keygen(po.BN254, srs, desc, k, nat["fixed"], asm.mapping, 16)
    rep = PO.transcript_repr(po.BN254, key, nat["selectors"])
    params = native.ParamsKZG.setup(ctx, pkg.fields.BN254, k, s)
    pk = native.ProvingKey.keygen(ctx, params, cs, nat["fixed"], asm, nat["selectors"])
    pk.transcript_repr = rep
    side = pkg.Context(0)
    P = native.Prover(params, pk, ctx, side)
    tr, info = P.create_proof_circuit(native.CIRCUIT_MOD_POW, [[]], prover.SeededRng(5), **inputs)
    adv_m = np.stack([co.field_op(0, "to_mont", nat["advice"][i]) for i in range(5)])
    want, _ = PO.create_proof(po.BN254, srs, key, adv_m, [[]], PO.ScalarStream(5), rep, 16)
    proof = tr.finalize()
    assert proof == want and info["rows"] == nat["rows"]
    assert V.verify_proof(po.BN254, desc, k, key["fixed_commitments"], key["perm_commitments"], rep, (1, 2), pr.G2, pr.g2_mul(s, pr.G2), [[]], proof)
    P.release(); pk.release(); params.release(); side.close()


def test_mod_pow_rows_satisfy_the_constraint_system(pkg):
    """benches/mod_pow.rs's RSACircuit (RSA region only, :63-110) on a small exponent: rows check out and x^e mod n is right."""
    from dehalo2_amd import witness as W

    v = rsa_vectors()[0]
    n, x = int(v["n"]), int(v["signature"])
    p = pkg.fields.BN254_FR.p
    circ, info = W.mod_pow_witness(p, 14, n, 0b1, x, 1)
    assert info.rsa_result == pow(x, 1, n) and info.total_rows == info.rsa_rows == circ.used_rows
    assert W.check_rows(circ, p) == info.total_rows
    assert len(circ.cs.lookups) == 5 and circ.cs.degree() == 5


@pytest.mark.gpu
def test_device_proof_of_the_real_mod_pow_witness(pkg, po, co, ctx):
    """BASELINE configs[2]: benches/mod_pow.rs -- RSACircuit, BITS_LEN = 2048, EXP_LIMB_BITS = 5 (:47-49), K = 17 (:258),
    create_proof at :201-209.  Real witness -> device proof == the CPU restatement's proof, byte for byte; verifier accepts."""
    import pairing as pr
    import plonk_oracle as PO
    import verifier as V
    from dehalo2_amd import keygen, prover, transcript, witness as W

    v = rsa_vectors()[1]
    n, x = int(v["n"]), int(v["signature"])
    assert n.bit_length() == 2048
    e = 0b10111                                                      # a 5-bit exponent (RandomBits(EXP_LIMB_BITS), :146)
    k = 17
    circ, info = W.mod_pow_witness(pkg.fields.BN254_FR.p, k, n, e, x, 5)
    assert info.rsa_result == pow(x, e, n) and info.total_rows == info.rsa_rows
    assert info.rsa_rows == 41766 + 1                               # benches/README.md:73 (K = 16 there: the published figure is the last used row's index)
    c = _oracle_chain(po, co, circ, k, 16)
    params = keygen.ParamsKZG(ctx, pkg.fields.BN254, k, c["srs"]["g"], c["srs"]["g_lagrange"])
    pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    tr = transcript.Blake2bWrite(pkg.fields.BN254)
    fgp.Prover(params, pk).create_proof(c["adv"], [[]], prover.SeededRng(5), tr)
    want, _ = PO.create_proof(po.BN254, c["srs"], c["key"], c["adv"], [[]], PO.ScalarStream(5), c["rep"], 16)
    proof = tr.finalize()
    assert len(proof) == 2848 and proof == want
    assert V.verify_proof(po.BN254, c["desc"], k, c["key"]["fixed_commitments"], c["key"]["perm_commitments"], c["rep"], (1, 2), pr.G2, c["s_g2"], [[]], proof)
    params.release()
    # the same witness through the whole-call C ABI (dehalo_params_setup -> dehalo_keygen -> dehalo_create_proof, with a side context), and with the circuit
    # synthesized inside the call (dehalo_create_proof_circuit): the same bytes
    from dehalo2_amd import native
    side = pkg.Context(0)
    nparams = native.ParamsKZG.setup(ctx, pkg.fields.BN254, k, c["s"])
    npk = native.ProvingKey.keygen(ctx, nparams, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    npk.transcript_repr = c["rep"]
    nprover = native.Prover(nparams, npk, ctx, side)
    assert nprover.create_proof(c["adv"], [[]], prover.SeededRng(5)).finalize() == want
    tr2, info2 = nprover.create_proof_circuit(native.CIRCUIT_MOD_POW, [[]], prover.SeededRng(5), n_big=n, e=e, x=x, exp_bits=5)
    assert tr2.finalize() == want and info2["rows"] == info.total_rows
    nprover.release(); npk.release(); nparams.release(); side.close()


@pytest.mark.gpu
def test_batch_mode_provers_make_the_proofs_a_lone_prover_makes(pkg, po, co, ctx):
    """BASELINE configs[4] on one GPU: four provers, each on its own context and host thread, sharing the SRS tables, the proving key
    and the compiled programs (bench.py's batch mode).  Every proof must equal, byte for byte, the proof a single prover with a side
    context makes alone from the same seed; one of them is also compared with the CPU restatement's and verified."""
    import threading

    import pairing as pr
    import plonk_oracle as PO
    import verifier as V
    from dehalo2_amd import keygen, prover, transcript, witness as W

    v = rsa_vectors()[0]
    k = 14
    circ, info = W.delay_enc_witness(pkg.fields.BN254_FR.p, k, int(v["n"]), 0b1, int(v["signature"]), 1, [0, 0])
    c = _oracle_chain(po, co, circ, k, 16)
    curve = pkg.fields.BN254
    params = keygen.ParamsKZG(ctx, curve, k, c["srs"]["g"], c["srs"]["g_lagrange"])
    pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
    side = pkg.Context(0)
    lone = fgp.Prover(params, pk, ctx, side)
    seeds = list(range(100, 124))
    alone = {}
    for sd in seeds:
        tr = transcript.Blake2bWrite(curve)
        lone.create_proof(c["adv"], [[]], prover.SeededRng(sd), tr)
        alone[sd] = tr.finalize()
    assert len(set(alone.values())) == len(seeds)
    want, _ = PO.create_proof(po.BN254, c["srs"], c["key"], c["adv"], [[]], PO.ScalarStream(seeds[3]), c["rep"], 16)
    assert alone[seeds[3]] == want
    ctxs = [pkg.Context(0) for _ in range(4)]
    provers = [fgp.Prover(params, pk, ctx=cx) for cx in ctxs]
    got, errors = {}, []

    def work(j):
        try:
            for sd in seeds[j::4]:
                tr = transcript.Blake2bWrite(curve)
                provers[j].create_proof(c["adv"], [[]], prover.SeededRng(sd), tr)
                got[sd] = tr.finalize()
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)

    th = [threading.Thread(target=work, args=(j,)) for j in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    bad = [sd for sd in seeds if got.get(sd) != alone[sd]]
    assert not bad, "batch-mode proofs differ from the lone prover's for seeds %r" % bad
    assert V.verify_proof(po.BN254, c["desc"], k, c["key"]["fixed_commitments"], c["key"]["perm_commitments"], c["rep"], (1, 2), pr.G2, c["s_g2"], [[]], got[seeds[-1]])
    for cx in ctxs:
        cx.close()
    side.close()
    params.release()


# ---------------------------------------------------------------- native witness generation (dehalo_synthesize, csrc/witness.hip)
def _same_circuit(nat, circ, info):
    assert nat["rows"] == info.total_rows == circ.used_rows and nat["rsa_rows"] == info.rsa_rows
    assert nat["rsa_result"] == info.rsa_result and nat["cipher"] == info.cipher
    assert np.array_equal(nat["advice"], circ.advice)
    assert np.array_equal(nat["fixed"], circ.fixed)
    assert np.array_equal(nat["mapping"].astype(np.int64), circ.assembly.mapping)
    assert len(nat["selectors"]) == len(circ.selectors) and all(np.array_equal(a, b) for a, b in zip(nat["selectors"], circ.selectors))


def test_native_synthesize_equals_witness_py_bit_for_bit(pkg):
    """The C++ restatement of the three circuits' synthesize lays out the same rows as witness.py: advice, fixed columns (range table
    included), the permutation assembly and the selectors are identical arrays; x^e mod n and the ciphertext are the same values."""
    import time
    from dehalo2_amd import native, witness as W

    p = pkg.fields.BN254_FR.p
    v = rsa_vectors()
    n, x = int(v[0]["n"]), int(v[0]["signature"])
    # delay_enc, 1-bit exponent, k = 14
    circ, info = W.delay_enc_witness(p, 14, n, 0b1, x, 1, [0, 0])
    _same_circuit(native.synthesize(native.CIRCUIT_DELAY_ENC, 14, n_big=n, e=1, x=x, exp_bits=1, message=[0, 0], keygen=True), circ, info)
    # the reference's checked-in parameters: 5-bit exponent, k = 16 (src/lib.rs:122-124); other modulus, other message
    n2, x2 = int(v[1]["n"]), int(v[1]["signature"])
    circ, info = W.delay_enc_witness(p, 16, n2, 0b10011, x2, 5, [0, 0])
    t = time.perf_counter()
    nat = native.synthesize(native.CIRCUIT_DELAY_ENC, 16, n_big=n2, e=0b10011, x=x2, exp_bits=5, message=[0, 0], keygen=True)
    assert time.perf_counter() - t < 2.0
    _same_circuit(nat, circ, info)
    assert nat["rsa_result"] == pow(x2, 0b10011, n2)
    # mod_pow (RSA region only) and an 8-bit exponent (range-assigned exponent cell)
    circ, info = W.mod_pow_witness(p, 15, n, 0b101, x, 3)
    _same_circuit(native.synthesize(native.CIRCUIT_MOD_POW, 15, n_big=n, e=0b101, x=x, exp_bits=3, keygen=True), circ, info)
    circ, info = W.mod_pow_witness(p, 17, n2, 0xA7, x2, 8)
    _same_circuit(native.synthesize(native.CIRCUIT_MOD_POW, 17, n_big=n2, e=0xA7, x=x2, exp_bits=8, keygen=True), circ, info)
    # pose_enc, K = 11 (MainGate only: 9 fixed columns, no selectors)
    circ, info = W.pose_enc_witness(p, 11, [0xABCDEF, p - 1], [0, 0])
    _same_circuit(native.synthesize(native.CIRCUIT_POSE_ENC, 11, key=[0xABCDEF, p - 1], message=[0, 0], keygen=True), circ, info)
    # advice alone (what a proof needs) is the same array; a circuit that does not fit is refused
    adv_only = native.synthesize(native.CIRCUIT_DELAY_ENC, 14, n_big=n, e=1, x=x, exp_bits=1, message=[0, 0])
    assert np.array_equal(adv_only["advice"], W.delay_enc_witness(p, 14, n, 0b1, x, 1, [0, 0])[0].advice)
    # (the proving call writes the multiplication rows by a word-arithmetic fast path, the keygen call by the general one: the same rows, and zeros behind them)
    fast = native.synthesize(native.CIRCUIT_DELAY_ENC, 16, n_big=n2, e=0b10011, x=x2, exp_bits=5, message=[0, 0])
    assert np.array_equal(fast["advice"], nat["advice"]) and fast["rsa_result"] == nat["rsa_result"] == pow(x2, 0b10011, n2) and fast["cipher"] == nat["cipher"]
    with pytest.raises(ValueError):
        native.synthesize(native.CIRCUIT_DELAY_ENC, 13, n_big=n, e=1, x=x, exp_bits=1, message=[0, 0])


def test_native_synthesize_is_fast_at_the_north_star_size(pkg):
    """k = 17, 2048-bit modulus, 15-bit exponent (benches/README.md:59-60): the advice columns of one proof."""
    import random
    import time
    from dehalo2_amd import native

    rnd = random.Random(5)
    n_big, x, e = rnd.getrandbits(2048) | (1 << 2047) | 1, rnd.getrandbits(2040), rnd.getrandbits(15) | (1 << 14)
    best = None
    for _ in range(3):
        t = time.perf_counter()
        nat = native.synthesize(native.CIRCUIT_DELAY_ENC, 17, n_big=n_big, e=e, x=x, exp_bits=15, message=[0, 0])
        el = time.perf_counter() - t
        best = el if best is None or el < best else best
    # the RSA region is the published mod_pow figure (benches/README.md:84: 121,578 = the last row's index); hash and cipher regions of the checked-in source behind it
    assert nat["rsa_result"] == pow(x, e, n_big) and nat["rsa_rows"] == 121578 + 1 and nat["rows"] == 121579 + 2182 + 1453 == 125214
    print("native synthesize, k = 17, 15-bit exponent: %.1f ms for %d rows" % (1e3 * best, nat["rows"]))
    assert best < 0.5


def test_native_synthesize_rows_do_not_depend_on_the_number_of_threads(pkg):
    """dehalo_synthesize writes the RSA regions of a proving call from several host threads (BigIntChip::pow_mod_threads), each through a cursor at the row
    the sequential order gives its region: k = 17, 15-bit exponent, 1 / 3 / 8 threads -- the same advice columns (digest of all 5 x 2^17 values)."""
    import subprocess
    import sys
    from conftest import ROOT

    seen = set()
    for threads in ("1", "3", "8"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "synth_bench.py")], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, DEHALO_SYNTH_THREADS=threads))
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.strip().splitlines()
        assert lines[-2].endswith("True")                      # x^e mod n read off the rows
        seen.add(lines[-1])
    assert len(seen) == 1


def test_synthesis_threads_under_thread_sanitizer(pkg):
    """The same host code built with -fsanitize=thread: a proving call at k = 16 with a 5-bit exponent (eight multiplication regions and the hash / cipher
    task on eight threads) reports no data race and writes the keygen call's advice columns."""
    import subprocess
    from conftest import ROOT

    out = subprocess.run(["make", "-C", ROOT, "host_tsan"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    run = subprocess.run([os.path.join(ROOT, "tests", "native_host", "host_tsan"), "16", "5"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, DEHALO_SYNTH_THREADS="8"))
    assert run.returncode == 0 and "ThreadSanitizer" not in run.stderr and "rows" in run.stdout, run.stdout + run.stderr


def test_native_synthesize_in_a_forked_child(pkg):
    """The synthesis worker threads sleep between calls and do not exist in a forked child: the child makes its own and writes the same rows."""
    from dehalo2_amd import native

    v = rsa_vectors()[1]
    n, x = int(v["n"]), int(v["signature"])
    a = native.synthesize(native.CIRCUIT_MOD_POW, 17, n_big=n, e=0b10111, x=x, exp_bits=5)
    pid = os.fork()
    if pid == 0:
        code = 3
        try:
            b = native.synthesize(native.CIRCUIT_MOD_POW, 17, n_big=n, e=0b10111, x=x, exp_bits=5)
            code = 0 if np.array_equal(a["advice"], b["advice"]) else 4
        finally:
            os._exit(code)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


def test_host_code_under_address_and_ub_sanitizers(pkg):
    """The library's host-only code -- dehalo_synthesize (big-integer division, layouter, Grain / Poseidon), hostfield, Blake2b, the random-scalar
    sources -- rebuilt by g++ with -fsanitize=address,undefined (GPU sanitizers are not available on the pool) and run on a k = 15 delay_enc circuit:
    no report, and the digest of everything it computed equals the digest of the regular build's output for the same inputs."""
    import ctypes as C
    import hashlib
    import subprocess
    from conftest import ROOT
    from dehalo2_amd import native
    from dehalo2_amd._lib import CRng, load_library

    out = subprocess.run(["make", "-C", ROOT, "host_sanitize"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    run = subprocess.run([os.path.join(ROOT, "tests", "native_host", "host_sanitize"), "15", "3"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stdout + run.stderr
    digest, _, rows = run.stdout.split()
    # the same inputs through the regular library
    s, M = 0x9E3779B97F4A7C15, (1 << 64) - 1
    def nxt():
        nonlocal s
        s ^= (s << 13) & M; s ^= s >> 7; s ^= (s << 17) & M
        return s
    nl = [nxt() for _ in range(32)]
    xl = [nxt() for _ in range(32)]
    nl[31] |= 1 << 63; nl[0] |= 1; xl[31] >>= 8
    n_big, x_big = sum(v << (64 * i) for i, v in enumerate(nl)), sum(v << (64 * i) for i, v in enumerate(xl))
    nat = native.synthesize(native.CIRCUIT_DELAY_ENC, 15, n_big=n_big, e=(1 << 2) | 1, x=x_big, exp_bits=3, message=[0, 0], keygen=True)
    assert nat["rows"] == int(rows) and nat["rsa_result"] == pow(x_big, 5, n_big)
    r = CRng()
    r.kind, r.pcg_state[0], r.pcg_inc[0] = 1, 123, 457
    a4 = np.zeros(4, dtype=np.uint64)
    assert load_library().dehalo_rng_scalars(C.byref(r), 0, 5000, a4.ctypes.data, 1) == 0
    h = hashlib.blake2b(digest_size=32, person=b"host-sanitize-ru")
    for part in (nat["advice"], nat["fixed"], nat["mapping"], nat["selectors"][0].astype(np.uint8), nat["selectors"][1].astype(np.uint8), a4):
        h.update(np.ascontiguousarray(part).tobytes())
    assert h.hexdigest() == digest


def test_row_counts_match_reference_readme(pkg, oracles):
    """The rows a circuit takes are a property of the layouter: the reference's come from halo2wrong's MainGate / RangeChip and are published in
    benches/README.md:56-99 (32 (configuration -> advice rows) pairs; held as data by the checker, oracle/rowcheck.py).  witness.py / csrc/witness.hip lay every
    instruction out the way halo2wrong does, and reproduce them:
      * mod_pow, 13 of 13: published = rows - 1 (the published figure is the index of the last used row: the count less one);
      * pose_enc, 10 of 11 by the same rule (718 rows per permutation, 4 per message word); the 11th (|msg| = 17: 4,394) repeats the figure of |msg| = 20
        in the table itself -- 17 words take 4,382;
      * delay_enc, 8 of 8 up to ONE constant: published = rows - 1 + 5,035.  The RSA region is mod_pow's exactly (row 1 with e = 0, whose result has one
        limb; row 4's "7-bit" is the 14-bit figure: 122,267 needs k = 17); the constant is the hash region of an earlier revision of src/lib.rs -- the
        checked-in one packs the 32 limbs three to a field element (src/lib.rs:222-249: 11 inputs, 3 permutations, 2,182 rows), the published table
        was made when that region took 7,217 rows (ten permutations' worth).  DESIGN.md section 5 carries the table."""
    import random
    import rowcheck
    from dehalo2_amd import native, witness as W
    p = pkg.fields.BN254_FR.p
    rnd = random.Random(1)
    n_big, x = rnd.getrandbits(2048) | (1 << 2047) | 1, rnd.getrandbits(2040)
    assert len(rowcheck.README_MOD_POW) + len(rowcheck.README_DELAY_ENC) + len(rowcheck.README_POSE_ENC) == 32
    for k, published, bits in rowcheck.README_MOD_POW:
        e = (1 << (bits - 1)) | 1
        nat = native.synthesize(native.CIRCUIT_MOD_POW, k, n_big=n_big, e=e, x=x, exp_bits=bits)
        assert nat["rows"] == W.rsa_region_rows(bits) == published + 1, (k, bits)
        assert nat["rsa_result"] == pow(x, e, n_big) and nat["rsa_result"] >> 1984            # a 32-limb result, as a random one is
    # the layouter itself (Python) on the small ones, and the closed form against it
    for bits in (1, 2):
        _, info = W.mod_pow_witness(p, 15, n_big, (1 << (bits - 1)) | 1, x, bits)
        assert info.total_rows == W.rsa_region_rows(bits)
    hash_and_cipher = W.hash_region_rows() + W.cipher_region_rows(2, True)
    assert hash_and_cipher == 2182 + 1453
    residuals = set()
    for k, published, bits_printed, msg in rowcheck.README_DELAY_ENC:
        bits = 14 if (k, bits_printed) == (17, 7) else bits_printed
        e = 0 if bits == 2 else (1 << (bits - 1)) | 1                  # the published 2-bit run drew e = 0: x^0 = 1 is a one-limb constant, 31 rows fewer
        nat = native.synthesize(native.CIRCUIT_DELAY_ENC, k, n_big=n_big, e=e, x=x, exp_bits=bits, message=[0] * msg)
        assert nat["rsa_rows"] == W.rsa_region_rows(bits, 1 if e == 0 else 32) and nat["rows"] == nat["rsa_rows"] + hash_and_cipher
        residuals.add(published - (nat["rows"] - 1))
    assert residuals == {5035} and 5035 == 7217 - 2182                 # one constant for all eight: the earlier revision's hash region
    spec = W.poseidon_spec(p)
    for k, published, msg in rowcheck.README_POSE_ENC:
        lay = W.Layouter(p)
        W.cipher_region(lay, spec, [5, 6], [0] * msg)
        assert lay.rows == W.cipher_region_rows(msg, False) and lay.rows + 1 <= (1 << k) - 6
        assert lay.rows == (4382 if msg == 17 else published) + 1, msg


def test_synthesized_circuits_pass_the_checkers_own_row_check(pkg, oracles):
    """What dehalo_synthesize (C++) lays out -- advice, fixed columns, permutation -- checked by oracle/rowcheck.py, a MockProver written from
    oracle/shapes.py alone: every gate row, every lookup input, every copy constraint, the range table; and a broken cell is caught."""
    import rowcheck
    import shapes
    from dehalo2_amd import native
    p = pkg.fields.BN254_FR.p
    v = rsa_vectors()
    n, x = int(v[0]["n"]), int(v[0]["signature"])
    nat = native.synthesize(native.CIRCUIT_DELAY_ENC, 15, n_big=n, e=0b10, x=x, exp_bits=2, message=[0, 0], keygen=True)
    desc = shapes.maingate_description(True)
    got = rowcheck.verify(desc, 15, p, nat["fixed"], nat["advice"], nat["mapping"], nat["rows"])
    assert got["rows"] == (1 << 15) - 6 and got["cells_in_cycles"] > nat["rows"]
    rowcheck.verify_range_table(nat["fixed"], 15)
    assert nat["rsa_result"] == pow(x, 2, n)
    row = 1000 + int(np.nonzero(nat["fixed"][shapes.SD, 1000:].any(axis=1))[0][0])            # a row whose column d enters the gate
    bad = nat["advice"].copy()
    bad[3, row, 0] ^= np.uint64(1)
    with pytest.raises(AssertionError):
        rowcheck.verify(desc, 15, p, nat["fixed"], bad, nat["mapping"], nat["rows"])
    nat = native.synthesize(native.CIRCUIT_MOD_POW, 14, n_big=n, e=1, x=x, exp_bits=1, keygen=True)
    rowcheck.verify(desc, 14, p, nat["fixed"], nat["advice"], nat["mapping"], nat["rows"])
    nat = native.synthesize(native.CIRCUIT_POSE_ENC, 11, key=[5, 6], message=[0, 0], keygen=True)
    rowcheck.verify(shapes.maingate_description(False), 11, p, nat["fixed"], nat["advice"], nat["mapping"], nat["rows"])
    # BASELINE configs[2] names a 1024-bit modulus (the checked-in bench has BITS_LEN = 2048): 16 limbs configure another RangeChip table (5-bit overflow limbs
    # instead of 6: compute_range_lens) -- same layouter, C++ == Python bit for bit, the checker's MockProver and its own statement of that table agree
    import random
    from dehalo2_amd import witness as W
    rnd = random.Random(1024)
    n16, x16 = rnd.getrandbits(1024) | (1 << 1023) | 1, rnd.getrandbits(1000)
    nat = native.synthesize(native.CIRCUIT_MOD_POW, 14, n_big=n16, e=0b11, x=x16, exp_bits=2, bits_len=1024, keygen=True)
    circ, info = W.mod_pow_witness(p, 14, n16, 0b11, x16, 2, num_limbs=16)
    _same_circuit(nat, circ, info)
    assert nat["rows"] == W.rsa_region_rows(2, 16, 16) and nat["rsa_result"] == pow(x16, 3, n16)
    rowcheck.verify(desc, 14, p, nat["fixed"], nat["advice"], nat["mapping"], nat["rows"])
    rowcheck.verify_range_table(nat["fixed"], 14, 16)
    assert [b for b in sorted({t for t, _ in rowcheck.expected_range_table(16)})] == [0, 1, 2, 3, 4] and len(rowcheck.expected_range_table(16)) == 1 + 2 + 16 + 32 + 256
    # what the reference's circuit cannot satisfy is refused: x >= n (assert_in_field), an exponent wider than exp_bits (to_bits), a non-zero message
    for bad_inputs in (dict(x=n + 5), dict(e=7), dict(message=[1, 2])):
        kw = dict(n_big=n, e=1, x=x, exp_bits=1, message=[0, 0])
        kw.update(bad_inputs)
        with pytest.raises(ValueError):
            native.synthesize(native.CIRCUIT_DELAY_ENC, 14, **kw)
