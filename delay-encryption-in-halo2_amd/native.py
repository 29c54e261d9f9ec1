"""The whole call through the C ABI: ParamsKZG, keygen, ProvingKey / VerifyingKey, Blake2bWrite and `create_proof` as ONE
library call each (include/dehalo.h, "the whole call"; csrc/prover.hip) -- the objects and the call of the reference's benches
(benches/delay_enc.rs:41-54 params, :84-115 keys, :120-134 create_proof into a Blake2bWrite transcript).  Python is a caller
here: it serialises the ConstraintSystem into the C descriptor and passes pointers; phases, transcript hashing and every
launch run in C++.  (tests/fine_grained_prover.py -- test infrastructure -- drives the same proof from Python through the fine-grained entry points.)"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import plonk
from ._lib import (CCircuitInputs, CColumnQuery, CConstraintSystem, CExprNode, CRng, CSynthesisInfo, Context, DehaloError, RNG_FILL_FN, load_library)
from .fields import CurveSpec, FieldSpec

EXPR_KIND = {"const": 0, plonk.FIXED: 1, plonk.ADVICE: 2, plonk.INSTANCE: 3, "neg": 4, "sum": 5, "product": 6, "scaled": 7}
COLUMN_KIND = {plonk.ADVICE: 0, plonk.FIXED: 1, plonk.INSTANCE: 2}
RNG_OS, RNG_PCG64, RNG_CALLBACK = 0, 1, 2
KEYGEN_FIXED_CANONICAL = 1
PROOF_ADVICE_ON_DEVICE, PROOF_ADVICE_CANONICAL = 1, 2
CIRCUIT_DELAY_ENC, CIRCUIT_MOD_POW, CIRCUIT_POSE_ENC = 0, 1, 2
PHASES = ("advice", "lookups", "products", "random", "quotient", "evaluations", "openings", "total")


class ConstraintSystemDescriptor:
    """dehalo_constraint_system built from a plonk.ConstraintSystem; owns the arrays the C struct points into."""

    def __init__(self, cs: plonk.ConstraintSystem, field: FieldSpec):
        nodes: List[tuple] = []
        consts: List[int] = []
        memo: Dict[tuple, int] = {}

        def const(c: int) -> int:
            c %= field.p
            if c not in consts:
                consts.append(c)
            return consts.index(c)

        def emit(e) -> int:
            if e in memo:
                return memo[e]
            k = e[0]
            if k == "const":
                node = (0, const(e[1]), 0, 0)
            elif k in (plonk.FIXED, plonk.ADVICE, plonk.INSTANCE):
                node = (EXPR_KIND[k], e[1], 0, e[2])
            elif k == "neg":
                node = (4, emit(e[1]), 0, 0)
            elif k in ("sum", "product"):
                a = emit(e[1])
                node = (EXPR_KIND[k], a, emit(e[2]), 0)
            elif k == "scaled":
                node = (7, emit(e[1]), const(e[2]), 0)
            else:
                raise ValueError("unknown expression node %r" % (k,))
            nodes.append(node)
            memo[e] = len(nodes) - 1
            return memo[e]

        gates = [emit(g) for g in cs.gates]
        lens, lin, ltab = [], [], []
        for inputs, tables in cs.lookups:
            if len(inputs) != len(tables):
                raise ValueError("lookup: input and table expression counts differ")
            lens.append(len(inputs))
            lin += [emit(e) for e in inputs]
            ltab += [emit(e) for e in tables]
        self._nodes = (CExprNode * max(1, len(nodes)))(*[CExprNode(*nd) for nd in nodes])
        self._consts = field.encode_many(consts) if consts else np.zeros((1, 4), dtype=np.uint64)
        u32 = lambda xs: (C.c_uint32 * max(1, len(xs)))(*xs)
        self._gates, self._lens, self._lin, self._ltab = u32(gates), u32(lens), u32(lin), u32(ltab)
        self._perm = (CColumnQuery * max(1, len(cs.permutation_columns)))(*[CColumnQuery(COLUMN_KIND[k], i, 0) for k, i in cs.permutation_columns])
        mk = lambda lst, kind: (CColumnQuery * max(1, len(lst)))(*[CColumnQuery(COLUMN_KIND[kind], c, r) for c, r in lst])
        self._aq, self._fq, self._iq = mk(cs.advice_queries, plonk.ADVICE), mk(cs.fixed_queries, plonk.FIXED), mk(cs.instance_queries, plonk.INSTANCE)
        d = CConstraintSystem()
        d.num_advice, d.num_fixed, d.num_instance, d.minimum_degree = cs.num_advice, cs.num_fixed, cs.num_instance, cs.minimum_degree
        d.nodes, d.num_nodes = self._nodes, len(nodes)
        d.constants, d.num_constants = self._consts.ctypes.data, len(consts)
        d.gates, d.num_gates = self._gates, len(gates)
        d.lookup_lens, d.num_lookups, d.lookup_inputs, d.lookup_tables = self._lens, len(lens), self._lin, self._ltab
        d.permutation_columns, d.num_permutation_columns = self._perm, len(cs.permutation_columns)
        d.advice_queries, d.num_advice_queries = self._aq, len(cs.advice_queries)
        d.fixed_queries, d.num_fixed_queries = self._fq, len(cs.fixed_queries)
        d.instance_queries, d.num_instance_queries = self._iq, len(cs.instance_queries)
        self.struct = d

    def ref(self):
        return C.byref(self.struct)


def _check(ctx: Context, rc: int):
    if rc != 0:
        raise DehaloError(rc, load_library().dehalo_last_error(ctx.handle).decode())


def _bytes_ptr(b):
    return C.cast(C.c_char_p(bytes(b)), C.c_void_p)


class ParamsKZG:
    """dehalo_params: ParamsKZG<Bn256> resident on the device."""

    def __init__(self, ctx: Context, curve: CurveSpec, handle):
        self.ctx, self.curve, self.handle = ctx, curve, handle

    @classmethod
    def create(cls, ctx: Context, curve: CurveSpec, k: int, g, g_lagrange, g2: bytes = b"", s_g2: bytes = b"") -> "ParamsKZG":
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
        gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
        if g.shape[0] != 1 << k or gl.shape[0] != 1 << k:
            raise ValueError("g and g_lagrange must hold 2^k points")
        h = C.c_void_p()
        b2, bs = bytes(g2).ljust(128, b"\0"), bytes(s_g2).ljust(128, b"\0")
        _check(ctx, load_library().dehalo_params_create(ctx.handle, curve.id, k, g.ctypes.data, gl.ctypes.data, _bytes_ptr(b2), _bytes_ptr(bs), C.byref(h)))
        return cls(ctx, curve, h)

    @classmethod
    def setup(cls, ctx: Context, curve: CurveSpec, k: int, s: Optional[int] = None) -> "ParamsKZG":
        """ParamsKZG::setup(k, rng) (benches/delay_enc.rs:43): the SRS for the secret `s` (a field element; None = drawn from OS entropy, as the reference's OsRng
        does), generated on the device -- dehalo_params_setup."""
        if s is None:
            import secrets
            s = secrets.randbelow(curve.scalar.p - 2) + 2
        h = C.c_void_p()
        sm = curve.scalar.encode(s % curve.scalar.p)
        _check(ctx, load_library().dehalo_params_setup(ctx.handle, curve.id, k, sm.ctypes.data, C.byref(h)))
        return cls(ctx, curve, h)

    @classmethod
    def read(cls, ctx: Context, curve: CurveSpec, data: bytes) -> "ParamsKZG":
        h = C.c_void_p()
        buf = np.frombuffer(data, dtype=np.uint8)
        _check(ctx, load_library().dehalo_params_read(ctx.handle, curve.id, buf.ctypes.data, len(data), C.byref(h)))
        return cls(ctx, curve, h)

    def write(self) -> bytes:
        lib = load_library()
        out = np.empty(lib.dehalo_params_size(self.handle), dtype=np.uint8)
        _check(self.ctx, lib.dehalo_params_write(self.handle, out.ctypes.data, out.size))
        return out.tobytes()

    def release(self):
        if self.handle is not None:
            load_library().dehalo_params_release(self.ctx.handle, self.handle)
            self.handle = None


class ProvingKey:
    """dehalo_pk: ProvingKey (with its VerifyingKey and compiled programs) resident on the device."""

    def __init__(self, ctx: Context, curve: CurveSpec, cs: plonk.ConstraintSystem, handle):
        self.ctx, self.curve, self.cs, self.handle = ctx, curve, cs, handle

    @classmethod
    def keygen(cls, ctx: Context, params: ParamsKZG, cs: plonk.ConstraintSystem, fixed_canonical, assembly: plonk.Assembly, selectors: Sequence = ()) -> "ProvingKey":
        """keygen_vk + keygen_pk (benches/delay_enc.rs:86,103): fixed_canonical = (num_fixed, n, 4) u64 canonical values."""
        f = params.curve.scalar
        desc = ConstraintSystemDescriptor(cs, f)
        fixed = np.ascontiguousarray(fixed_canonical, dtype=np.uint64)
        mapping = np.ascontiguousarray(assembly.mapping, dtype=np.uint64)
        sels = [np.ascontiguousarray(np.asarray(s, dtype=np.uint8)) for s in selectors]
        sel_ptrs = (C.c_void_p * max(1, len(sels)))(*[s.ctypes.data for s in sels])
        h = C.c_void_p()
        _check(ctx, load_library().dehalo_keygen(ctx.handle, params.handle, desc.ref(), fixed.ctypes.data, mapping.ctypes.data, sel_ptrs, len(sels), KEYGEN_FIXED_CANONICAL,
                                                 C.byref(h)))
        return cls(ctx, params.curve, cs, h)

    @classmethod
    def read(cls, ctx: Context, curve: CurveSpec, cs: plonk.ConstraintSystem, data: bytes, num_selectors: int = 0) -> "ProvingKey":
        desc = ConstraintSystemDescriptor(cs, curve.scalar)
        buf = np.frombuffer(data, dtype=np.uint8)
        h = C.c_void_p()
        _check(ctx, load_library().dehalo_pk_read(ctx.handle, curve.id, desc.ref(), buf.ctypes.data, len(data), num_selectors, C.byref(h)))
        return cls(ctx, curve, cs, h)

    def write(self) -> bytes:
        lib = load_library()
        out = np.empty(lib.dehalo_pk_size(self.handle), dtype=np.uint8)
        _check(self.ctx, lib.dehalo_pk_write(self.ctx.handle, self.handle, out.ctypes.data, out.size))
        return out.tobytes()

    def vk_bytes(self) -> bytes:
        lib = load_library()
        out = np.empty(lib.dehalo_vk_size(self.handle), dtype=np.uint8)
        _check(self.ctx, lib.dehalo_vk_write(self.handle, out.ctypes.data, out.size))
        return out.tobytes()

    @property
    def transcript_repr(self) -> int:
        out = np.zeros(4, dtype=np.uint64)
        _check(self.ctx, load_library().dehalo_pk_get_transcript_repr(self.handle, out.ctypes.data))
        return self.curve.scalar.decode(out)

    @transcript_repr.setter
    def transcript_repr(self, value: int):
        _check(self.ctx, load_library().dehalo_pk_set_transcript_repr(self.handle, self.curve.scalar.encode(value).ctypes.data))

    def info(self) -> dict:
        out = (C.c_uint32 * 8)()
        _check(self.ctx, load_library().dehalo_pk_info(self.handle, out))
        return dict(zip(("k", "extended_k", "blinding_factors", "degree", "permutation_sets", "commitments_before_evaluations", "evaluations", "opening_points"), out))

    def release(self):
        if self.handle is not None:
            load_library().dehalo_pk_release(self.ctx.handle, self.handle)
            self.handle = None


class Blake2bWrite:
    """dehalo_transcript: Blake2bWrite<Vec<u8>, C, Challenge255<C>> held by the library."""

    def __init__(self, curve: CurveSpec):
        self.curve = curve
        self.handle = C.c_void_p()
        rc = load_library().dehalo_transcript_create(curve.id, C.byref(self.handle))
        if rc:
            raise DehaloError(rc, "transcript_create")

    def _s(self, v: int):
        return self.curve.scalar.encode(v).ctypes.data

    def common_scalar(self, s: int):
        load_library().dehalo_transcript_common_scalar(self.handle, self._s(s))

    def write_scalar(self, s: int):
        load_library().dehalo_transcript_write_scalar(self.handle, self._s(s))

    def write_point(self, P):
        if P is None:
            raise ValueError("cannot write points at infinity to the transcript")
        xy = np.concatenate([self.curve.base.encode(P[0]), self.curve.base.encode(P[1])])
        if load_library().dehalo_transcript_write_point(self.handle, xy.ctypes.data):
            raise ValueError("cannot write points at infinity to the transcript")

    def squeeze_challenge_scalar(self) -> int:
        out = np.zeros(4, dtype=np.uint64)
        load_library().dehalo_transcript_squeeze_challenge(self.handle, out.ctypes.data)
        return self.curve.scalar.decode(out)

    def finalize(self) -> bytes:
        lib = load_library()
        out = np.empty(max(1, lib.dehalo_transcript_len(self.handle)), dtype=np.uint8)
        n = lib.dehalo_transcript_len(self.handle)
        if lib.dehalo_transcript_finalize(self.handle, out.ctypes.data, out.size):
            raise DehaloError(-1, "transcript_finalize")
        return out[:n].tobytes()

    def __del__(self):
        try:
            if self.handle:
                load_library().dehalo_transcript_release(self.handle)
                self.handle = None
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass


def rng_struct(rng) -> Optional[CRng]:
    """None / prover.OsRng -> NULL (operating-system entropy inside the library); prover.SeededRng -> its PCG64 state; any other object
    with .scalars(count) -> a callback."""
    if rng is None or type(rng).__name__ == "OsRng":
        return None
    r = CRng()
    gen = getattr(rng, "gen", None)
    if gen is not None and type(gen.bit_generator).__name__ == "PCG64":
        st = gen.bit_generator.state["state"]
        r.kind = RNG_PCG64
        r.pcg_state[0], r.pcg_state[1] = st["state"] & ((1 << 64) - 1), st["state"] >> 64
        r.pcg_inc[0], r.pcg_inc[1] = st["inc"] & ((1 << 64) - 1), st["inc"] >> 64
        return r
    r.kind = RNG_CALLBACK

    def fill(_user, out, count, _position):
        a = np.ascontiguousarray(rng.scalars(count), dtype=np.uint64).reshape(-1)
        C.memmove(out, a.ctypes.data, 32 * count)
        return 0
    r._keep = RNG_FILL_FN(fill)
    r.fill = r._keep
    return r


def rng_writeback(rng, r: Optional[CRng]):
    """The library advanced the PCG64 state past the proof's draws (upstream's `&mut rng`): mirror it in the Python generator."""
    if r is not None and r.kind == RNG_PCG64:
        st = rng.gen.bit_generator.state
        st["state"]["state"] = int(r.pcg_state[0]) | (int(r.pcg_state[1]) << 64)
        rng.gen.bit_generator.state = st


class Prover:
    """dehalo_prover: the device buffers of one proof in flight; create_proof is ONE library call."""

    def __init__(self, params: ParamsKZG, pk: ProvingKey, ctx: Optional[Context] = None, side_ctx: Optional[Context] = None):
        self.params, self.pk = params, pk
        self.ctx = ctx if ctx is not None else pk.ctx
        self.side = side_ctx
        self.handle = C.c_void_p()
        _check(self.ctx, load_library().dehalo_prover_create(self.ctx.handle, side_ctx.handle if side_ctx is not None else None, params.handle, pk.handle, C.byref(self.handle)))

    def set_shard(self, rank: int, world: int, gather=None):
        """dehalo_prover_set_shard: this process runs the MSMs of its share of every multi-column commitment phase only; `gather(points, first, num)` --
        points a (count, 8) uint64 array whose rows [first[rank], first[rank] + num[rank]) are filled -- must fill in the other rows in place (an all-gather:
        sharding.gather_points).  Every process must prove with the same inputs and the same seeded rng.  world = 1 switches it off."""
        from ._lib import GATHER_FN

        def thunk(_user, pts, count, first, num, w):
            try:
                arr = np.ctypeslib.as_array(pts, shape=(count, 8))
                gather(arr, [first[i] for i in range(w)], [num[i] for i in range(w)])
                return 0
            except Exception:      # noqa: BLE001  (an exception must not unwind through the C frames)
                import traceback
                traceback.print_exc()
                return -1

        self._gather_cb = GATHER_FN(thunk) if world > 1 else GATHER_FN()      # kept alive as long as the prover may call it
        _check(self.ctx, load_library().dehalo_prover_set_shard(self.handle, rank, world, self._gather_cb, None))

    def create_proof(self, advice, instances: Sequence[Sequence[int]] = ((),), rng=None, transcript: Optional[Blake2bWrite] = None, canonical: bool = False) -> Blake2bWrite:
        """advice: (num_advice, n, 4) u64 Montgomery (or plain integers < p with canonical=True: what native.synthesize returns) -- a host
        array or a device tensor (anything with .data_ptr()).  instances: one list of canonical ints per instance column.  rng: None = OS
        entropy; prover.SeededRng for reproducible test proofs."""
        lib = load_library()
        tr = transcript if transcript is not None else Blake2bWrite(self.pk.curve)
        f = self.pk.curve.scalar
        flags = PROOF_ADVICE_CANONICAL if canonical else 0
        if hasattr(advice, "data_ptr"):
            adv_ptr, flags = advice.data_ptr(), flags | PROOF_ADVICE_ON_DEVICE
            if hasattr(advice, "is_cuda") and advice.is_cuda:
                import torch
                torch.cuda.current_stream().synchronize()      # the library reads the tensor on its own stream
        else:
            adv = np.ascontiguousarray(advice, dtype=np.uint64)
            adv_ptr = adv.ctypes.data
        cols = [f.encode_many(list(v)) if len(v) else np.zeros((0, 4), dtype=np.uint64) for v in instances]
        ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data if c.size else None for c in cols])
        lens = (C.c_size_t * max(1, len(cols)))(*[c.shape[0] for c in cols])
        r = rng_struct(rng)
        rc = lib.dehalo_create_proof(self.handle, adv_ptr, ptrs, lens, len(cols), C.byref(r) if r is not None else None, tr.handle, flags)
        if rc != 0:
            msg = lib.dehalo_last_error(self.ctx.handle).decode()
            if rc == -1 and ("instance" in msg or "infinity" in msg):
                raise ValueError(msg)
            raise DehaloError(rc, msg)
        rng_writeback(rng, r)
        return tr

    def create_proof_circuit(self, circuit: int, instances: Sequence[Sequence[int]] = ((),), rng=None, transcript: Optional[Blake2bWrite] = None, **inputs):
        """create_proof(&params, &pk, &[circuit], instances, rng, &mut transcript) as the reference calls it (benches/delay_enc.rs:123-131): the circuit is
        synthesized inside the call (dehalo_create_proof_circuit).  `inputs`: n_big, e, x, exp_bits, message, key, bits_len as for synthesize().
        -> (transcript, {"rows", "rsa_rows", "rsa_result", "cipher"})"""
        lib = load_library()
        tr = transcript if transcript is not None else Blake2bWrite(self.pk.curve)
        f = self.pk.curve.scalar
        inp, keep = circuit_inputs(circuit, self.pk.info()["k"], **inputs)
        cols = [f.encode_many(list(v)) if len(v) else np.zeros((0, 4), dtype=np.uint64) for v in instances]
        ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data if c.size else None for c in cols])
        lens = (C.c_size_t * max(1, len(cols)))(*[c.shape[0] for c in cols])
        r = rng_struct(rng)
        info = CSynthesisInfo()
        rc = lib.dehalo_create_proof_circuit(self.handle, C.byref(inp), C.byref(info), ptrs, lens, len(cols), C.byref(r) if r is not None else None, tr.handle)
        if rc != 0:
            raise DehaloError(rc, lib.dehalo_last_error(self.ctx.handle).decode())
        rng_writeback(rng, r)
        return tr, _info_dict(info)

    def last_timings(self) -> dict:
        out = (C.c_double * 8)()
        load_library().dehalo_prover_last_timings(self.handle, out)
        return dict(zip(PHASES, out))

    def release(self):
        if self.handle:
            load_library().dehalo_prover_release(self.handle)
            self.handle = C.c_void_p()


def create_proofs(provers: Sequence[Prover], advice, rngs: Sequence, count: Optional[int] = None) -> List[bytes]:
    """Batch / throughput mode (BASELINE configs[4]): proof i on prover i mod len(provers), one library thread per prover, no interpreter
    in the loop.  `advice`: one device tensor / host array used for every proof, or a list of them."""
    lib = load_library()
    count = len(rngs) if count is None else count
    advs = list(advice) if isinstance(advice, (list, tuple)) else [advice] * count
    on_device = hasattr(advs[0], "data_ptr")
    if on_device:
        import torch
        torch.cuda.current_stream().synchronize()
        adv_ptrs = [a.data_ptr() for a in advs]
    else:
        keep = [np.ascontiguousarray(a, dtype=np.uint64) for a in advs]
        adv_ptrs = [a.ctypes.data for a in keep]
    info = provers[0].pk.info()
    cap = 32 * (info["commitments_before_evaluations"] + info["evaluations"] + info["opening_points"])
    bufs = [np.empty(cap, dtype=np.uint8) for _ in range(count)]
    lens = (C.c_size_t * max(1, count))()
    rs = (CRng * max(1, count))()
    keepalive = []
    for i, g in enumerate(rngs):
        r = rng_struct(g)
        if r is None:
            rs[i].kind = RNG_OS
        else:
            keepalive.append(r)
            rs[i] = r
    ph = (C.c_void_p * len(provers))(*[p.handle.value for p in provers])
    rc = lib.dehalo_create_proofs(ph, len(provers), (C.c_void_p * max(1, count))(*adv_ptrs), count, rs, PROOF_ADVICE_ON_DEVICE if on_device else 0,
                                  (C.c_void_p * max(1, count))(*[b.ctypes.data for b in bufs]), cap, lens)
    if rc != 0:
        raise DehaloError(rc, "; ".join(lib.dehalo_last_error(p.ctx.handle).decode() for p in provers))
    return [bufs[i][:lens[i]].tobytes() for i in range(count)]


def create_proofs_circuit(provers: Sequence[Prover], circuit: int, inputs: Sequence[dict], rngs: Sequence) -> List[bytes]:
    """Batch mode with every proof's circuit synthesized inside its call (dehalo_create_proofs_circuit): inputs[i] = the keyword arguments of
    synthesize() for proof i (n_big, e, x, exp_bits, message, key, bits_len)."""
    lib = load_library()
    count = len(inputs)
    k = provers[0].pk.info()["k"]
    arr = (CCircuitInputs * max(1, count))()
    keep = []
    for i, kw in enumerate(inputs):
        inp, arrays = circuit_inputs(circuit, k, **kw)
        arr[i] = inp
        keep.append(arrays)
    info = provers[0].pk.info()
    cap = 32 * (info["commitments_before_evaluations"] + info["evaluations"] + info["opening_points"])
    bufs = [np.empty(cap, dtype=np.uint8) for _ in range(count)]
    lens = (C.c_size_t * max(1, count))()
    rs = (CRng * max(1, count))()
    keepalive = []
    for i, g in enumerate(rngs):
        r = rng_struct(g)
        if r is None:
            rs[i].kind = RNG_OS
        else:
            keepalive.append(r)
            rs[i] = r
    ph = (C.c_void_p * len(provers))(*[p.handle.value for p in provers])
    rc = lib.dehalo_create_proofs_circuit(ph, len(provers), arr, count, rs, (C.c_void_p * max(1, count))(*[b.ctypes.data for b in bufs]), cap, lens)
    if rc != 0:
        raise DehaloError(rc, "; ".join(lib.dehalo_last_error(p.ctx.handle).decode() for p in provers))
    return [bufs[i][:lens[i]].tobytes() for i in range(count)]


def _limbs(x: int, count: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(8 * count, "little"), dtype=np.uint64).copy()


def circuit_inputs(circuit: int, k: int, *, n_big: int = 0, e: int = 0, x: int = 0, exp_bits: int = 0, message: Sequence[int] = (), key: Sequence[int] = (), bits_len: int = 2048):
    """-> (dehalo_circuit_inputs, the arrays it points into): the reference's circuit structs as values (DelayEncryptCircuit { n, e, x, spec, enc_key... },
    RSACircuit, PoseidonEncCircuit)."""
    nl = bits_len // 64
    inp = CCircuitInputs()
    inp.circuit, inp.k, inp.bits_len, inp.exp_bits, inp.e = circuit, k, bits_len, exp_bits, e
    keep = [_limbs(n_big, nl), _limbs(x, nl), np.ascontiguousarray(np.frombuffer(b"".join(int(m).to_bytes(32, "little") for m in message) or bytes(32), dtype=np.uint64)),
            np.ascontiguousarray(np.frombuffer(b"".join(int(m).to_bytes(32, "little") for m in key) or bytes(64), dtype=np.uint64))]
    inp.n, inp.x, inp.message, inp.message_len, inp.key = keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, len(message), keep[3].ctypes.data
    return inp, keep


def _info_dict(info) -> dict:
    return {"rows": int(info.total_rows), "rsa_rows": int(info.rsa_rows), "rsa_result": sum(int(v) << (64 * i) for i, v in enumerate(info.rsa_result)),
            "cipher": [sum(int(info.cipher[4 * i + j]) << (64 * j) for j in range(4)) for i in range(info.cipher_len)]}


def synthesize(circuit: int, k: int, *, n_big: int = 0, e: int = 0, x: int = 0, exp_bits: int = 0, message: Sequence[int] = (), key: Sequence[int] = (), bits_len: int = 2048,
               keygen: bool = False, out=None) -> dict:
    """dehalo_synthesize: Circuit::synthesize of the reference's circuits as values, in C++ (csrc/witness.hip; src/lib.rs:164-318,
    benches/mod_pow.rs:63-110, src/encryption/chip.rs:114-204).  -> {"advice": (5, n, 4) u64 CANONICAL, "rows", "rsa_rows", "rsa_result",
    "cipher"} and, with keygen=True, "fixed" (canonical), "mapping", "selectors".  `out`: a C-contiguous (5, n, 4) uint64 array the advice columns
    are written into (the library writes them in place: a caller that proves repeatedly hands over one page-locked buffer and uploads from it)."""
    lib = load_library()
    inp, keep = circuit_inputs(circuit, k, n_big=n_big, e=e, x=x, exp_bits=exp_bits, message=message, key=key, bits_len=bits_len)
    n = 1 << k
    nfix = 9 if circuit == CIRCUIT_POSE_ENC else 15
    if out is not None:
        if out.shape != (5, n, 4) or out.dtype != np.uint64 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("synthesize: `out` must be a C-contiguous (5, %d, 4) uint64 array" % n)
        advice = out
    else:
        advice = np.empty((5, n, 4), dtype=np.uint64)
    fixed = np.empty((nfix, n, 4), dtype=np.uint64) if keygen else None
    mapping = np.empty(6 * n, dtype=np.uint64) if keygen else None
    sels = [np.zeros(n, dtype=np.uint8) for _ in range(2)] if keygen and circuit != CIRCUIT_POSE_ENC else []
    sel_ptrs = (C.c_void_p * 2)(*[s.ctypes.data for s in sels]) if sels else None
    info = CSynthesisInfo()
    rc = lib.dehalo_synthesize(C.byref(inp), advice.ctypes.data, fixed.ctypes.data if keygen else None, mapping.ctypes.data if keygen else None, sel_ptrs, C.byref(info))
    if rc != 0:
        raise ValueError("dehalo_synthesize failed (%d): bad inputs or not enough rows available" % rc)
    out = dict(_info_dict(info), advice=advice)
    if keygen:
        out.update(fixed=fixed, mapping=mapping, selectors=[s.astype(bool) for s in sels])
    return out
