//! The one-call integration: `plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC<_>, Challenge255<_>, _, Blake2bWrite<_, _, _>, _>` behind the C ABI
//! [UPSTREAM halo2_proofs/src/plonk/prover.rs @ v2023_04_20; the reference's timed call: benches/delay_enc.rs:123-131, benches/mod_pow.rs:201-209,
//! benches/pose_enc.rs:127-135].  From the first advice commitment to the last opening -- phases, Blake2b transcript, blinding, every launch -- in C++
//! (csrc/prover.hip); this module builds the objects and passes pointers, the sequence of `host/example.cpp` (which `tests/test_native.py` runs).
//!
//! The patched `plonk/prover.rs` keeps its signature; its body becomes
//!
//! ```ignore
//! let advice = synthesize_advice(params, pk, circuits, instances)?;          // upstream's own first step, unchanged: WitnessCollection + batch_invert_assigned
//! let proof = DEHALO.with(|h| h.prover.create_proof(&advice, instances[0], &mut rng))?;
//! transcript.write_raw(&proof)                                                // same wire format: 32-byte compressed points, 32-byte LE scalars
//! ```
use crate::constraint_system::Descriptor;
use crate::params::DehaloParamsKZG;
use crate::{Context, DehaloError};
use core::ffi::c_void;
use dehalo_sys as sys;
use ff::Field;
use halo2_proofs::plonk::ConstraintSystem;
use halo2curves::bn256::Fr;
use rand_core::RngCore;

/// `ProvingKey<G1Affine>` (with its `VerifyingKey`) on the device.
pub struct DehaloProvingKey<'c> {
    ctx: &'c Context,
    raw: *mut sys::dehalo_pk,
}

impl<'c> DehaloProvingKey<'c> {
    /// `keygen_vk` + `keygen_pk` (benches/delay_enc.rs:86,103): `fixed` = the circuit's fixed columns after selector compression, `num_fixed x 2^k` Montgomery
    /// elements, column after column; `mapping` = the permutation assembly, `columns x 2^k` cells as `column * 2^k + row`; `selectors` only feed `transcript_repr`.
    pub fn keygen(ctx: &'c Context, params: &DehaloParamsKZG<'c>, cs: &ConstraintSystem<Fr>, fixed: &[Fr], mapping: &[u64], selectors: &[Vec<u8>]) -> Result<Self, DehaloError> {
        let d = Descriptor::from_constraint_system(cs);
        let c = d.as_c();
        let sel: Vec<*const u8> = selectors.iter().map(|s| s.as_ptr()).collect();
        let mut raw = core::ptr::null_mut();
        ctx.check(unsafe {
            sys::dehalo_keygen(ctx.as_ptr(), params.as_ptr(), &c, fixed.as_ptr() as *const u64, mapping.as_ptr(), sel.as_ptr(), sel.len() as u32, 0, &mut raw)
        })?;
        Ok(DehaloProvingKey { ctx, raw })
    }

    /// `ProvingKey::read::<_, Circuit>(&mut reader, SerdeFormat::RawBytes)` of the file the bench caches (benches/delay_enc.rs:111-115).
    pub fn read(ctx: &'c Context, cs: &ConstraintSystem<Fr>, bytes: &[u8], num_selectors: u32) -> Result<Self, DehaloError> {
        let d = Descriptor::from_constraint_system(cs);
        let c = d.as_c();
        let mut raw = core::ptr::null_mut();
        ctx.check(unsafe { sys::dehalo_pk_read(ctx.as_ptr(), sys::DEHALO_CURVE_BN254_G1, &c, bytes.as_ptr(), bytes.len(), num_selectors, &mut raw) })?;
        Ok(DehaloProvingKey { ctx, raw })
    }

    /// `pk.write(&mut writer, SerdeFormat::RawBytes)` (`:105`)
    pub fn write(&self) -> Result<Vec<u8>, DehaloError> {
        let mut out = vec![0u8; unsafe { sys::dehalo_pk_size(self.raw) }];
        self.ctx.check(unsafe { sys::dehalo_pk_write(self.ctx.as_ptr(), self.raw, out.as_mut_ptr(), out.len()) })?;
        Ok(out)
    }

    /// `vk.write(..)` (`:88`)
    pub fn vk_write(&self) -> Result<Vec<u8>, DehaloError> {
        let mut out = vec![0u8; unsafe { sys::dehalo_vk_size(self.raw) }];
        self.ctx.check(unsafe { sys::dehalo_vk_write(self.raw, out.as_mut_ptr(), out.len()) })?;
        Ok(out)
    }

    /// upstream derives `vk.transcript_repr` from Rust `Debug` text, so it is passed in: with upstream's value set, the first absorbed scalar is upstream's
    pub fn set_transcript_repr(&mut self, repr: &Fr) -> Result<(), DehaloError> {
        self.ctx.check(unsafe { sys::dehalo_pk_set_transcript_repr(self.raw, repr as *const Fr as *const u64) })
    }
}

impl Drop for DehaloProvingKey<'_> {
    fn drop(&mut self) {
        unsafe { sys::dehalo_pk_release(self.ctx.as_ptr(), self.raw) };
    }
}

/// `dehalo_rng` with `DEHALO_RNG_CALLBACK`: blinding scalars are drawn from the caller's `RngCore` in upstream's order.  (The random polynomial's `n` scalars are
/// requested early, from a helper thread, with their `position` in that order: a generator that cannot seek sees them out of order -- the proof is equally valid;
/// byte identity with a CPU run of the same seeded generator needs a seekable one, which `DEHALO_RNG_PCG64` is.)
/// `R: Send`: include/dehalo.h lets the library call `fill` from a helper thread of its own, so the generator crosses threads (`ThreadRng` is refused at compile time).
unsafe extern "C" fn fill_from<R: RngCore + Send>(user: *mut c_void, out: *mut u64, count: usize, _position: u64) -> i32 {
    let rng = &mut *(user as *mut R);
    for i in 0..count {
        let s = Fr::random(&mut *rng);
        core::ptr::copy_nonoverlapping(&s as *const Fr as *const u64, out.add(4 * i), 4);
    }
    0
}

/// Device buffers of ONE proof in flight, reused by every `create_proof` on it; `side` = a second context whose stream runs the NTTs and the random
/// polynomial's commitment beside the commitment phases.
pub struct DehaloProver<'c> {
    ctx: &'c Context,
    raw: *mut sys::dehalo_prover,
}

impl<'c> DehaloProver<'c> {
    pub fn new(ctx: &'c Context, side: Option<&'c Context>, params: &DehaloParamsKZG<'c>, pk: &DehaloProvingKey<'c>) -> Result<Self, DehaloError> {
        let mut raw = core::ptr::null_mut();
        ctx.check(unsafe { sys::dehalo_prover_create(ctx.as_ptr(), side.map_or(core::ptr::null_mut(), |s| s.as_ptr()), params.as_ptr(), pk.raw, &mut raw) })?;
        Ok(DehaloProver { ctx, raw })
    }

    /// ONE proof on several GPUs (`dehalo_prover_set_shard`): of every multi-column commitment phase this process runs the MSMs of its share of the columns only and
    /// `gather` (an all-gather of at most ten affine points over RCCL / xGMI, see include/dehalo.h) fills in the rest; every process must then call `create_proof`
    /// with the same advice, instances and the same seeded generator.  `world == 1` switches it off.
    ///
    /// # Safety
    /// `gather` is called on the proving thread with `user`; both must stay valid for as long as the prover may prove.
    pub unsafe fn set_shard(&self, rank: u32, world: u32, gather: sys::dehalo_gather_fn, user: *mut c_void) -> Result<(), DehaloError> {
        self.ctx.check(sys::dehalo_prover_set_shard(self.raw, rank, world, gather, user))
    }

    /// `create_proof(&params, &pk, &[circuit], &[instances], rng, &mut transcript)` for ONE circuit: `advice` = what `circuit.synthesize` assigned
    /// (`num_advice x 2^k` Montgomery elements, column after column, host memory); returns the transcript bytes (`transcript.finalize()`).
    pub fn create_proof<R: RngCore + Send>(&self, advice: &[Fr], instances: &[&[Fr]], rng: &mut R) -> Result<Vec<u8>, DehaloError> {
        let inst_ptrs: Vec<*const u64> = instances.iter().map(|c| c.as_ptr() as *const u64).collect();
        let inst_lens: Vec<usize> = instances.iter().map(|c| c.len()).collect();
        let mut r = sys::dehalo_rng { kind: sys::DEHALO_RNG_CALLBACK, pcg_state: [0; 2], pcg_inc: [0; 2], fill: Some(fill_from::<R>), user: rng as *mut R as *mut c_void };
        let mut t = core::ptr::null_mut();
        self.ctx.check(unsafe { sys::dehalo_transcript_create(sys::DEHALO_CURVE_BN254_G1, &mut t) })?;
        let rc = unsafe {
            sys::dehalo_create_proof(self.raw, advice.as_ptr() as *const u64, inst_ptrs.as_ptr(), inst_lens.as_ptr(), inst_ptrs.len() as u32, &mut r, t, 0)
        };
        let out = if rc == 0 {
            let mut bytes = vec![0u8; unsafe { sys::dehalo_transcript_len(t) }];
            let rc2 = unsafe { sys::dehalo_transcript_finalize(t, bytes.as_mut_ptr(), bytes.len()) };
            if rc2 == 0 { Ok(bytes) } else { self.ctx.check(rc2).map(|_| vec![]) }
        } else {
            self.ctx.check(rc).map(|_| vec![])
        };
        unsafe { sys::dehalo_transcript_release(t) };
        out
    }

    /// The reference's call shape in one entry point: the circuit's own inputs instead of its advice columns -- synthesized inside the call by the library's restatement
    /// of the three circuits (csrc/witness.hip: halo2wrong's layout instruction by instruction, unverified cell for cell against the crates -- pair it with a key from `keygen` over ITS fixed columns: `dehalo_synthesize`).
    pub fn create_proof_circuit<R: RngCore + Send>(&self, inputs: &sys::dehalo_circuit_inputs, rng: &mut R) -> Result<(Vec<u8>, sys::dehalo_synthesis_info), DehaloError> {
        let mut r = sys::dehalo_rng { kind: sys::DEHALO_RNG_CALLBACK, pcg_state: [0; 2], pcg_inc: [0; 2], fill: Some(fill_from::<R>), user: rng as *mut R as *mut c_void };
        let mut info: sys::dehalo_synthesis_info = unsafe { core::mem::zeroed() };
        let mut t = core::ptr::null_mut();
        self.ctx.check(unsafe { sys::dehalo_transcript_create(sys::DEHALO_CURVE_BN254_G1, &mut t) })?;
        let empty: [*const u64; 1] = [core::ptr::null()];
        let lens: [usize; 1] = [0];
        let rc = unsafe { sys::dehalo_create_proof_circuit(self.raw, inputs, &mut info, empty.as_ptr(), lens.as_ptr(), 1, &mut r, t) };
        let out = if rc == 0 {
            let mut bytes = vec![0u8; unsafe { sys::dehalo_transcript_len(t) }];
            let rc2 = unsafe { sys::dehalo_transcript_finalize(t, bytes.as_mut_ptr(), bytes.len()) };
            if rc2 == 0 { Ok((bytes, info)) } else { self.ctx.check(rc2).map(|_| (vec![], info)) }
        } else {
            self.ctx.check(rc).map(|_| (vec![], info))
        };
        unsafe { sys::dehalo_transcript_release(t) };
        out
    }

    /// milliseconds of the last proof by phase: advice, lookups, products, random, quotient, evaluations, openings, total
    pub fn last_timings(&self) -> Result<[f64; 8], DehaloError> {
        let mut t = [0f64; 8];
        self.ctx.check(unsafe { sys::dehalo_prover_last_timings(self.raw, t.as_mut_ptr()) })?;
        Ok(t)
    }
}

impl Drop for DehaloProver<'_> {
    fn drop(&mut self) {
        unsafe { sys::dehalo_prover_release(self.raw) };
    }
}

/// The bench's flow end to end (benches/delay_enc.rs:41-54, 84-131) for a caller that holds upstream's objects: SRS file bytes, the circuit's constraint system, its
/// fixed columns / permutation mapping, the advice columns of one proof.  One-time objects are built once and reused by the caller; shown as one function for the reader.
pub fn create_proof<R: RngCore + Send>(device: i32, params_raw_bytes: &[u8], cs: &ConstraintSystem<Fr>, fixed: &[Fr], mapping: &[u64], selectors: &[Vec<u8>],
                                transcript_repr: &Fr, advice: &[Fr], instances: &[&[Fr]], rng: &mut R) -> Result<Vec<u8>, DehaloError> {
    let ctx = Context::new(device)?;
    let side = Context::with_priority(device, 1)?;
    let params = DehaloParamsKZG::read(&ctx, params_raw_bytes)?;
    let mut pk = DehaloProvingKey::keygen(&ctx, &params, cs, fixed, mapping, selectors)?;
    pk.set_transcript_repr(transcript_repr)?;
    let prover = DehaloProver::new(&ctx, Some(&side), &params, &pk)?;
    prover.create_proof(advice, instances, rng)
}
