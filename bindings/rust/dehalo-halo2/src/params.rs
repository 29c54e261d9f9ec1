//! `ParamsKZG<Bn256>` with `g` / `g_lagrange` resident in HBM [UPSTREAM halo2_proofs/src/poly/kzg/commitment.rs @ v2023_04_20; reference call sites
//! benches/delay_enc.rs:41-54 -- `ParamsKZG::<Bn256>::setup(K, OsRng)`, `params.write(..)`, `ParamsKZG::read(..)`].
//!
//! The patched `ParamsKZG` keeps upstream's public interface and holds one of these next to (or instead of) its `Vec<G1Affine>` fields:
//!
//! ```ignore
//! impl<'params> Params<'params, G1Affine> for ParamsKZG<Bn256> {
//!     fn commit_lagrange(&self, poly: &Polynomial<Fr, LagrangeCoeff>, _: Blind<Fr>) -> G1 { self.dehalo.commit_lagrange(&poly.values) }
//! }
//! impl<'params> ParamsProver<'params, G1Affine> for ParamsKZG<Bn256> {
//!     fn commit(&self, poly: &Polynomial<Fr, Coeff>, _: Blind<Fr>) -> G1 { self.dehalo.commit(&poly.values) }
//! }
//! ```
//! (KZG ignores the blind, exactly as upstream.)
use crate::{Context, DehaloError};
use dehalo_sys as sys;
use ff::Field;
use halo2curves::bn256::{Fr, G1};
use rand_core::RngCore;
use std::cell::Cell;

/// `dehalo_params`: the SRS with its precomputed window tables on the device and the RawBytes image on the host.
pub struct DehaloParamsKZG<'c> {
    pub(crate) ctx: &'c Context,
    pub(crate) raw: *mut sys::dehalo_params,
    pub k: u32,
    bases_g: Cell<*mut sys::dehalo_bases>,
    bases_gl: Cell<*mut sys::dehalo_bases>,
}

impl<'c> DehaloParamsKZG<'c> {
    /// `ParamsKZG::<Bn256>::setup(k, rng)` (benches/delay_enc.rs:43): draws `s` exactly as upstream does (`<Fr>::random(rng)`) and hands it over; `g[i] = [s^i] G` and
    /// `g_lagrange[i] = [L_i(s)] G` are made on the device.  k <= 25 (include/dehalo.h).
    pub fn setup<R: RngCore>(ctx: &'c Context, k: u32, mut rng: R) -> Result<Self, DehaloError> {
        let s = Fr::random(&mut rng);
        let mut raw = core::ptr::null_mut();
        ctx.check(unsafe { sys::dehalo_params_setup(ctx.as_ptr(), sys::DEHALO_CURVE_BN254_G1, k, &s as *const Fr as *const u64, &mut raw) })?;
        Self::finish(ctx, raw, k)
    }

    /// `ParamsKZG::read(&mut reader)` of the RawBytes file the bench caches (`:54`): `k: u32 LE | g | g_lagrange | g2 | s_g2`.
    pub fn read(ctx: &'c Context, bytes: &[u8]) -> Result<Self, DehaloError> {
        let mut raw = core::ptr::null_mut();
        ctx.check(unsafe { sys::dehalo_params_read(ctx.as_ptr(), sys::DEHALO_CURVE_BN254_G1, bytes.as_ptr(), bytes.len(), &mut raw) })?;
        let k = u32::from_le_bytes([bytes[0], bytes[1], bytes[2], bytes[3]]);
        Self::finish(ctx, raw, k)
    }

    /// From vectors upstream already holds (`params.get_g()`, `g_lagrange`, the two G2 points as 128 raw bytes each).
    pub fn from_vectors(ctx: &'c Context, k: u32, g: &[halo2curves::bn256::G1Affine], g_lagrange: &[halo2curves::bn256::G1Affine], g2: &[u8; 128], s_g2: &[u8; 128]) -> Result<Self, DehaloError> {
        assert_eq!(g.len(), 1usize << k);
        assert_eq!(g_lagrange.len(), 1usize << k);
        let mut raw = core::ptr::null_mut();
        ctx.check(unsafe {
            sys::dehalo_params_create(ctx.as_ptr(), sys::DEHALO_CURVE_BN254_G1, k, g.as_ptr() as *const u64, g_lagrange.as_ptr() as *const u64, g2.as_ptr(), s_g2.as_ptr(), &mut raw)
        })?;
        Self::finish(ctx, raw, k)
    }

    fn finish(ctx: &'c Context, raw: *mut sys::dehalo_params, k: u32) -> Result<Self, DehaloError> {
        Ok(DehaloParamsKZG { ctx, raw, k, bases_g: Cell::new(core::ptr::null_mut()), bases_gl: Cell::new(core::ptr::null_mut()) })
    }

    /// `params.write(&mut writer)` (RawBytes; benches/delay_enc.rs:45)
    pub fn write(&self) -> Result<Vec<u8>, DehaloError> {
        let mut out = vec![0u8; unsafe { sys::dehalo_params_size(self.raw) }];
        self.ctx.check(unsafe { sys::dehalo_params_write(self.raw, out.as_mut_ptr(), out.len()) })?;
        Ok(out)
    }

    /// `ParamsProver::commit(poly, _)`: `best_multiexp(&poly.values, &self.g[..len])` over the resident table; `len` may be shorter than n (a prefix).
    pub fn commit(&self, coeffs: &[Fr]) -> G1 {
        self.msm(coeffs, false)
    }

    /// `Params::commit_lagrange(poly, _)`: the same over `g_lagrange`.
    pub fn commit_lagrange(&self, values: &[Fr]) -> G1 {
        self.msm(values, true)
    }

    fn msm(&self, scalars: &[Fr], lagrange: bool) -> G1 {
        assert!(scalars.len() <= 1usize << self.k);
        // dehalo_params_commit_device is the prover's path (polynomials already in HBM); for a host slice the entry is dehalo_msm over registered bases.  Same group element.
        let n = scalars.len();
        let mut out = [0u64; 12];
        let bases = self.bases(lagrange).expect("dehalo: registering the SRS");
        self.ctx.check(unsafe { sys::dehalo_msm(self.ctx.as_ptr(), bases, scalars.as_ptr() as *const u64, n, out.as_mut_ptr()) }).expect("dehalo_msm");
        unsafe { core::mem::transmute_copy::<[u64; 12], G1>(&out) }
    }

    /// The bases handle of `g` / `g_lagrange` for `dehalo_msm` on host scalars.  `dehalo_params` keeps its own tables for the prover; a host-slice commit registers
    /// the same points once more on first use (the RawBytes image holds them) and keeps the handle.  (`Cell`: the struct is not `Sync`; one per proving thread.)
    fn bases(&self, lagrange: bool) -> Result<*const sys::dehalo_bases, DehaloError> {
        let slot = if lagrange { &self.bases_gl } else { &self.bases_g };
        if slot.get().is_null() {
            let raw = self.write()?;
            let n = 1usize << self.k;
            let off = 4 + if lagrange { 64 * n } else { 0 };
            // the file's points are 64-byte {x, y} Montgomery records: the layout dehalo_bases_register takes (stride 64); copied to an aligned buffer first
            let mut pts = vec![0u64; 8 * n];
            unsafe { core::ptr::copy_nonoverlapping(raw.as_ptr().add(off), pts.as_mut_ptr() as *mut u8, 64 * n) };
            let mut h = core::ptr::null_mut();
            self.ctx.check(unsafe { sys::dehalo_bases_register(self.ctx.as_ptr(), sys::DEHALO_CURVE_BN254_G1, pts.as_ptr(), n, 64, 0, 1, &mut h) })?;
            slot.set(h);
        }
        Ok(slot.get() as *const sys::dehalo_bases)
    }

    pub fn as_ptr(&self) -> *const sys::dehalo_params {
        self.raw
    }
}

impl Drop for DehaloParamsKZG<'_> {
    fn drop(&mut self) {
        unsafe {
            for h in [self.bases_g.get(), self.bases_gl.get()] {
                if !h.is_null() {
                    sys::dehalo_bases_release(self.ctx.as_ptr(), h);
                }
            }
            sys::dehalo_params_release(self.ctx.as_ptr(), self.raw);
        }
    }
}
