//! dehalo-halo2: the halo2_proofs-shaped front of libdehalo.so (MI355X).  Source only -- the build image of the repository has no Rust toolchain, so this crate
//! is NOT compiled or tested there; the same C ABI is exercised by the ctypes mirror (`delay-encryption-in-halo2_amd/_lib.py`, `native.py`) and the C++ mirror
//! (`host/halo2_backend.hpp`, `host/example.cpp`), whose call sequences these modules follow line by line.
//!
//! What a maintainer of the reference (`radiusxyz/delay-encryption-in-halo2`, Cargo.toml:17 -> halo2_proofs @ v2023_04_20) does with it:
//!
//! * fine-grained: `[patch]` halo2_proofs so that `arithmetic.rs::{best_multiexp, best_fft}` forward to [`arithmetic`], and
//!   `ParamsKZG::{setup, commit, commit_lagrange}` to [`params::DehaloParamsKZG`] (the SRS stays resident in HBM);
//! * one call: replace the body of `plonk::create_proof` for `KZGCommitmentScheme<Bn256>` / `ProverGWC` / `Blake2bWrite` by [`prover::create_proof`]
//!   (benches/delay_enc.rs:123-131), keys and SRS read from the same RawBytes files the bench caches (`:45,54,88,94-98,105,111-115`).
pub mod arithmetic;
pub mod constraint_system;
pub mod params;
pub mod prover;

use dehalo_sys as sys;
use std::ffi::CStr;
use std::sync::OnceLock;

/// Error of a library call: the status code of include/dehalo.h and the context's message.
#[derive(Debug, Clone)]
pub struct DehaloError {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for DehaloError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "dehalo error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for DehaloError {}

/// One device context (= one HIP stream + workspace).  `Send + Sync`: the library serialises calls on a context with its own mutex (include/dehalo.h).
pub struct Context(pub(crate) *mut sys::dehalo_ctx);
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

impl Context {
    /// `dehalo_ctx_create`.  There is no CPU fallback: without a gfx950 device this is `DEHALO_ERR_NO_DEVICE`.
    pub fn new(device: i32) -> Result<Self, DehaloError> {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { sys::dehalo_ctx_create(device, &mut p) };
        if rc != 0 {
            return Err(DehaloError { code: rc, message: "dehalo_ctx_create failed (no gfx950 device?)".into() });
        }
        Ok(Context(p))
    }
    /// a context whose short kernels must not queue behind another context's long ones (the prover's side context)
    pub fn with_priority(device: i32, priority: i32) -> Result<Self, DehaloError> {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { sys::dehalo_ctx_create_with_priority(device, priority, &mut p) };
        if rc != 0 {
            return Err(DehaloError { code: rc, message: "dehalo_ctx_create_with_priority failed".into() });
        }
        Ok(Context(p))
    }
    pub(crate) fn check(&self, rc: i32) -> Result<(), DehaloError> {
        if rc == 0 {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(sys::dehalo_last_error(self.0)) }.to_string_lossy().into_owned();
        Err(DehaloError { code: rc, message: msg })
    }
    pub fn as_ptr(&self) -> *mut sys::dehalo_ctx {
        self.0
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::dehalo_ctx_destroy(self.0) }
    }
}

static GLOBAL: OnceLock<Context> = OnceLock::new();

/// The process-wide context the free functions of [`arithmetic`] use (upstream's `best_multiexp` / `best_fft` take no handle).  Device ordinal from
/// `DEHALO_DEVICE` (default 0).  Creation failure is a panic, as a missing rayon pool would be upstream: a build that wants a CPU path keeps upstream's
/// functions under a cargo feature instead of calling these.
pub fn ctx() -> &'static Context {
    GLOBAL.get_or_init(|| {
        let dev = std::env::var("DEHALO_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        Context::new(dev).expect("dehalo: no MI355X context")
    })
}
