//! Drop-in bodies for `halo2_proofs::arithmetic::{best_multiexp, best_fft}` [UPSTREAM halo2_proofs/src/arithmetic.rs @ v2023_04_20; reached from
//! `Params::commit*` and `EvaluationDomain` inside `create_proof`, benches/delay_enc.rs:123-131].
//!
//! Upstream's contract is kept: slices are borrowed and never retained, `assert_eq!` on the lengths (a panic, not a `Result`), pure functions.  The patched
//! `arithmetic.rs` is two forwarding lines:
//!
//! ```ignore
//! pub fn best_multiexp<C: CurveAffine + DehaloCurve>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve { dehalo_halo2::arithmetic::best_multiexp(coeffs, bases) }
//! pub fn best_fft<F: PrimeField + DehaloField>(a: &mut [F], omega: F, log_n: u32) { dehalo_halo2::arithmetic::best_fft(a, omega, log_n) }
//! ```
//! (`best_fft`'s group-valued instantiation `G = C::Curve` only occurs in `ParamsIPA::new`; it keeps upstream's body.)
use crate::ctx;
use dehalo_sys as sys;
use halo2curves::bn256;
use halo2curves::pasta::{EpAffine, EqAffine, Fp as PastaFp, Fq as PastaFq};
use halo2curves::CurveAffine;

/// Curve identities understood by the library (`dehalo_curve`).  Layout facts relied on: the affine struct is `{x, y}` of two
/// 4 x u64 Montgomery field elements (64 bytes), the identity is `(0, 0)`; the projective struct is Jacobian `{x, y, z}` (96 bytes).
pub trait DehaloCurve: CurveAffine {
    const CURVE_ID: i32;
}
impl DehaloCurve for bn256::G1Affine {
    const CURVE_ID: i32 = sys::DEHALO_CURVE_BN254_G1;
}
impl DehaloCurve for EpAffine {
    const CURVE_ID: i32 = sys::DEHALO_CURVE_PALLAS;
}
impl DehaloCurve for EqAffine {
    const CURVE_ID: i32 = sys::DEHALO_CURVE_VESTA;
}

/// Field identities (`dehalo_field`): 4 x u64 little-endian limbs in Montgomery form (R = 2^256), the in-memory representation of halo2curves' fields.
pub trait DehaloField: ff::PrimeField {
    const FIELD_ID: i32;
}
impl DehaloField for bn256::Fr {
    const FIELD_ID: i32 = sys::DEHALO_FIELD_BN254_FR;
}
impl DehaloField for bn256::Fq {
    const FIELD_ID: i32 = sys::DEHALO_FIELD_BN254_FQ;
}
impl DehaloField for PastaFp {
    const FIELD_ID: i32 = sys::DEHALO_FIELD_PASTA_FP;
}
impl DehaloField for PastaFq {
    const FIELD_ID: i32 = sys::DEHALO_FIELD_PASTA_FQ;
}

/// `best_multiexp(coeffs, bases) = sum_i coeffs[i] * bases[i]`.  One-shot form: the bases are uploaded for this call (`dehalo_best_multiexp`); a caller that
/// commits repeatedly against one SRS uses [`crate::params::DehaloParamsKZG`], whose tables stay resident.
pub fn best_multiexp<C: DehaloCurve>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {
    assert_eq!(coeffs.len(), bases.len()); // upstream's contract
    assert_eq!(core::mem::size_of::<C>(), 64, "affine points must be {{x, y}} (64 bytes); pass another stride through dehalo_bases_register");
    assert_eq!(core::mem::size_of::<C::Scalar>(), 32);
    assert_eq!(core::mem::size_of::<C::Curve>(), 96);
    let mut out = [0u64; 12];
    let c = ctx();
    let rc = unsafe { sys::dehalo_best_multiexp(c.as_ptr(), C::CURVE_ID, coeffs.as_ptr() as *const u64, bases.as_ptr() as *const u64, coeffs.len(), out.as_mut_ptr()) };
    c.check(rc).expect("dehalo_best_multiexp");
    // Jacobian {x, y, z} in Montgomery limbs is the memory image of C::Curve.  (Any representative of the class: compare after to_affine(), as between two
    // CPU runs with different thread counts.)
    unsafe { core::mem::transmute_copy::<[u64; 12], C::Curve>(&out) }
}

/// `best_fft(a, omega, log_n)`: in place, natural order in and out, `omega` a primitive 2^log_n-th root (forward or inverse), no scaling.
pub fn best_fft<F: DehaloField>(a: &mut [F], omega: F, log_n: u32) {
    assert_eq!(a.len(), 1usize << log_n); // upstream's contract
    assert_eq!(core::mem::size_of::<F>(), 32);
    let c = ctx();
    let rc = unsafe { sys::dehalo_ntt(c.as_ptr(), F::FIELD_ID, a.as_mut_ptr() as *mut u64, log_n, &omega as *const F as *const u64) };
    c.check(rc).expect("dehalo_ntt");
}

/// `EvaluationDomain::lagrange_to_coeff`: inverse transform and the 1/n scaling in one call (`omega_inv`, `ifft_divisor` are the domain's own constants).
pub fn lagrange_to_coeff<F: DehaloField>(a: &mut [F], omega_inv: F, ifft_divisor: F, k: u32) {
    assert_eq!(a.len(), 1usize << k);
    let c = ctx();
    let rc = unsafe { sys::dehalo_intt_scaled(c.as_ptr(), F::FIELD_ID, a.as_mut_ptr() as *mut u64, k, &omega_inv as *const F as *const u64, &ifft_divisor as *const F as *const u64) };
    c.check(rc).expect("dehalo_intt_scaled");
}

/// `EvaluationDomain::coeff_to_extended`: x zeta^(i mod 3), zero-padding to 2^extended_k and the forward transform (`g_coset` = the domain's zeta).
pub fn coeff_to_extended<F: DehaloField>(coeffs: &[F], k: u32, extended_k: u32, extended_omega: F, g_coset: F) -> Vec<F> {
    assert_eq!(coeffs.len(), 1usize << k);
    let mut ext = vec![F::ZERO; 1usize << extended_k];
    let c = ctx();
    let rc = unsafe {
        sys::dehalo_coset_ntt(c.as_ptr(), F::FIELD_ID, coeffs.as_ptr() as *const u64, k, ext.as_mut_ptr() as *mut u64, extended_k,
                              &extended_omega as *const F as *const u64, &g_coset as *const F as *const u64)
    };
    c.check(rc).expect("dehalo_coset_ntt");
    ext
}

/// `EvaluationDomain::extended_to_coeff` (before upstream's truncation to the quotient's length).
pub fn extended_to_coeff<F: DehaloField>(a: &mut [F], extended_k: u32, extended_omega_inv: F, extended_ifft_divisor: F, g_coset: F) {
    assert_eq!(a.len(), 1usize << extended_k);
    let c = ctx();
    let rc = unsafe {
        sys::dehalo_coset_intt(c.as_ptr(), F::FIELD_ID, a.as_mut_ptr() as *mut u64, extended_k, &extended_omega_inv as *const F as *const u64,
                               &extended_ifft_divisor as *const F as *const u64, &g_coset as *const F as *const u64)
    };
    c.check(rc).expect("dehalo_coset_intt");
}
