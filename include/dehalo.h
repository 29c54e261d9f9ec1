/*
 * dehalo.h -- C ABI of the MI355X (gfx950) MSM / NTT prover backend.
 *
 * Drop-in boundary for the hot path of radiusxyz/delay-encryption-in-halo2: the
 * commitments and FFTs that `create_proof` (reference call sites
 * benches/delay_enc.rs:123-131, benches/mod_pow.rs:201-209, benches/pose_enc.rs:127-135;
 * keygen at benches/delay_enc.rs:86,103) performs through the un-vendored dependency
 * halo2_proofs @ tag v2023_04_20 (Cargo.toml:17).  Each entry point names the upstream
 * function a patched halo2_proofs forwards to it (bindings: INTEGRATION.md).
 *
 * Conventions (all entry points):
 *  - plain pointers and sizes only; caller-owned buffers, never retained after return
 *    (except bases handed to dehalo_bases_register, which are COPIED to the device);
 *  - field element  = 4 x u64 little-endian limbs, MONTGOMERY form, R = 2^256: exactly the
 *    in-memory representation of halo2curves' bn256::{Fr,Fq} / pasta::{Fp,Fq};
 *  - affine point   = {x, y} (64 B), identity encoded as x = y = 0;
 *  - projective out = Jacobian {x, y, z} (96 B), identity z = 0.  Any representative of the
 *    class may be returned (as with rayon's thread-count-dependent summation order
 *    upstream); compare after to_affine();
 *  - return 0 on success, a negative dehalo_status otherwise; never aborts, never throws;
 *  - "host" entry points take host pointers and are synchronous; "_device" entry points
 *    take device pointers and a hipStream_t (as void*; NULL = the context's stream) and
 *    are asynchronous on that stream;
 *  - a context is bound to one device and owns one grow-only workspace in HBM that every call on
 *    it uses: entry points may be called from several threads (a mutex serialises them), but
 *    the device work of one context must be ordered -- use the context's own stream, or one
 *    stream of yours per context.  Work that should overlap (independent proofs, columns of
 *    different phases) goes to DIFFERENT contexts; registered bases and compiled graphs may be
 *    shared between contexts of the same device.  There is NO CPU fallback: without a usable
 *    gfx950 device dehalo_ctx_create fails with DEHALO_ERR_NO_DEVICE.
 */
#ifndef DEHALO_H
#define DEHALO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DEHALO_OK = 0,
    DEHALO_ERR_INVALID = -1,    /* bad argument: null pointer, length mismatch, log_n out of range */
    DEHALO_ERR_NO_DEVICE = -2,  /* no HIP device / wrong architecture */
    DEHALO_ERR_OOM = -3,        /* device allocation failed */
    DEHALO_ERR_HIP = -4,        /* any other HIP runtime error (see dehalo_last_error) */
    DEHALO_ERR_UNSUPPORTED = -5, /* e.g. NTT over a field whose two-adicity is too small */
    DEHALO_ERR_NOT_IN_TABLE = -6 /* permute_expression_pair: upstream's Error::ConstraintSystemFailure */
} dehalo_status;

/* halo2curves curve / field identities */
typedef enum { DEHALO_CURVE_BN254_G1 = 0, DEHALO_CURVE_PALLAS = 1, DEHALO_CURVE_VESTA = 2 } dehalo_curve;
typedef enum { DEHALO_FIELD_BN254_FR = 0, DEHALO_FIELD_BN254_FQ = 1, DEHALO_FIELD_PASTA_FP = 2, DEHALO_FIELD_PASTA_FQ = 3 } dehalo_field;

typedef struct dehalo_ctx dehalo_ctx;
typedef struct dehalo_bases dehalo_bases;

/* ---- context --------------------------------------------------------------------------- */
int dehalo_ctx_create(int device, dehalo_ctx** out);
/* The same with a stream priority: > 0 the device's highest (a context of short kernels that must not queue behind another context's
 * long ones -- the prover's side context), < 0 its lowest, 0 the default. */
int dehalo_ctx_create_with_priority(int device, int priority, dehalo_ctx** out);
void dehalo_ctx_destroy(dehalo_ctx* ctx);
/* Human-readable text of the last error on this context (a per-thread copy: valid until the calling thread's next call of this function). */
const char* dehalo_last_error(const dehalo_ctx* ctx);
/* Launch-geometry knobs (results never depend on them).  "msm_acc_points" (default 48): the bucket-accumulation grid of an MSM is 4, 6, 8, ... layers of one wave per SIMD,
 * the fewest that leave a lane at most this many points; 0 selects the older rule, whole rounds of "msm_acc_waves" in [1, 4]
 * waves per SIMD (below 4 the kernel leaves wave slots and registers free for the latency-bound kernels of other contexts).
 * "msm_sort_block" (1024, the default, or 512): threads per workgroup of the MSM sort's two scalar-decoding kernels; 512 halves their LDS (64 + 58 KiB instead of
 * 128 + 115 KiB: the histogram packs two 16-bit counters a word) so that they start on a compute unit another context's NTT tile or bucket reduction occupies.
 * "msm_acc_block" (128 or 768): threads per workgroup of the bucket accumulation.  768 = one
 * 12-wave workgroup per compute unit, three waves per SIMD, which leaves a quarter of every SIMD's registers and all of the LDS to the
 * kernels of other contexts that need at most 128 VGPRs (the bucket reduction, the merge, the NTT's half tiles).
 * "ntt_full_table_log" (default 0, in [0, 30]): transforms of up to 2^value points keep all N powers of omega on the device
 * (32 B x N) so that an inter-pass twiddle is one load; larger ones keep N / 2 and negate (measured equal on MI355X).
 * "host_wait_spin_us" (default 400, in [0, 1000000]): a host wait for the context's stream
 * (dehalo_download and the calls built on it: every transcript round trip of a proof) polls for up to this many microseconds before it
 * blocks in the runtime, whose wake-up comes ~15 us after the stream has drained; 0: block at once (no CPU spent waiting). */
int dehalo_ctx_set_tuning(dehalo_ctx* ctx, const char* key, int value);
/* Environment variables.  The library as shipped reads THREE, all host-side diagnostics; none changes which kernels run or with what geometry:
 *   DEHALO_SYNTH_THREADS    host threads dehalo_synthesize writes the RSA regions of a proving call with (default: 8, or the machine's hardware threads if fewer; 1: none)
 *   DEHALO_SYNTH_TRACE      dehalo_synthesize writes the time of its stages to stderr
 *   DEHALO_PROVER_TRACE     dehalo_create_proof writes the host's timeline inside the phases to stderr
 * Tuning is per context and validated: dehalo_ctx_set_tuning above.  The A/B switches of the measurement scripts (tools/ab_*.sh: DEHALO_MSM_BRED_BLOCK, DEHALO_NTT_SKIP,
 * DEHALO_CU_PARTITION, the *_STAMPS phase stamps, ...) exist only in a library built with `make EXPERIMENTS=1` (-DDEHALO_EXPERIMENTS); in the default build their
 * names are not in the binary (tests/test_abi.py counts them). */
/* The context's own stream (a hipStream_t): lets the caller order its own device work (copies, fills) with the library's
 * kernels by enqueueing it on the same stream. */
void* dehalo_ctx_stream(dehalo_ctx* ctx);
/* Blocks until everything queued on the context's own stream has finished. */
int dehalo_ctx_synchronize(dehalo_ctx* ctx);
/* Copies `bytes` from device memory to the host after everything queued on the context's own stream, and waits for it: the read-back
 * of a phase's commitments / evaluations (what Blake2bWrite::write_point needs on the host, halo2_proofs/src/transcript.rs) in one
 * call instead of a synchronize followed by a separate copy. */
int dehalo_download(dehalo_ctx* ctx, const void* d_src, size_t bytes, void* host_dst);
/* Copies `bytes` from CALLER host memory (a `&[F]`, any ordinary pageable allocation) to device memory on the context's own stream and
 * returns once the source has been read in full and the copy has completed: what a host does with a witness column or a polynomial it
 * wants resident (the `Vec<F>` -> HBM step in front of every `*_device` entry point).  Like every host entry point of this library the
 * transfer never leaves the device holding a mapping of caller memory: the bytes pass through the context's page-locked staging chunks
 * (or move by DMA from pages that are page-locked by the caller / for the duration of the call). */
int dehalo_upload(dehalo_ctx* ctx, const void* host_src, size_t bytes, void* d_dst);

/* ---- SRS / bases ------------------------------------------------------------------------
 * Replaces the `g` / `g_lagrange` vectors of ParamsKZG / ParamsIPA
 * [halo2_proofs/src/poly/kzg/commitment.rs, .../ipa/commitment.rs; built at
 * benches/delay_enc.rs:41-54].  Bases are constant per SRS, so they are uploaded once and
 * stay resident in HBM.
 *   affine_xy     n points, `stride_bytes` apart (>= 64; 64 for halo2curves' {x, y} structs)
 *   window_bits   0 = choose from n (with precompute = 1: 17 from 2^20 points, 16 at 2^19, 15 from 2^17, 13 below); otherwise the Pippenger window c in [4, 16], or 17 with precompute = 1
 *   precompute    1 = also store [2^(c*w)]P_i for every window w (n * ceil(256/c) * 64 B of
 *                 HBM): all windows then share one bucket set and the per-window doublings
 *                 vanish.  0 = store the n points only.
 * Limits: n < 2^30; with precompute = 1 also n x windows < 2^30 (30-bit table indices in the sorted list; windows = the signed-digit windows of the
 * scalar field at the chosen c, e.g. 15 for BN254 and the Pasta fields at c = 17): the default window admits tables up to n = 2^26.  Beyond: DEHALO_ERR_INVALID.
 */
int dehalo_bases_register(dehalo_ctx* ctx, int curve, const uint64_t* affine_xy, size_t n, size_t stride_bytes,
                          int window_bits, int precompute, dehalo_bases** out);
/* The same for points that are in device memory already (n x {x, y}, 64 bytes apart, standard Montgomery form): an SRS generated on the device
 * (dehalo_params_setup) or left there by the caller.  The points are read before the call returns. */
int dehalo_bases_register_device(dehalo_ctx* ctx, int curve, const uint64_t* d_affine_xy, size_t n, int window_bits, int precompute, dehalo_bases** out);
int dehalo_bases_release(dehalo_ctx* ctx, dehalo_bases* bases);
size_t dehalo_bases_len(const dehalo_bases* bases);
/* The Pippenger window c chosen for these bases, the number of windows ceil(256 / c) and whether their multiples are stored. */
int dehalo_bases_info(const dehalo_bases* bases, uint32_t* window_bits, uint32_t* windows, int* precomputed);

/* ---- MSM == halo2_proofs::arithmetic::best_multiexp(coeffs, bases) -> C::Curve ------------
 * [halo2_proofs/src/arithmetic.rs; reached from ParamsKZG::commit / commit_lagrange].
 * out = sum_{i < len} scalars[i] * bases[i]; len may be smaller than the registered n
 * (a prefix is used, as commit() does).  scalars: len x 4 u64 (Montgomery).
 * out_jacobian: 12 u64.  upstream's `assert_eq!(coeffs.len(), bases.len())` becomes
 * DEHALO_ERR_INVALID when len > registered n.                                              */
int dehalo_msm(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* scalars, size_t len, uint64_t out_jacobian[12]);
/* `batch` independent MSMs over the same bases (one per committed column):
 * scalars[b] points to len x 4 u64; out_jacobian receives batch x 12 u64.                   */
int dehalo_msm_batch(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* const* scalars, size_t len, size_t batch,
                     uint64_t* out_jacobian);
/* Device-resident form: d_scalars = batch x len x 4 u64 contiguous in HBM, d_out = batch x 12 u64 in HBM. */
int dehalo_msm_device(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* d_scalars, size_t len, size_t batch,
                      uint64_t* d_out_jacobian, void* stream);
/* The same MSMs with the results ALSO (d_out_jacobian may be null: only) written as affine points {x, y} (64 B each, identity =
 * (0, 0), standard Montgomery form): what Blake2bWrite::write_point needs after ParamsKZG::commit_lagrange(..).to_affine()
 * [UPSTREAM halo2_proofs/src/plonk/prover.rs: `C::Curve::batch_normalize`] -- normalised by the kernel that finishes the MSM. */
int dehalo_msm_device_affine(dehalo_ctx* ctx, const dehalo_bases* bases, const uint64_t* d_scalars, size_t len, size_t batch, uint64_t* d_out_jacobian,
                             uint64_t* d_out_affine, void* stream);
/* One-shot best_multiexp over bases that are not registered (uploaded, used, dropped). */
int dehalo_best_multiexp(dehalo_ctx* ctx, int curve, const uint64_t* scalars, const uint64_t* affine_xy, size_t len,
                         uint64_t out_jacobian[12]);
/* G1::to_affine / batch_normalize of MSM outputs: count x 12 u64 -> count x 8 u64 (host buffers). */
int dehalo_to_affine(dehalo_ctx* ctx, int curve, const uint64_t* jacobian, size_t count, uint64_t* affine_xy);
/* Sum of `count` Jacobian points -> one Jacobian point (identity for count == 0): the combine step when ONE large MSM
 * is split by point range over several GPUs: each rank registers its slice of the bases, runs
 * dehalo_msm_device on its slice of the scalars, the 96-byte partial results are all-gathered (RCCL) and summed here.   */
int dehalo_point_sum_device(dehalo_ctx* ctx, int curve, const uint64_t* d_jacobian, size_t count, uint64_t* d_out_jacobian, void* stream);
/* device-pointer form: `count` Jacobian points (96 B each) -> affine (64 B each), asynchronous on the stream */
int dehalo_to_affine_device(dehalo_ctx* ctx, int curve, const uint64_t* d_jacobian, size_t count, uint64_t* d_affine_xy, void* stream);

/* ---- NTT == halo2_proofs::arithmetic::best_fft(a, omega, log_n) ---------------------------
 * [halo2_proofs/src/arithmetic.rs].  In place, natural order in and out,
 * a'[i] = sum_j a[j] * omega^(i*j), no scaling.  a: 2^log_n x 4 u64.  omega must be a
 * primitive 2^log_n-th root of unity (forward or inverse), Montgomery form.
 * upstream's `assert_eq!(a.len(), 1 << log_n)` is the caller's contract here.              */
int dehalo_ntt(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_n, const uint64_t omega[4]);
int dehalo_ntt_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_n, const uint64_t omega[4], size_t batch,
                      void* stream);

/* ---- EvaluationDomain conveniences [halo2_proofs/src/poly/domain.rs] ----------------------
 * lagrange_to_coeff:  best_fft(a, omega_inv, k); a[i] *= n_inv                              */
int dehalo_intt_scaled(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_n, const uint64_t omega_inv[4], const uint64_t n_inv[4]);
/* coeff_to_extended:  ext[i] = coeffs[i] * [1, zeta, zeta^2][i % 3] for i < 2^log_n, zero
 * above; best_fft(ext, omega_ext, log_ext).  coeffs and ext_out may not alias.              */
int dehalo_coset_ntt(dehalo_ctx* ctx, int field, const uint64_t* coeffs, uint32_t log_n, uint64_t* ext_out, uint32_t log_ext,
                     const uint64_t omega_ext[4], const uint64_t zeta[4]);
/* extended_to_coeff:  best_fft(a, omega_ext_inv, log_ext); a[i] *= ext_n_inv * [1, zeta^2, zeta][i % 3].
 * (The caller truncates to n * quotient_poly_degree, as upstream does.)                     */
int dehalo_coset_intt(dehalo_ctx* ctx, int field, uint64_t* a, uint32_t log_ext, const uint64_t omega_ext_inv[4],
                      const uint64_t ext_n_inv[4], const uint64_t zeta[4]);
/* Device-resident forms (batch polynomials contiguous in HBM). */
int dehalo_intt_scaled_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_n, const uint64_t omega_inv[4],
                              const uint64_t n_inv[4], size_t batch, void* stream);
/* The same out of place: d_coeffs = lagrange_to_coeff(d_values) for `batch` contiguous columns; d_values is left as it is (the prover keeps the
 * Lagrange values it has committed to and needs the coefficients beside them).  d_coeffs == d_values is allowed. */
int dehalo_lagrange_to_coeff_device(dehalo_ctx* ctx, int field, const uint64_t* d_values, uint64_t* d_coeffs, uint32_t log_n, const uint64_t omega_inv[4],
                                    const uint64_t n_inv[4], size_t batch, void* stream);
int dehalo_coset_ntt_device(dehalo_ctx* ctx, int field, const uint64_t* d_coeffs, uint32_t log_n, uint64_t* d_ext_out,
                            uint32_t log_ext, const uint64_t omega_ext[4], const uint64_t zeta[4], size_t batch, void* stream);
int dehalo_coset_intt_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_ext, const uint64_t omega_ext_inv[4],
                             const uint64_t ext_n_inv[4], const uint64_t zeta[4], size_t batch, void* stream);

/* ---- element-wise field ops (halo2curves Fr/Fq/Fp::{add,sub,mul,invert,to_repr,from_repr})
 * Used by the parity tests to pin the device field arithmetic directly (e.g. against the
 * reference's Poseidon known-answer vectors).  Host buffers, n elements each.
 * op: 0 add, 1 sub, 2 mul, 3 invert, 4 canonical->Montgomery, 5 Montgomery->canonical,
 *     6 mul evaluated through the kernels' internal carry-free 29-bit-limb representation.     */
int dehalo_field_op(dehalo_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* device-pointer form (d_b may be NULL for the unary ops; d_out may alias d_a): e.g. canonical -> Montgomery of a column uploaded as raw integers */
int dehalo_field_op_device(dehalo_ctx* ctx, int field, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n, void* stream);

/* ---- field-vector primitives around the path ---------------------------
 * Everything create_proof does between two commitments that is a scan over a column of field
 * elements, so that columns can stay in HBM from one MSM / NTT to the next.  Elements are
 * upstream's in-memory form (4 x u64 Montgomery limbs, canonical).
 *
 * dehalo_eval_polynomial: out = sum_i coeffs[i] * point^i; 0 for len == 0.  Replaces
 *   halo2_proofs::arithmetic::eval_polynomial (halo2_proofs/src/arithmetic.rs @ v2023_04_20;
 *   called for every opened polynomial in plonk/prover.rs after the challenge x is squeezed).
 *   The device form evaluates `batch` polynomials (stride_elems apart) at the same point into
 *   d_out[batch][4].
 * dehalo_batch_invert: values[i] <- values[i]^-1 in place, zero elements left zero.  Replaces
 *   ff::BatchInvert::batch_invert as used on the denominators of the permutation and lookup
 *   grand products (plonk/permutation/prover.rs, plonk/lookup/prover.rs).
 * dehalo_prefix_product_device: out[0] = 1, out[i] = in[0] * ... * in[i-1]  (in == out allowed).
 * dehalo_grand_product: z[0] = 1, z[i] = prod_{j<i} num[j] / den[j] -- the running product
 *   upstream builds after its batch_invert ("z.push(one); for row in 1..n { tmp *= product[row-1] }");
 *   the caller appends upstream's blinding rows.  A zero denominator behaves like upstream's
 *   batch_invert (left zero), so the product is zero from that row on.                          */
/* dehalo_permute_expression_pair: the lookup argument's permuted columns, replacing
 *   plonk/lookup/prover.rs permute_expression_pair: permuted_input = the `usable_rows` input values
 *   sorted ascending by canonical value; permuted_table[row] = permuted_input[row] at every first
 *   occurrence, the table's remaining values (ascending) in the repeated rows taken from the end.
 *   Only the table is sorted (merge sort on the 256-bit keys); every input value finds its table position by binary search and the
 *   outputs are written from the position counts (csrc/lookup_permute.hip).
 *   DEHALO_ERR_NOT_IN_TABLE when an input value does not occur in the table.  The caller appends
 *   upstream's random blinding rows.  The device form synchronises its stream (it has to return
 *   that error).                                                                                 */
int dehalo_permute_expression_pair(dehalo_ctx* ctx, int field, const uint64_t* input, const uint64_t* table, size_t usable_rows,
                                   uint64_t* permuted_input, uint64_t* permuted_table);
int dehalo_permute_expression_pair_device(dehalo_ctx* ctx, int field, const uint64_t* d_input, const uint64_t* d_table, size_t usable_rows,
                                          uint64_t* d_permuted_input, uint64_t* d_permuted_table, void* stream);
/* `batch` lookup arguments in one call (a proof's five): column y of each of the four arrays starts y * stride_elems elements in.
 * All key columns share the sort passes, so the cost barely grows with the batch.                                                */
int dehalo_permute_expression_pair_batch_device(dehalo_ctx* ctx, int field, const uint64_t* d_inputs, const uint64_t* d_tables, size_t usable_rows, size_t batch,
                                                size_t stride_elems, uint64_t* d_permuted_inputs, uint64_t* d_permuted_tables, void* stream);
/* The same with the columns given as HOST arrays of DEVICE pointers.  Lookups whose table pointers are EQUAL share one table sort (the five
 * range lookups of the reference's circuit compress the same (tag, value) table expressions: one sort per proof instead of five). */
int dehalo_permute_expression_pair_ptrs_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows, size_t batch,
                                               uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, void* stream);
/* The same without the synchronisation: d_status (DEVICE, `batch` int32) receives one flag per lookup -- non-zero: an input value is not in the table, the call
 * above would have returned DEHALO_ERR_NOT_IN_TABLE -- for the caller to read back with whatever it reads next (the prover: with the permuted columns' commitments),
 * so that the stream runs from the permutation straight into the commitment. */
int dehalo_permute_expression_pair_ptrs_deferred_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows,
                                                        size_t batch, uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, int32_t* d_status, void* stream);
/* The same for tables of FIXED columns (every table expression of the lookup reads fixed columns only -- the reference circuit's range tables): which rows of such
 * a table are equal does not depend on theta, only the order of the compressed values does.  Per lookup y the caller passes one representative row (an index
 * < usable_rows) of every distinct table tuple and how many of the usable rows hold that tuple (DEVICE arrays of distinct_count[y] uint32 each; the multiplicities add
 * up to usable_rows -- otherwise the outputs are unspecified, never an out-of-bounds access; lookups that share a table pass the same arrays).  Then only the distinct compressed values are sorted -- 339 instead of 131,066 for the
 * delay-encryption circuit at k = 17 -- and the sorted table is written out from (value, multiplicity): the same A' and S'.  distinct_count[y] == 0 or > 2048 for any
 * lookup of the call: the general path.  d_status as above, or null: the call synchronises and returns DEHALO_ERR_NOT_IN_TABLE itself. */
int dehalo_permute_expression_pair_distinct_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_inputs, const uint64_t* const* d_tables, size_t usable_rows, size_t batch,
                                                   uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables, const uint32_t* const* d_rep_rows,
                                                   const uint32_t* const* d_multiplicities, const uint32_t* distinct_count, int32_t* d_status, void* stream);
int dehalo_eval_polynomial(dehalo_ctx* ctx, int field, const uint64_t* coeffs, size_t len, const uint64_t point[4], uint64_t out[4]);
int dehalo_eval_polynomial_device(dehalo_ctx* ctx, int field, const uint64_t* d_coeffs, size_t len, size_t stride_elems, size_t batch,
                                  const uint64_t point[4], uint64_t* d_out, void* stream);
/* `count` polynomials (d_polys: HOST array of DEVICE pointers, len coefficients each) at 1..4 points in one pass over the coefficients
 * (create_proof opens its columns at x, omega x, omega^-1 x and omega^last x): points = num_points x 4 u64 (host),
 * d_out[point][polynomial][4]. */
int dehalo_eval_polynomial_multi_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_polys, size_t count, size_t len, const uint64_t* points,
                                        uint32_t num_points, uint64_t* d_out, void* stream);
/* The same for a caller that needs only some (polynomial, point) pairs: wanted[j] (host, one byte per polynomial; NULL = everything) has bit i set when polynomial j
 * is wanted at point i.  The other entries of d_out are written as zero and cost nothing (a proof asks for about 60 of its 47 x 4 values). */
int dehalo_eval_polynomial_multi_masked_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_polys, size_t count, size_t len, const uint64_t* points,
                                               uint32_t num_points, const uint8_t* wanted, uint64_t* d_out, void* stream);
int dehalo_batch_invert(dehalo_ctx* ctx, int field, uint64_t* values, size_t len);
int dehalo_batch_invert_device(dehalo_ctx* ctx, int field, uint64_t* d_values, size_t len, void* stream);
int dehalo_prefix_product_device(dehalo_ctx* ctx, int field, const uint64_t* d_in, size_t len, uint64_t* d_out, void* stream);
int dehalo_grand_product(dehalo_ctx* ctx, int field, const uint64_t* num, const uint64_t* den, size_t len, uint64_t* z);
int dehalo_grand_product_device(dehalo_ctx* ctx, int field, const uint64_t* d_num, const uint64_t* d_den, size_t len, uint64_t* d_z, void* stream);
/* `batch` columns, stride_elems apart in num, den and z alike: the denominators of the whole batch share one inversion
 * (a proof's 2 permutation + 5 lookup products are one call).                                                            */
int dehalo_grand_product_batch_device(dehalo_ctx* ctx, int field, const uint64_t* d_num, const uint64_t* d_den, size_t len, size_t batch,
                                      size_t stride_elems, uint64_t* d_z, void* stream);

/* ---- the remaining element-wise steps of create_proof (so that a proof's columns never leave HBM) ----------
 * dehalo_lincomb_device: out[i] = sum_j coefs[j] * cols[j][i]; sub_const (may be NULL) is subtracted from out[0].
 *   d_cols = HOST array of `count` DEVICE column pointers (len elements each), coefs = host, count x 4 u64.
 *   Replaces the polynomial folds of plonk/prover.rs / vanishing::Constructed::evaluate (h pieces with x^n) and
 *   poly/kzg/multiopen/gwc/prover.rs ("poly_batch = sum v^i poly_i; poly_batch - eval_batch").  out may alias no column.
 * dehalo_scale_device: a[i] *= pattern[i mod period] (host pattern, period in {0 (none), 1, 2, 4, 8}) and, if
 *   d_factor != NULL, *= *d_factor (one element in DEVICE memory).  Replaces EvaluationDomain::divide_by_vanishing_poly
 *   (poly/domain.rs: t_evaluations has 2^(extended_k - k) entries) and the "z = vec![last_z]" carry between the
 *   permutation argument's product columns (plonk/permutation/prover.rs).
 * dehalo_kate_division: q = (a(X) - a(point)) / (X - point), len - 1 coefficients; replaces
 *   halo2_proofs::arithmetic::kate_division (called once per opening point by ProverGWC::create_proof).          */
int dehalo_lincomb_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_cols, const uint64_t* coefs, size_t count, size_t len, uint64_t* d_out,
                          const uint64_t* sub_const, void* stream);
int dehalo_scale_device(dehalo_ctx* ctx, int field, uint64_t* d_a, size_t len, const uint64_t* pattern, uint32_t period, const uint64_t* d_factor, void* stream);
int dehalo_kate_division(dehalo_ctx* ctx, int field, const uint64_t* a, size_t len, const uint64_t point[4], uint64_t* q);
int dehalo_kate_division_device(dehalo_ctx* ctx, int field, const uint64_t* d_a, size_t len, const uint64_t point[4], uint64_t* d_q, void* stream);
/* up to 8 divisions of equal length (<= 2^22 coefficients) in one set of launches: d_a / d_q = HOST arrays of DEVICE pointers,
 * points = count x 4 u64 (ProverGWC::create_proof: one division per distinct opening point) */
int dehalo_kate_division_batch_device(dehalo_ctx* ctx, int field, const uint64_t* const* d_a, size_t len, const uint64_t* points, uint64_t* const* d_q,
                                      size_t count, void* stream);

/* ---- quotient numerator: evaluate_h --------------------------------------
 * The row loops of halo2_proofs/src/plonk/evaluation.rs @ v2023_04_20 on device-resident
 * extended-domain columns (each `rows = 1 << log_rows` elements, standard Montgomery form), so the
 * cosets produced by dehalo_coset_ntt_device never leave HBM before dehalo_coset_intt_device.
 *
 * A dehalo_graph is upstream's GraphEvaluator { constants, rotations, calculations,
 * num_intermediates } handed over verbatim: ValueSource -> dehalo_source (kind, index, rotation
 * = index into rotations[]), Calculation -> dehalo_calculation (Horner(start, parts, factor):
 * a = start, b = factor, parts = horner_parts[parts_begin .. parts_begin + parts_len]),
 * CalculationInfo.target -> target.  dehalo_graph_evaluate_device computes, for every row idx,
 *   out[idx] = GraphEvaluator::evaluate(.., previous_value = previous[idx] (0 if NULL), idx, rot_scale, isize = rows)
 * i.e. the value of the LAST calculation (0 for an empty graph); a column source reads
 * column[(idx + rotations[rotation] * rot_scale) mod rows].  previous == out is allowed (custom
 * gates: values[idx] = evaluate(.., values[idx], ..)).
 *
 * dehalo_permutation_h_device / dehalo_lookup_h_device fold upstream's hard-coded argument terms
 * into `values` in place (value = value * y + term, in upstream's order):
 *   permutation: l0 (1 - z_0); l_last (z_last^2 - z_last); for sets s >= 1: l0 (z_s - z_{s-1}(w^last X));
 *                for every set: l_active (z_s(wX) prod_j (col_j + beta sigma_j + gamma)
 *                                         - z_s(X) prod_j (col_j + delta^j beta X + gamma)),
 *                X = zeta * extended_omega^idx, delta^j running across the sets' column chunks;
 *   lookup:      l0 (1 - z); l_last (z^2 - z); l_active (z(wX)(a' + beta)(s' + gamma) - z(X) table_value);
 *                l0 (a' - s'); l_active (a' - s')(a' - a'(w^-1 X)),
 *                table_value = the lookup's GraphEvaluator output for the row.                    */
/* Optional device-internal element form.  Every `_device` entry point above takes and returns
 * upstream's standard form; converting it costs the evaluate_h kernels one multiplication per
 * column load.  Columns that only travel between this library's kernels can stay in the kernels'
 * own form instead (x * 2^261 mod p, canonical, 32 B -- opaque to the caller):
 *   dehalo_convert_form_device             standard <-> internal, element-wise (proving-key columns, once);
 *   dehalo_coset_ntt_form_device           coeff_to_extended emitting the internal form (DEHALO_FORM_OUT_INTERNAL);
 *   dehalo_coset_intt_form_device          extended_to_coeff consuming it (DEHALO_FORM_IN_INTERNAL);
 *   form_flags of the three *_inputs structs below: DEHALO_EVAL_COLUMNS_INTERNAL (every column pointer of the
 *   struct), DEHALO_EVAL_VALUES_INTERNAL (d_previous / d_out / d_values / table_value).              */
enum { DEHALO_FORM_OUT_INTERNAL = 1, DEHALO_FORM_IN_INTERNAL = 2 };
enum { DEHALO_EVAL_COLUMNS_INTERNAL = 1, DEHALO_EVAL_VALUES_INTERNAL = 2 };
int dehalo_convert_form_device(dehalo_ctx* ctx, int field, const uint64_t* d_in, uint64_t* d_out, size_t n, int to_internal, void* stream);
int dehalo_coset_ntt_form_device(dehalo_ctx* ctx, int field, const uint64_t* d_coeffs, uint32_t log_n, uint64_t* d_ext_out, uint32_t log_ext,
                                 const uint64_t omega_ext[4], const uint64_t zeta[4], size_t batch, uint32_t form_flags, void* stream);
int dehalo_coset_intt_form_device(dehalo_ctx* ctx, int field, uint64_t* d_a, uint32_t log_ext, const uint64_t omega_ext_inv[4],
                                  const uint64_t ext_n_inv[4], const uint64_t zeta[4], size_t batch, uint32_t form_flags, void* stream);

typedef enum {
    DEHALO_SRC_CONSTANT = 0, DEHALO_SRC_INTERMEDIATE = 1, DEHALO_SRC_FIXED = 2, DEHALO_SRC_ADVICE = 3, DEHALO_SRC_INSTANCE = 4,
    DEHALO_SRC_CHALLENGE = 5, DEHALO_SRC_BETA = 6, DEHALO_SRC_GAMMA = 7, DEHALO_SRC_THETA = 8, DEHALO_SRC_Y = 9, DEHALO_SRC_PREVIOUS = 10
} dehalo_source_kind;
typedef struct { uint32_t kind, index, rotation; } dehalo_source;
typedef enum {
    DEHALO_CALC_ADD = 0, DEHALO_CALC_SUB = 1, DEHALO_CALC_MUL = 2, DEHALO_CALC_SQUARE = 3, DEHALO_CALC_DOUBLE = 4, DEHALO_CALC_NEGATE = 5,
    DEHALO_CALC_HORNER = 6, DEHALO_CALC_STORE = 7
} dehalo_calc_op;
typedef struct {
    uint32_t op;
    dehalo_source a, b;
    uint32_t parts_begin, parts_len;
    uint32_t target;
} dehalo_calculation;
typedef struct dehalo_graph dehalo_graph;
int dehalo_graph_create(dehalo_ctx* ctx, int field, const uint64_t* constants, uint32_t num_constants, const int32_t* rotations, uint32_t num_rotations,
                        const dehalo_calculation* calcs, uint32_t num_calcs, const dehalo_source* horner_parts, uint32_t num_horner_parts,
                        uint32_t num_intermediates, dehalo_graph** out);
int dehalo_graph_release(dehalo_ctx* ctx, dehalo_graph* graph);
typedef struct {
    const uint64_t* const* fixed; uint32_t num_fixed;        /* host arrays of DEVICE column pointers */
    const uint64_t* const* advice; uint32_t num_advice;
    const uint64_t* const* instance; uint32_t num_instance;
    const uint64_t* challenges; uint32_t num_challenges;     /* host, 4 u64 each */
    const uint64_t *beta, *gamma, *theta, *y;                /* host, 4 u64 each; NULL = 0 */
    uint32_t form_flags;                                     /* 0, or DEHALO_EVAL_* */
} dehalo_eval_inputs;
int dehalo_graph_evaluate_device(dehalo_ctx* ctx, const dehalo_graph* graph, const dehalo_eval_inputs* in, uint32_t log_rows, uint32_t rot_scale,
                                 const uint64_t* d_previous, uint64_t* d_out, void* stream);
/* `count` programs over the SAME inputs (no previous value), program i into d_outs[i]: the theta-compressed input and table expressions
 * of every lookup [UPSTREAM plonk/lookup/prover.rs commit_permuted: compress_expressions], one call from the host instead of `count`. */
int dehalo_graph_evaluate_batch_device(dehalo_ctx* ctx, const dehalo_graph* const* graphs, uint32_t count, const dehalo_eval_inputs* in, uint32_t log_rows,
                                       uint32_t rot_scale, uint64_t* const* d_outs, void* stream);
typedef struct {
    const uint64_t* const* z; uint32_t num_sets;             /* permutation_product_coset per set (device pointers) */
    const uint64_t* const* columns; const uint64_t* const* sigma; uint32_t num_columns;   /* column cosets / pk.permutation.cosets */
    uint32_t chunk_len;                                      /* cs.degree() - 2 */
    int32_t last_rotation;                                   /* -(blinding_factors + 1) */
    const uint64_t *l0, *l_last, *l_active_row;              /* device */
    const uint64_t *beta, *gamma, *y, *delta, *beta_zeta, *extended_omega;   /* host, 4 u64 each */
    uint32_t form_flags;                                     /* 0, or DEHALO_EVAL_* */
} dehalo_perm_inputs;
int dehalo_permutation_h_device(dehalo_ctx* ctx, int field, const dehalo_perm_inputs* in, uint32_t log_rows, uint32_t rot_scale, uint64_t* d_values,
                                void* stream);
typedef struct {
    const uint64_t *product_coset, *permuted_input_coset, *permuted_table_coset, *table_value;   /* device */
    const uint64_t *l0, *l_last, *l_active_row;              /* device */
    const uint64_t *beta, *gamma, *y;                        /* host */
    uint32_t form_flags;                                     /* 0, or DEHALO_EVAL_*; table_value counts as a value */
} dehalo_lookup_inputs;
int dehalo_lookup_h_device(dehalo_ctx* ctx, int field, const dehalo_lookup_inputs* in, uint32_t log_rows, uint32_t rot_scale, uint64_t* d_values,
                           void* stream);
/* The same terms for `count` <= 8 lookups (in[0], in[1], ... in upstream's order: evaluate_h's `for lookup in lookups` loop) in ONE pass
 * over the rows: d_values and the three Lagrange columns are read once instead of once per lookup.  The lookups of one call share
 * l0 / l_last / l_active_row, beta / gamma / y and form_flags (checked); each brings its own cosets and table_value column. */
int dehalo_lookup_h_batch_device(dehalo_ctx* ctx, int field, const dehalo_lookup_inputs* in, uint32_t count, uint32_t log_rows, uint32_t rot_scale,
                                 uint64_t* d_values, void* stream);

/* ==== the whole call: halo2_proofs::plonk::{keygen_vk, keygen_pk, create_proof} over KZG / GWC =====================================
 * The reference's timed call is
 *     create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC<_>, Challenge255<_>, _, Blake2bWrite<_, _, _>, _>(&params, &pk, &[circuit],
 *                                                                                                          &[&[&[]]], &mut OsRng, &mut transcript)
 * (benches/delay_enc.rs:123-131, mod_pow.rs:201-209, pose_enc.rs:127-135) over objects it builds once and caches on disk (params
 * :41-54, vk / pk :84-115).  The entry points below are that call and those objects, one C function each: every column stays in HBM
 * from the witness upload to the last opening; the host hashes the transcript (Blake2b-512) and orders the phases, in C++.
 * What arrives from the caller is what halo2 holds at that point: the constraint system as data, the fixed columns and the permutation
 * assembly keygen produced, the advice columns `synthesize` filled (witness generation stays the front-end's).
 */
typedef struct dehalo_params dehalo_params;         /* ParamsKZG<E>: k, g, g_lagrange (resident MSM tables), g2, s_g2 */
typedef struct dehalo_pk dehalo_pk;                 /* ProvingKey<C> incl. its VerifyingKey and the compiled GraphEvaluator programs */
typedef struct dehalo_prover dehalo_prover;         /* device buffers of ONE proof in flight (reused by every create_proof on it) */
typedef struct dehalo_transcript dehalo_transcript; /* Blake2bWrite<Vec<u8>, C, Challenge255<C>> */

/* ParamsKZG [UPSTREAM halo2_proofs/src/poly/kzg/commitment.rs].  g / g_lagrange: 2^k affine points each ({x, y} Montgomery, 64 B);
 * g2 / s_g2: 128 raw bytes each (kept for write(); the prover never uses them).  read / write: SerdeFormat::RawBytes --
 * k: u32 LE | g | g_lagrange | g2 | s_g2  (what benches/delay_enc.rs:45,54 write and read). */
int dehalo_params_create(dehalo_ctx* ctx, int curve, uint32_t k, const uint64_t* g, const uint64_t* g_lagrange, const uint8_t* g2, const uint8_t* s_g2,
                         dehalo_params** out);
/* ParamsKZG::setup(k, rng) with the rng's draw handed over: `s` is the toxic waste as a Montgomery scalar (what `<E::Scalar>::random(rng)` returns).  g[i] = [s^i] G
 * and g_lagrange[i] = [L_i(s)] G are made on the device (fixed-base table multiplication, one batch inversion), g2 / s_g2 on the host: the reference's
 * `ParamsKZG::<Bn256>::setup(K, OsRng)` (benches/delay_enc.rs:43).  BN254 only (a KZG SRS needs the pairing); k <= 25 (checked before any device work: k = 26, whose two tables
 * would take 128 GB, is DEHALO_ERR_INVALID). */
int dehalo_params_setup(dehalo_ctx* ctx, int curve, uint32_t k, const uint64_t s[4], dehalo_params** out);
int dehalo_params_read(dehalo_ctx* ctx, int curve, const uint8_t* bytes, size_t len, dehalo_params** out);
size_t dehalo_params_size(const dehalo_params* params);
int dehalo_params_write(const dehalo_params* params, uint8_t* out, size_t cap);
int dehalo_params_release(dehalo_ctx* ctx, dehalo_params* params);
/* ParamsKZG::{commit, commit_lagrange} of `batch` device-resident columns of 2^k elements (n apart) -> affine points in HBM. */
int dehalo_params_commit_device(dehalo_ctx* ctx, const dehalo_params* params, const uint64_t* d_polys, size_t batch, int lagrange, uint64_t* d_out_affine, void* stream);

/* The circuit as data: the fields of halo2_proofs::plonk::ConstraintSystem that keygen and the prover read
 * [UPSTREAM halo2_proofs/src/plonk/circuit.rs].  Expressions are a flat array of nodes, children before parents:
 *   CONSTANT  a = index into `constants`;     FIXED / ADVICE / INSTANCE  a = column index, rotation;
 *   NEGATED   a = child;  SUM / PRODUCT  a, b = children;  SCALED  a = child, b = index into `constants`.                          */
typedef enum {
    DEHALO_EXPR_CONSTANT = 0, DEHALO_EXPR_FIXED = 1, DEHALO_EXPR_ADVICE = 2, DEHALO_EXPR_INSTANCE = 3, DEHALO_EXPR_NEGATED = 4, DEHALO_EXPR_SUM = 5,
    DEHALO_EXPR_PRODUCT = 6, DEHALO_EXPR_SCALED = 7
} dehalo_expr_kind;
typedef struct { uint32_t kind, a, b; int32_t rotation; } dehalo_expr_node;
typedef enum { DEHALO_COLUMN_ADVICE = 0, DEHALO_COLUMN_FIXED = 1, DEHALO_COLUMN_INSTANCE = 2 } dehalo_column_kind;
typedef struct { uint32_t kind, index; int32_t rotation; } dehalo_column_query;          /* (Column, Rotation); rotation unused for permutation columns */
typedef struct {
    uint32_t num_advice, num_fixed, num_instance, minimum_degree;
    const dehalo_expr_node* nodes; uint32_t num_nodes;
    const uint64_t* constants; uint32_t num_constants;              /* 4 x u64 Montgomery each (Expression::Constant(F)) */
    const uint32_t* gates; uint32_t num_gates;                      /* root node of every gate polynomial, in order (gates.flat_map(polynomials)) */
    const uint32_t* lookup_lens; uint32_t num_lookups;              /* lookup l has lookup_lens[l] (input, table) expression pairs ...          */
    const uint32_t* lookup_inputs; const uint32_t* lookup_tables;   /* ... their root nodes, lookups concatenated                                */
    const dehalo_column_query* permutation_columns; uint32_t num_permutation_columns;      /* cs.permutation.columns, enable_equality order */
    const dehalo_column_query* advice_queries; uint32_t num_advice_queries;                /* first-use order: the order of the evaluations in the proof */
    const dehalo_column_query* fixed_queries; uint32_t num_fixed_queries;
    const dehalo_column_query* instance_queries; uint32_t num_instance_queries;
} dehalo_constraint_system;

/* keygen_vk + keygen_pk [UPSTREAM halo2_proofs/src/plonk/keygen.rs; benches/delay_enc.rs:86,103] on the device: commit_lagrange,
 * lagrange_to_coeff and coeff_to_extended of every fixed and permutation column, l0 / l_last / l_active_row.
 *   fixed               num_fixed x 2^k x 4 u64, host; Montgomery unless DEHALO_KEYGEN_FIXED_CANONICAL (selectors already compressed into them)
 *   permutation_mapping num_permutation_columns x 2^k u64, host: cell (column j, row i) is mapped to cell value / 2^k, value % 2^k
 *                       (permutation::keygen::Assembly::mapping flattened); identity = j * 2^k + i
 *   selectors           num_selectors arrays of 2^k bytes (0 / 1): only serialised with the verifying key (vk.selectors)              */
enum { DEHALO_KEYGEN_FIXED_CANONICAL = 1 };
int dehalo_keygen(dehalo_ctx* ctx, const dehalo_params* params, const dehalo_constraint_system* cs, const uint64_t* fixed, const uint64_t* permutation_mapping,
                  const uint8_t* const* selectors, uint32_t num_selectors, uint32_t flags, dehalo_pk** out);
/* ProvingKey::{read, write} / VerifyingKey::write with SerdeFormat::RawBytes [UPSTREAM halo2_proofs/src/plonk.rs; benches/delay_enc.rs:
 * 88-115]: reading needs the circuit's constraint system, as upstream's read::<_, ConcreteCircuit> does. */
int dehalo_pk_read(dehalo_ctx* ctx, int curve, const dehalo_constraint_system* cs, const uint8_t* bytes, size_t len, uint32_t num_selectors, dehalo_pk** out);
size_t dehalo_pk_size(const dehalo_pk* pk);
int dehalo_pk_write(dehalo_ctx* ctx, const dehalo_pk* pk, uint8_t* out, size_t cap);
size_t dehalo_vk_size(const dehalo_pk* pk);
int dehalo_vk_write(const dehalo_pk* pk, uint8_t* out, size_t cap);
/* vk.transcript_repr, the first thing create_proof absorbs.  Upstream derives it from the Debug text of the pinned verifying key, which
 * cannot be reproduced outside the crate: an integrator passes upstream's value (4 x u64 Montgomery).  Until set, the key carries a
 * substitute: Blake2b-512("Halo2-Verify-Key") over the key's RawBytes and a binary encoding of the constraint system. */
int dehalo_pk_set_transcript_repr(dehalo_pk* pk, const uint64_t repr[4]);
int dehalo_pk_get_transcript_repr(const dehalo_pk* pk, uint64_t repr[4]);
/* out[0..7] = k, extended_k, blinding_factors, degree, permutation sets, commitments before the evaluations, evaluations, opening points */
int dehalo_pk_info(const dehalo_pk* pk, uint32_t out[8]);
int dehalo_pk_release(dehalo_ctx* ctx, dehalo_pk* pk);

/* Random scalars of one proof (see csrc/hostrng.hpp).  DEHALO_RNG_OS is what a NULL pointer selects. */
typedef enum { DEHALO_RNG_OS = 0, DEHALO_RNG_PCG64 = 1, DEHALO_RNG_CALLBACK = 2 } dehalo_rng_kind;
/* fills `count` scalars (4 x u64 Montgomery representations, < p); `position` = index of the first one in upstream's draw order.  May be
 * called from a helper thread of the library, never concurrently for one proof.  Returns 0. */
typedef int (*dehalo_rng_fill_fn)(void* user, uint64_t* out, size_t count, uint64_t position);
typedef struct {
    int kind;
    uint64_t pcg_state[2], pcg_inc[2];     /* DEHALO_RNG_PCG64: numpy PCG64 state / increment, low word first; advanced past the proof's draws on return */
    dehalo_rng_fill_fn fill; void* user;   /* DEHALO_RNG_CALLBACK */
} dehalo_rng;
/* `count` scalars of the field from a generator (what create_proof draws): lets a test pin the three kinds without a device. */
int dehalo_rng_scalars(dehalo_rng* rng, int field, uint64_t skip, uint64_t* out, size_t count);

/* Blake2bWrite / Challenge255 [UPSTREAM halo2_proofs/src/transcript.rs; benches/delay_enc.rs:120,134]. */
int dehalo_transcript_create(int curve, dehalo_transcript** out);
int dehalo_transcript_common_scalar(dehalo_transcript* t, const uint64_t scalar[4]);                 /* Montgomery */
int dehalo_transcript_write_scalar(dehalo_transcript* t, const uint64_t scalar[4]);
int dehalo_transcript_write_point(dehalo_transcript* t, const uint64_t affine_xy[8]);                /* Montgomery; identity rejected */
int dehalo_transcript_squeeze_challenge(dehalo_transcript* t, uint64_t out[4]);                      /* Montgomery */
size_t dehalo_transcript_len(const dehalo_transcript* t);
int dehalo_transcript_finalize(const dehalo_transcript* t, uint8_t* out, size_t cap);                /* the proof bytes written so far */
void dehalo_transcript_release(dehalo_transcript* t);

/* A prover = the device buffers of one proof for a given key, on `ctx` (stream + workspace).  `side_ctx` (may be NULL): a second context of
 * the same device; work no transcript challenge waits for (lagrange_to_coeff / coeff_to_extended of a phase's columns, the random
 * polynomial's commitment, the gate and table-value passes of evaluate_h) is queued there and runs beside the commitment phases.
 * Several provers over one key, each on its own context(s), may run concurrently from different threads (batch proving). */
int dehalo_prover_create(dehalo_ctx* ctx, dehalo_ctx* side_ctx, const dehalo_params* params, const dehalo_pk* pk, dehalo_prover** out);
int dehalo_prover_release(dehalo_prover* prover);
/* create_proof: one circuit, its instance columns, the caller's rng and transcript.
 *   advice     num_advice x 2^k x 4 u64 Montgomery (rows >= usable are overwritten by blinding), host memory, or device memory when
 *              DEHALO_PROOF_ADVICE_ON_DEVICE
 *   instances  num_instance_columns arrays of instance_lens[i] scalars (Montgomery); the reference passes none (&[&[&[]]])
 * Errors follow upstream's: DEHALO_ERR_INVALID for instances.len() != num_instance_columns / an instance column longer than the usable
 * rows / a commitment at infinity; DEHALO_ERR_NOT_IN_TABLE when a lookup input is missing from its table. */
enum { DEHALO_PROOF_ADVICE_ON_DEVICE = 1, DEHALO_PROOF_ADVICE_CANONICAL = 2 /* plain integers < p: converted on the device */ };
int dehalo_create_proof(dehalo_prover* prover, const uint64_t* advice, const uint64_t* const* instances, const size_t* instance_lens, uint32_t num_instance_columns,
                        dehalo_rng* rng, dehalo_transcript* transcript, uint32_t flags);
/* ONE proof on several GPUs ("single-proof mode -- round-robin columns of the current phase across GPUs"): every one of `world` processes runs
 * the same dehalo_create_proof on its own GPU with the same inputs and the same seeded random stream (DEHALO_RNG_PCG64, or a callback that hands every process the
 * same scalars); of every commitment phase with more than one column (advice, permuted lookup columns, the grand products, the quotient's pieces, the openings) a
 * process runs the MSM of columns [count * rank / world, count * (rank + 1) / world) only, and `gather` -- called on the proving thread with `points` = count affine
 * points (8 u64 each, Montgomery) of which this process's own slots [first[rank], first[rank] + num[rank]) are filled -- must fill in the others (an all-gather of
 * num[r] points from process r, e.g. ncclAllGather over xGMI: at most ten points a phase) and return 0.  Every process then writes the same points to its transcript,
 * squeezes the same challenges and ends with the same proof bytes as a lone prover.  Everything that is not an MSM is computed by every process (no column travels).
 * world = 1 (or gather = NULL with world = 1) switches it off. */
typedef int (*dehalo_gather_fn)(void* user, uint64_t* points, uint32_t count, const uint32_t* first, const uint32_t* num, uint32_t world);
int dehalo_prover_set_shard(dehalo_prover* prover, uint32_t rank, uint32_t world, dehalo_gather_fn gather, void* user);
/* Host wall-clock milliseconds the last create_proof on this prover spent per phase (advice, lookups, products, random, quotient, evaluations,
 * openings, total): out[8]. */
int dehalo_prover_last_timings(const dehalo_prover* prover, double out[8]);
/* `count` proofs of the same circuit on `num_provers` provers, one host thread each (proof i -> prover i mod num_provers), no interpreter
 * in the loop: advice[i] / rngs[i] / proofs_out[i] (capacity proof_cap bytes each, length into proof_lens[i]) per proof; every proof starts
 * a fresh transcript.  Batch / throughput mode (BASELINE configs[4]). */
int dehalo_create_proofs(dehalo_prover* const* provers, uint32_t num_provers, const uint64_t* const* advice, uint32_t count, dehalo_rng* rngs, uint32_t flags,
                         uint8_t* const* proofs_out, size_t proof_cap, size_t* proof_lens);

/* ---- Circuit::synthesize of the reference's circuits, as values (csrc/witness.hip) --------------------------------------------------
 * What the reference's timed create_proof does BEFORE its first commitment: DelayEncryptCircuit::synthesize (src/lib.rs:164-318: RSA
 * time-lock x^e mod n over BigIntChip rows, Poseidon sponge of the packed result, Poseidon cipher of the message under the hash's two
 * outputs), benches/mod_pow.rs:63-110's RSACircuit (the RSA region alone) and PoseidonEncCircuit (src/encryption/chip.rs:114-204, the
 * cipher region alone) -- laid out over MainGate + RangeChip instruction by instruction the way halo2wrong lays them out (row counts: the ones the reference
 * publishes, benches/README.md:56-99; cell for cell identity with upstream's synthesize is unverified -- the crates are not in the tree), so the columns belong
 * with a key made by dehalo_keygen from THIS call's fixed columns and mapping.  The reference's circuit is satisfiable only for x < n, e < 2^exp_bits and the
 * all-zero message its benches encrypt (its in-circuit cipher adds the message twice, its native one never to the state it permutes): other inputs are refused.
 *   advice     out, 5 x 2^k x 4 u64, CANONICAL values (pass DEHALO_PROOF_ADVICE_CANONICAL to dehalo_create_proof), rows beyond the
 *              circuit zero; NULL = not wanted
 *   fixed      out, (15 | 9 for pose_enc) x 2^k x 4 u64 canonical (DEHALO_KEYGEN_FIXED_CANONICAL), range table included; NULL = not wanted
 *   mapping    out, 6 x 2^k u64: the permutation assembly for dehalo_keygen; NULL = not wanted
 *   selectors  out, two arrays of 2^k bytes (s_composition, s_overflow; none for pose_enc); NULL = not wanted
 * Per proof only `advice` is needed; the other three are keygen's.  DEHALO_ERR_INVALID when the circuit does not fit 2^k rows or a constraint cannot hold. */
typedef enum { DEHALO_CIRCUIT_DELAY_ENC = 0, DEHALO_CIRCUIT_MOD_POW = 1, DEHALO_CIRCUIT_POSE_ENC = 2 } dehalo_circuit_kind;
typedef struct {
    uint32_t circuit, k;
    uint32_t bits_len, exp_bits;        /* RSA: modulus bits (the reference's BITS_LEN = 2048; any multiple of 64 -- the RangeChip table follows compute_range_lens(bits_len / 64)), exponent bits (<= 64) */
    const uint64_t *n, *x;              /* modulus and base, bits_len / 64 limbs, little-endian */
    uint64_t e;                         /* exponent */
    const uint64_t* message; uint32_t message_len;   /* <= 2 canonical field elements (4 u64 each) */
    const uint64_t* key;                /* pose_enc: 2 canonical field elements */
    uint32_t t, rate, r_f, r_p;         /* Poseidon; 0 = the reference's (5, 4, 8, 57) */
} dehalo_circuit_inputs;
typedef struct {
    uint64_t rsa_rows, total_rows;
    uint64_t rsa_result[128];           /* x^e mod n, little-endian limbs */
    uint64_t cipher[12]; uint32_t cipher_len;   /* the ciphertext (canonical), message_len + 1 elements */
} dehalo_synthesis_info;
int dehalo_synthesize(const dehalo_circuit_inputs* in, uint64_t* advice, uint64_t* fixed, uint64_t* mapping, uint8_t* const* selectors, dehalo_synthesis_info* info);
/* create_proof of a CIRCUIT -- the call the reference makes (benches/delay_enc.rs:123-131, benches/mod_pow.rs:201-209, benches/pose_enc.rs:127-135:
 * create_proof(&params, &pk, &[circuit], instances, rng, &mut transcript) synthesizes the circuit inside): dehalo_synthesize into a page-locked buffer the
 * prover keeps, upload, and the proof; the random polynomial's draw, upload and commitment (side context) run on the otherwise idle device meanwhile.
 * Same proof bytes as dehalo_synthesize + dehalo_create_proof with DEHALO_PROOF_ADVICE_CANONICAL.  `info` (optional): rows, x^e mod n, ciphertext. */
int dehalo_create_proof_circuit(dehalo_prover* prover, const dehalo_circuit_inputs* in, dehalo_synthesis_info* info, const uint64_t* const* instances,
                                const size_t* instance_lens, uint32_t num_instance_columns, dehalo_rng* rng, dehalo_transcript* transcript);
/* dehalo_create_proofs with every proof's circuit synthesized inside its call: inputs[i] -> proof i on prover i mod num_provers, one library thread per
 * prover (witness generation of one proof runs beside the device work of the others). */
int dehalo_create_proofs_circuit(dehalo_prover* const* provers, uint32_t num_provers, const dehalo_circuit_inputs* inputs, uint32_t count, dehalo_rng* rngs,
                                 uint8_t* const* proofs_out, size_t proof_cap, size_t* proof_lens);

/* The grand products' per-row factors, every product of a proof in one launch [UPSTREAM plonk/permutation/prover.rs Argument::commit:
 * per set of columns  den = prod_j (value_j + beta sigma_j + gamma),  num = prod_j (value_j + delta^j beta omega^i + gamma);
 * plonk/lookup/prover.rs Permuted::commit_product:  den = (a' + beta)(s' + gamma),  num = (A + beta)(S + gamma)] over the `n` rows of the
 * ORIGINAL domain.  Product p (permutation sets first, then lookups) is written to d_num / d_den + p * stride_elems; all columns and
 * results in upstream's standard form: the inputs of dehalo_grand_product_batch_device. */
typedef struct {
    const uint64_t* const* columns; const uint64_t* const* sigma; uint32_t num_columns, chunk_len;   /* the permutation's columns (values) and sigma_j values */
    const uint64_t* omega_powers;                    /* device column: omega^i */
    const uint64_t *beta, *gamma, *delta;            /* host, 4 u64 each */
    const uint64_t* set_factors;                     /* host, one per set: beta * delta^(chunk_len * set) */
    const uint64_t* const* compressed_input; const uint64_t* const* compressed_table;     /* per lookup: A, S (theta-compressed) ... */
    const uint64_t* const* permuted_input; const uint64_t* const* permuted_table;         /* ... a', s' */
    uint32_t num_lookups;
} dehalo_product_inputs;
int dehalo_product_terms_device(dehalo_ctx* ctx, int field, const dehalo_product_inputs* in, size_t n, uint64_t* d_num, uint64_t* d_den, size_t stride_elems, void* stream);

/* ---- measurement ---------------------------------------------------------------------------
 * Per-kernel device time measured with HIP events on the launching stream (bench.py's
 * roofline leg).  kernel ids: see dehalo_kernel_id.                                          */
typedef enum {
    DEHALO_K_MSM_ACCUMULATE = 0, /* bucket accumulation (the dominant MSM kernel) */
    DEHALO_K_MSM_SORT = 1,       /* digit histogram + scatter                       */
    DEHALO_K_MSM_REDUCE = 2,     /* partial merge + bucket reduction                */
    DEHALO_K_NTT_PASS = 3,       /* all NTT passes of one transform                 */
    DEHALO_K_POLY = 4,           /* eval_polynomial / batch_invert / prefix product */
    DEHALO_K_EVAL_H = 5,         /* quotient-numerator kernels (graph, permutation, lookup) */
    DEHALO_K_COUNT = 6
} dehalo_kernel_id;
int dehalo_timing_enable(dehalo_ctx* ctx, int on);
int dehalo_timing_reset(dehalo_ctx* ctx);
/* Synchronises the context, then returns total milliseconds and number of timed regions. */
int dehalo_timing_get(dehalo_ctx* ctx, int kernel_id, double* total_ms, uint64_t* count);

/* Shape of the context's most recent MSM launch, for tuning (synchronises): out[0..3] = buckets whose partial sums were merged by one
 * lane or quad / 32 lanes / one wave / a whole block, out[4] = points per lane of the accumulation, out[5] = (bucket, point) pairs sorted. */
int dehalo_msm_last_shape(dehalo_ctx* ctx, uint32_t out[6]);

/* The host-side constants of a field as the library computed them (4 x u64 Montgomery each): out[0] = modulus (plain integer), out[1] = R mod p (one),
 * out[2] = ROOT_OF_UNITY, out[3] = ZETA, out[4] = DELTA, out[5] = MULTIPLICATIVE_GENERATOR [UPSTREAM halo2curves / pasta_curves PrimeField, WithSmallOrderMulGroup]. */
int dehalo_field_info(int field, uint64_t out[24]);

/* Library / build info, e.g. "dehalo 0.1 gfx950". */
const char* dehalo_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DEHALO_H */
