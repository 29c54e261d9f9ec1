"""ctypes binding of libdehalo.so (C ABI: include/dehalo.h).  Fails loudly when the HIP
library is missing or no gfx950 device is present -- there is no fallback."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Tuple, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SYMBOLS = [
    "dehalo_version", "dehalo_ctx_create", "dehalo_ctx_create_with_priority", "dehalo_ctx_destroy", "dehalo_last_error", "dehalo_ctx_synchronize", "dehalo_download", "dehalo_upload", "dehalo_ctx_stream", "dehalo_ctx_set_tuning",
    "dehalo_bases_register", "dehalo_bases_release", "dehalo_bases_len", "dehalo_bases_info",
    "dehalo_msm", "dehalo_msm_batch", "dehalo_msm_device", "dehalo_msm_device_affine", "dehalo_msm_last_shape", "dehalo_lookup_h_batch_device", "dehalo_product_terms_device", "dehalo_best_multiexp", "dehalo_to_affine", "dehalo_to_affine_device", "dehalo_point_sum_device",
    "dehalo_ntt", "dehalo_ntt_device", "dehalo_intt_scaled", "dehalo_coset_ntt", "dehalo_coset_intt",
    "dehalo_intt_scaled_device", "dehalo_lagrange_to_coeff_device", "dehalo_coset_ntt_device", "dehalo_coset_intt_device",
    "dehalo_field_op", "dehalo_field_op_device", "dehalo_timing_enable", "dehalo_timing_reset", "dehalo_timing_get",
    "dehalo_eval_polynomial", "dehalo_eval_polynomial_device", "dehalo_eval_polynomial_multi_device", "dehalo_eval_polynomial_multi_masked_device", "dehalo_batch_invert", "dehalo_batch_invert_device",
    "dehalo_prefix_product_device", "dehalo_grand_product", "dehalo_grand_product_device", "dehalo_grand_product_batch_device",
    "dehalo_permute_expression_pair", "dehalo_permute_expression_pair_device", "dehalo_permute_expression_pair_batch_device", "dehalo_permute_expression_pair_ptrs_device", "dehalo_permute_expression_pair_distinct_device", "dehalo_permute_expression_pair_ptrs_deferred_device",
    "dehalo_convert_form_device", "dehalo_coset_ntt_form_device", "dehalo_coset_intt_form_device",
    "dehalo_lincomb_device", "dehalo_scale_device", "dehalo_kate_division", "dehalo_kate_division_device", "dehalo_kate_division_batch_device",
    "dehalo_params_create", "dehalo_params_setup", "dehalo_bases_register_device", "dehalo_params_read", "dehalo_params_size", "dehalo_params_write", "dehalo_params_release", "dehalo_params_commit_device",
    "dehalo_create_proof_circuit", "dehalo_create_proofs_circuit", "dehalo_keygen", "dehalo_pk_read", "dehalo_pk_size", "dehalo_pk_write", "dehalo_vk_size", "dehalo_vk_write", "dehalo_pk_set_transcript_repr",
    "dehalo_pk_get_transcript_repr", "dehalo_pk_info", "dehalo_pk_release", "dehalo_rng_scalars", "dehalo_field_info", "dehalo_synthesize",
    "dehalo_transcript_create", "dehalo_transcript_common_scalar", "dehalo_transcript_write_scalar", "dehalo_transcript_write_point",
    "dehalo_transcript_squeeze_challenge", "dehalo_transcript_len", "dehalo_transcript_finalize", "dehalo_transcript_release",
    "dehalo_prover_create", "dehalo_prover_release", "dehalo_create_proof", "dehalo_prover_set_shard", "dehalo_prover_last_timings", "dehalo_create_proofs",
    "dehalo_graph_create", "dehalo_graph_release", "dehalo_graph_evaluate_device", "dehalo_graph_evaluate_batch_device", "dehalo_permutation_h_device", "dehalo_lookup_h_device",
]

K_MSM_ACCUMULATE, K_MSM_SORT, K_MSM_REDUCE, K_NTT_PASS, K_POLY, K_EVAL_H = 0, 1, 2, 3, 4, 5


class CSource(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("rotation", C.c_uint32)]


class CCalculation(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", CSource), ("b", CSource), ("parts_begin", C.c_uint32), ("parts_len", C.c_uint32), ("target", C.c_uint32)]


class CEvalInputs(C.Structure):
    _fields_ = [("fixed", C.POINTER(C.c_void_p)), ("num_fixed", C.c_uint32), ("advice", C.POINTER(C.c_void_p)), ("num_advice", C.c_uint32),
                ("instance", C.POINTER(C.c_void_p)), ("num_instance", C.c_uint32), ("challenges", C.c_void_p), ("num_challenges", C.c_uint32),
                ("beta", C.c_void_p), ("gamma", C.c_void_p), ("theta", C.c_void_p), ("y", C.c_void_p), ("form_flags", C.c_uint32)]


class CPermInputs(C.Structure):
    _fields_ = [("z", C.POINTER(C.c_void_p)), ("num_sets", C.c_uint32), ("columns", C.POINTER(C.c_void_p)), ("sigma", C.POINTER(C.c_void_p)),
                ("num_columns", C.c_uint32), ("chunk_len", C.c_uint32), ("last_rotation", C.c_int32),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p),
                ("beta", C.c_void_p), ("gamma", C.c_void_p), ("y", C.c_void_p), ("delta", C.c_void_p), ("beta_zeta", C.c_void_p), ("extended_omega", C.c_void_p),
                ("form_flags", C.c_uint32)]


class CLookupInputs(C.Structure):
    _fields_ = [("product_coset", C.c_void_p), ("permuted_input_coset", C.c_void_p), ("permuted_table_coset", C.c_void_p), ("table_value", C.c_void_p),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p), ("beta", C.c_void_p), ("gamma", C.c_void_p), ("y", C.c_void_p),
                ("form_flags", C.c_uint32)]


class CExprNode(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("rotation", C.c_int32)]


class CColumnQuery(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("rotation", C.c_int32)]


class CConstraintSystem(C.Structure):
    _fields_ = [("num_advice", C.c_uint32), ("num_fixed", C.c_uint32), ("num_instance", C.c_uint32), ("minimum_degree", C.c_uint32),
                ("nodes", C.POINTER(CExprNode)), ("num_nodes", C.c_uint32), ("constants", C.c_void_p), ("num_constants", C.c_uint32),
                ("gates", C.POINTER(C.c_uint32)), ("num_gates", C.c_uint32), ("lookup_lens", C.POINTER(C.c_uint32)), ("num_lookups", C.c_uint32),
                ("lookup_inputs", C.POINTER(C.c_uint32)), ("lookup_tables", C.POINTER(C.c_uint32)),
                ("permutation_columns", C.POINTER(CColumnQuery)), ("num_permutation_columns", C.c_uint32),
                ("advice_queries", C.POINTER(CColumnQuery)), ("num_advice_queries", C.c_uint32),
                ("fixed_queries", C.POINTER(CColumnQuery)), ("num_fixed_queries", C.c_uint32),
                ("instance_queries", C.POINTER(CColumnQuery)), ("num_instance_queries", C.c_uint32)]


class CCircuitInputs(C.Structure):
    _fields_ = [("circuit", C.c_uint32), ("k", C.c_uint32), ("bits_len", C.c_uint32), ("exp_bits", C.c_uint32), ("n", C.c_void_p), ("x", C.c_void_p), ("e", C.c_uint64),
                ("message", C.c_void_p), ("message_len", C.c_uint32), ("key", C.c_void_p), ("t", C.c_uint32), ("rate", C.c_uint32), ("r_f", C.c_uint32), ("r_p", C.c_uint32)]


class CSynthesisInfo(C.Structure):
    _fields_ = [("rsa_rows", C.c_uint64), ("total_rows", C.c_uint64), ("rsa_result", C.c_uint64 * 128), ("cipher", C.c_uint64 * 12), ("cipher_len", C.c_uint32)]


RNG_FILL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32)      # dehalo_gather_fn


class CRng(C.Structure):
    _fields_ = [("kind", C.c_int), ("pcg_state", C.c_uint64 * 2), ("pcg_inc", C.c_uint64 * 2), ("fill", RNG_FILL_FN), ("user", C.c_void_p)]


class CProductInputs(C.Structure):
    _fields_ = [("columns", C.POINTER(C.c_void_p)), ("sigma", C.POINTER(C.c_void_p)), ("num_columns", C.c_uint32), ("chunk_len", C.c_uint32), ("omega_powers", C.c_void_p),
                ("beta", C.c_void_p), ("gamma", C.c_void_p), ("delta", C.c_void_p), ("set_factors", C.c_void_p), ("compressed_input", C.POINTER(C.c_void_p)),
                ("compressed_table", C.POINTER(C.c_void_p)), ("permuted_input", C.POINTER(C.c_void_p)), ("permuted_table", C.POINTER(C.c_void_p)), ("num_lookups", C.c_uint32)]


class DehaloError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("dehalo error %d: %s" % (code, msg))
        self.code = code


def library_path() -> str:
    """DEHALO_LIBRARY overrides the in-tree build (A/B runs of two builds in one session)."""
    return os.environ.get("DEHALO_LIBRARY") or os.path.join(_HERE, "libdehalo.so")


def load_library():
    """Loads libdehalo.so.  If torch is importable it is imported FIRST so that the process
    holds a single HIP runtime (torch bundles libamdhip64.so.7; same soname as /opt/rocm's)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise DehaloError(-2, "libdehalo.so is not built (run `make` or __graft_entry__.build()); there is no CPU fallback")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    P, sz, u32, u64p = C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p
    lib.dehalo_version.restype = C.c_char_p
    lib.dehalo_last_error.restype = C.c_char_p
    lib.dehalo_last_error.argtypes = [P]
    lib.dehalo_ctx_create.argtypes = [C.c_int, C.POINTER(P)]
    lib.dehalo_ctx_create_with_priority.argtypes = [C.c_int, C.c_int, C.POINTER(P)]
    lib.dehalo_ctx_destroy.argtypes = [P]
    lib.dehalo_ctx_destroy.restype = None
    lib.dehalo_ctx_synchronize.argtypes = [P]
    lib.dehalo_download.argtypes = [P, P, C.c_size_t, P]
    lib.dehalo_upload.argtypes = [P, P, C.c_size_t, P]
    lib.dehalo_ctx_set_tuning.argtypes = [P, C.c_char_p, C.c_int]
    lib.dehalo_ctx_stream.argtypes = [P]
    lib.dehalo_ctx_stream.restype = C.c_void_p
    lib.dehalo_bases_register.argtypes = [P, C.c_int, u64p, sz, sz, C.c_int, C.c_int, C.POINTER(P)]
    lib.dehalo_bases_release.argtypes = [P, P]
    lib.dehalo_bases_len.argtypes = [P]
    lib.dehalo_bases_len.restype = sz
    lib.dehalo_bases_info.argtypes = [P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    lib.dehalo_msm.argtypes = [P, P, u64p, sz, u64p]
    lib.dehalo_msm_batch.argtypes = [P, P, C.POINTER(C.c_void_p), sz, sz, u64p]
    lib.dehalo_msm_device.argtypes = [P, P, u64p, sz, sz, u64p, P]
    lib.dehalo_msm_device_affine.argtypes = [P, P, u64p, sz, sz, u64p, u64p, P]
    lib.dehalo_msm_last_shape.argtypes = [P, C.POINTER(C.c_uint32)]
    lib.dehalo_best_multiexp.argtypes = [P, C.c_int, u64p, u64p, sz, u64p]
    lib.dehalo_to_affine.argtypes = [P, C.c_int, u64p, sz, u64p]
    lib.dehalo_to_affine_device.argtypes = [P, C.c_int, u64p, sz, u64p, P]
    lib.dehalo_point_sum_device.argtypes = [P, C.c_int, u64p, sz, u64p, P]
    lib.dehalo_ntt.argtypes = [P, C.c_int, u64p, u32, u64p]
    lib.dehalo_ntt_device.argtypes = [P, C.c_int, u64p, u32, u64p, sz, P]
    lib.dehalo_intt_scaled.argtypes = [P, C.c_int, u64p, u32, u64p, u64p]
    lib.dehalo_coset_ntt.argtypes = [P, C.c_int, u64p, u32, u64p, u32, u64p, u64p]
    lib.dehalo_coset_intt.argtypes = [P, C.c_int, u64p, u32, u64p, u64p, u64p]
    lib.dehalo_intt_scaled_device.argtypes = [P, C.c_int, u64p, u32, u64p, u64p, sz, P]
    if hasattr(lib, "dehalo_lagrange_to_coeff_device"):
        lib.dehalo_lagrange_to_coeff_device.argtypes = [P, C.c_int, u64p, u64p, u32, u64p, u64p, sz, P]
    lib.dehalo_coset_ntt_device.argtypes = [P, C.c_int, u64p, u32, u64p, u32, u64p, u64p, sz, P]
    lib.dehalo_coset_intt_device.argtypes = [P, C.c_int, u64p, u32, u64p, u64p, u64p, sz, P]
    lib.dehalo_field_op.argtypes = [P, C.c_int, C.c_int, u64p, u64p, u64p, sz]
    lib.dehalo_field_op_device.argtypes = [P, C.c_int, C.c_int, u64p, u64p, u64p, sz, P]
    lib.dehalo_eval_polynomial.argtypes = [P, C.c_int, u64p, sz, u64p, u64p]
    lib.dehalo_eval_polynomial_device.argtypes = [P, C.c_int, u64p, sz, sz, sz, u64p, u64p, P]
    lib.dehalo_eval_polynomial_multi_device.argtypes = [P, C.c_int, C.POINTER(C.c_void_p), sz, sz, u64p, u32, u64p, P]
    if hasattr(lib, "dehalo_eval_polynomial_multi_masked_device"):
        lib.dehalo_eval_polynomial_multi_masked_device.argtypes = [P, C.c_int, C.POINTER(C.c_void_p), sz, sz, u64p, u32, C.c_char_p, u64p, P]
    lib.dehalo_batch_invert.argtypes = [P, C.c_int, u64p, sz]
    lib.dehalo_batch_invert_device.argtypes = [P, C.c_int, u64p, sz, P]
    lib.dehalo_prefix_product_device.argtypes = [P, C.c_int, u64p, sz, u64p, P]
    lib.dehalo_grand_product.argtypes = [P, C.c_int, u64p, u64p, sz, u64p]
    lib.dehalo_grand_product_device.argtypes = [P, C.c_int, u64p, u64p, sz, u64p, P]
    lib.dehalo_grand_product_batch_device.argtypes = [P, C.c_int, u64p, u64p, sz, sz, sz, u64p, P]
    lib.dehalo_permute_expression_pair.argtypes = [P, C.c_int, u64p, u64p, sz, u64p, u64p]
    lib.dehalo_permute_expression_pair_device.argtypes = [P, C.c_int, u64p, u64p, sz, u64p, u64p, P]
    lib.dehalo_permute_expression_pair_batch_device.argtypes = [P, C.c_int, u64p, u64p, sz, sz, sz, u64p, u64p, P]
    if hasattr(lib, "dehalo_permute_expression_pair_ptrs_deferred_device"):
        lib.dehalo_permute_expression_pair_ptrs_deferred_device.argtypes = [P, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), sz, sz, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), P, P]
    if hasattr(lib, "dehalo_permute_expression_pair_distinct_device"):
        lib.dehalo_permute_expression_pair_distinct_device.argtypes = [P, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), sz, sz, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), P, P]
    if hasattr(lib, "dehalo_permute_expression_pair_ptrs_device"):      # (absent from an older build loaded through DEHALO_LIBRARY)
        lib.dehalo_permute_expression_pair_ptrs_device.argtypes = [P, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), sz, sz, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), P]
    lib.dehalo_lincomb_device.argtypes = [P, C.c_int, C.POINTER(C.c_void_p), u64p, sz, sz, u64p, u64p, P]
    lib.dehalo_scale_device.argtypes = [P, C.c_int, u64p, sz, u64p, u32, u64p, P]
    lib.dehalo_kate_division.argtypes = [P, C.c_int, u64p, sz, u64p, u64p]
    lib.dehalo_kate_division_device.argtypes = [P, C.c_int, u64p, sz, u64p, u64p, P]
    lib.dehalo_kate_division_batch_device.argtypes = [P, C.c_int, C.POINTER(C.c_void_p), sz, u64p, C.POINTER(C.c_void_p), sz, P]
    lib.dehalo_convert_form_device.argtypes = [P, C.c_int, u64p, u64p, sz, C.c_int, P]
    lib.dehalo_coset_ntt_form_device.argtypes = [P, C.c_int, u64p, u32, u64p, u32, u64p, u64p, sz, u32, P]
    lib.dehalo_coset_intt_form_device.argtypes = [P, C.c_int, u64p, u32, u64p, u64p, u64p, sz, u32, P]
    lib.dehalo_graph_create.argtypes = [P, C.c_int, u64p, u32, C.POINTER(C.c_int32), u32, C.POINTER(CCalculation), u32, C.POINTER(CSource), u32, u32, C.POINTER(P)]
    lib.dehalo_graph_release.argtypes = [P, P]
    lib.dehalo_graph_evaluate_device.argtypes = [P, P, C.POINTER(CEvalInputs), u32, u32, u64p, u64p, P]
    lib.dehalo_graph_evaluate_batch_device.argtypes = [P, C.POINTER(C.c_void_p), u32, C.POINTER(CEvalInputs), u32, u32, C.POINTER(C.c_void_p), P]
    lib.dehalo_permutation_h_device.argtypes = [P, C.c_int, C.POINTER(CPermInputs), u32, u32, u64p, P]
    lib.dehalo_lookup_h_device.argtypes = [P, C.c_int, C.POINTER(CLookupInputs), u32, u32, u64p, P]
    if hasattr(lib, "dehalo_product_terms_device"):
        lib.dehalo_product_terms_device.argtypes = [P, C.c_int, C.POINTER(CProductInputs), sz, u64p, u64p, sz, P]
    lib.dehalo_lookup_h_batch_device.argtypes = [P, C.c_int, C.POINTER(CLookupInputs), u32, u32, u32, u64p, P]
    PP = C.POINTER(P)
    lib.dehalo_params_create.argtypes = [P, C.c_int, u32, u64p, u64p, P, P, PP]
    lib.dehalo_params_setup.argtypes = [P, C.c_int, u32, u64p, PP]
    lib.dehalo_bases_register_device.argtypes = [P, C.c_int, u64p, C.c_size_t, C.c_int, C.c_int, PP]
    lib.dehalo_params_read.argtypes = [P, C.c_int, P, sz, PP]
    lib.dehalo_params_size.argtypes = [P]
    lib.dehalo_params_size.restype = sz
    lib.dehalo_params_write.argtypes = [P, P, sz]
    lib.dehalo_params_release.argtypes = [P, P]
    lib.dehalo_params_commit_device.argtypes = [P, P, u64p, sz, C.c_int, u64p, P]
    lib.dehalo_keygen.argtypes = [P, P, C.POINTER(CConstraintSystem), u64p, u64p, C.POINTER(C.c_void_p), u32, u32, PP]
    lib.dehalo_pk_read.argtypes = [P, C.c_int, C.POINTER(CConstraintSystem), P, sz, u32, PP]
    lib.dehalo_pk_size.argtypes = [P]
    lib.dehalo_pk_size.restype = sz
    lib.dehalo_pk_write.argtypes = [P, P, P, sz]
    lib.dehalo_vk_size.argtypes = [P]
    lib.dehalo_vk_size.restype = sz
    lib.dehalo_vk_write.argtypes = [P, P, sz]
    lib.dehalo_pk_set_transcript_repr.argtypes = [P, u64p]
    lib.dehalo_pk_get_transcript_repr.argtypes = [P, u64p]
    lib.dehalo_pk_info.argtypes = [P, C.POINTER(C.c_uint32)]
    lib.dehalo_pk_release.argtypes = [P, P]
    lib.dehalo_synthesize.argtypes = [C.POINTER(CCircuitInputs), u64p, u64p, u64p, C.POINTER(C.c_void_p), C.POINTER(CSynthesisInfo)]
    lib.dehalo_field_info.argtypes = [C.c_int, u64p]
    lib.dehalo_rng_scalars.argtypes = [C.POINTER(CRng), C.c_int, C.c_uint64, u64p, sz]
    lib.dehalo_transcript_create.argtypes = [C.c_int, PP]
    lib.dehalo_transcript_common_scalar.argtypes = [P, u64p]
    lib.dehalo_transcript_write_scalar.argtypes = [P, u64p]
    lib.dehalo_transcript_write_point.argtypes = [P, u64p]
    lib.dehalo_transcript_squeeze_challenge.argtypes = [P, u64p]
    lib.dehalo_transcript_len.argtypes = [P]
    lib.dehalo_transcript_len.restype = sz
    lib.dehalo_transcript_finalize.argtypes = [P, P, sz]
    lib.dehalo_transcript_release.argtypes = [P]
    lib.dehalo_transcript_release.restype = None
    lib.dehalo_prover_create.argtypes = [P, P, P, P, PP]
    lib.dehalo_prover_release.argtypes = [P]
    lib.dehalo_create_proof.argtypes = [P, u64p, C.POINTER(C.c_void_p), C.POINTER(sz), u32, C.POINTER(CRng), P, u32]
    lib.dehalo_create_proof_circuit.argtypes = [P, C.POINTER(CCircuitInputs), C.POINTER(CSynthesisInfo), C.POINTER(C.c_void_p), C.POINTER(sz), u32, C.POINTER(CRng), P]
    lib.dehalo_create_proofs_circuit.argtypes = [C.POINTER(C.c_void_p), u32, C.POINTER(CCircuitInputs), u32, C.POINTER(CRng), C.POINTER(C.c_void_p), sz, C.POINTER(sz)]
    lib.dehalo_prover_last_timings.argtypes = [P, C.POINTER(C.c_double)]
    lib.dehalo_prover_set_shard.argtypes = [P, C.c_uint32, C.c_uint32, GATHER_FN, P]
    lib.dehalo_create_proofs.argtypes = [C.POINTER(C.c_void_p), u32, C.POINTER(C.c_void_p), u32, C.POINTER(CRng), u32, C.POINTER(C.c_void_p), sz, C.POINTER(sz)]
    lib.dehalo_timing_enable.argtypes = [P, C.c_int]
    lib.dehalo_timing_reset.argtypes = [P]
    lib.dehalo_timing_get.argtypes = [P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    _LIB = lib
    return lib


def _ptr(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "expected contiguous uint64 array"
    return a.ctypes.data_as(C.c_void_p)


def _u64(a, shape_last: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, shape_last)


class Bases:
    """Device-resident SRS (ParamsKZG.g / g_lagrange)."""

    def __init__(self, ctx: "Context", handle, curve: int, n: int):
        self.ctx, self.handle, self.curve, self.n = ctx, handle, curve, n
        c, w, pre = C.c_uint32(), C.c_uint32(), C.c_int()
        ctx._check(ctx.lib.dehalo_bases_info(handle, C.byref(c), C.byref(w), C.byref(pre)))
        self.window_bits, self.windows, self.precomputed = c.value, w.value, bool(pre.value)

    def release(self):
        if self.handle is not None:
            self.ctx._check(self.ctx.lib.dehalo_bases_release(self.ctx.handle, self.handle))
            self.handle = None

    def __len__(self):
        return self.n


class Context:
    default_tuning: dict = {}

    def __init__(self, device: int = 0, priority: int = 0):
        """priority > 0: highest stream priority of the device (short kernels beside another context's long ones), < 0: lowest."""
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.dehalo_ctx_create_with_priority(device, priority, C.byref(h))
        if rc != 0:
            raise DehaloError(rc, "dehalo_ctx_create failed (no gfx950 device?); there is no CPU fallback")
        self.handle = h
        self.device = device
        self._tables = {}
        for key, value in Context.default_tuning.items():      # launch-geometry knobs every new context starts with (bench.py's A/B flags)
            self.set_tuning(key, value)

    def close(self):
        if self.handle is not None:
            self.lib.dehalo_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise DehaloError(rc, self.lib.dehalo_last_error(self.handle).decode())

    def synchronize(self):
        self._check(self.lib.dehalo_ctx_synchronize(self.handle))

    def download(self, d_src: int, count: int, width: int = 4) -> np.ndarray:
        """(count, width) u64 from device memory, after everything queued on the context's stream (one call: copy + wait)."""
        out = np.empty((count, width), dtype=np.uint64)
        self._check(self.lib.dehalo_download(self.handle, d_src, out.nbytes, out.ctypes.data))
        return out

    def upload(self, host, dtype=None):
        """A host array (any numpy array of 64-bit / 32-bit words or bytes) -> a torch tensor of the same shape in HBM, through dehalo_upload: the library's page-locked staging
        chunks or a DMA from pages the library pins for the call -- never torch's / the HIP runtime's handling of a pageable source (`tensor.cuda()`)."""
        import torch

        a = np.ascontiguousarray(host)
        assert a.dtype.itemsize in (8, 4, 1), "upload: 64-bit words (field elements, indices), 32-bit words or bytes"
        if dtype is None:
            dtype = {8: torch.int64, 4: torch.int32, 1: torch.uint8}[a.dtype.itemsize]
        out = torch.empty(a.shape, dtype=dtype, device=torch.device("cuda", self.device))      # the CONTEXT's device, not torch's current one
        self._check(self.lib.dehalo_upload(self.handle, a.ctypes.data, a.nbytes, out.data_ptr()))
        return out

    def download_tensor(self, t) -> np.ndarray:
        """A contiguous device tensor of 64-bit words -> numpy (uint64, same shape), through dehalo_download (staged like upload)."""
        assert t.is_contiguous() and t.element_size() == 8
        out = np.empty(tuple(t.shape), dtype=np.uint64)
        self._check(self.lib.dehalo_download(self.handle, t.data_ptr(), out.nbytes, out.ctypes.data))
        return out

    def set_tuning(self, key: str, value: int):
        self._check(self.lib.dehalo_ctx_set_tuning(self.handle, key.encode(), value))

    def torch_stream_obj(self):
        """The context's stream as a torch.cuda.ExternalStream (for events: record on one context, wait on another)."""
        self.torch_stream()
        return self._tstream

    def torch_stream(self):
        """The context's stream as a torch stream: `with torch.cuda.stream(ctx.torch_stream()):` puts torch's own copies and
        fills on the stream the library's kernels run on, so they are ordered with them (the context's stream is
        non-blocking: nothing orders it with torch's default stream)."""
        import torch

        if getattr(self, "_tstream", None) is None:
            self._tstream = torch.cuda.ExternalStream(self.lib.dehalo_ctx_stream(self.handle))
        return torch.cuda.stream(self._tstream)

    # ---- bases / MSM ----
    def register_bases(self, curve: int, affine_xy, window_bits: int = 0, precompute: bool = True) -> Bases:
        a = _u64(affine_xy, 8)
        h = C.c_void_p()
        self._check(self.lib.dehalo_bases_register(self.handle, curve, _ptr(a), a.shape[0], 64, window_bits, int(precompute), C.byref(h)))
        return Bases(self, h, curve, a.shape[0])

    def msm(self, bases: Bases, scalars) -> np.ndarray:
        s = _u64(scalars, 4)
        out = np.zeros(12, dtype=np.uint64)
        self._check(self.lib.dehalo_msm(self.handle, bases.handle, _ptr(s), s.shape[0], _ptr(out)))
        return out

    def msm_batch(self, bases: Bases, columns: Sequence) -> np.ndarray:
        cols = [_u64(c, 4) for c in columns]
        n = cols[0].shape[0] if cols else 0
        assert all(c.shape[0] == n for c in cols)
        ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        out = np.zeros((len(cols), 12), dtype=np.uint64)
        self._check(self.lib.dehalo_msm_batch(self.handle, bases.handle, ptrs, n, len(cols), _ptr(out)))
        return out

    def msm_device(self, bases: Bases, d_scalars: int, length: int, batch: int, d_out: int, stream: int = 0):
        self._check(self.lib.dehalo_msm_device(self.handle, bases.handle, d_scalars, length, batch, d_out, stream or None))

    def msm_device_affine(self, bases: Bases, d_scalars: int, length: int, batch: int, d_out_jacobian: int, d_out_affine: int, stream: int = 0):
        """msm_device with the results (also) as affine points, normalised by the kernel that finishes the MSM (d_out_jacobian may be 0)."""
        self._check(self.lib.dehalo_msm_device_affine(self.handle, bases.handle, d_scalars, length, batch, d_out_jacobian or None, d_out_affine, stream or None))

    def msm_last_shape(self) -> dict:
        out = (C.c_uint32 * 6)()
        self._check(self.lib.dehalo_msm_last_shape(self.handle, out))
        return {"merge_light": out[0], "merge_32": out[1], "merge_wave": out[2], "merge_block": out[3], "points_per_lane": out[4], "pairs": out[5]}

    def best_multiexp(self, curve: int, scalars, affine_xy) -> np.ndarray:
        s, a = _u64(scalars, 4), _u64(affine_xy, 8)
        if s.shape[0] != a.shape[0]:
            raise ValueError("best_multiexp: coeffs.len() != bases.len()")  # upstream assert_eq!
        out = np.zeros(12, dtype=np.uint64)
        self._check(self.lib.dehalo_best_multiexp(self.handle, curve, _ptr(s), _ptr(a), s.shape[0], _ptr(out)))
        return out

    def to_affine(self, curve: int, jacobian) -> np.ndarray:
        j = _u64(jacobian, 12)
        out = np.zeros((j.shape[0], 8), dtype=np.uint64)
        self._check(self.lib.dehalo_to_affine(self.handle, curve, _ptr(j), j.shape[0], _ptr(out)))
        return out

    def point_sum_device(self, curve: int, d_jacobian: int, count: int, d_out: int, stream: int = 0):
        self._check(self.lib.dehalo_point_sum_device(self.handle, curve, d_jacobian, count, d_out, stream or None))

    def to_affine_device(self, curve: int, d_jacobian: int, count: int, d_affine: int, stream: int = 0):
        self._check(self.lib.dehalo_to_affine_device(self.handle, curve, d_jacobian, count, d_affine, stream or None))

    # ---- NTT family (host buffers; in place on a copy, returned) ----
    def ntt(self, field: int, a, log_n: int, omega) -> np.ndarray:
        a = np.array(a, dtype=np.uint64).reshape(-1, 4)
        if a.shape[0] != 1 << log_n:
            raise ValueError("best_fft: a.len() != 1 << log_n")  # upstream assert_eq!
        self._check(self.lib.dehalo_ntt(self.handle, field, _ptr(a), log_n, _ptr(_u64(omega, 4))))
        return a

    def intt_scaled(self, field: int, a, log_n: int, omega_inv, n_inv) -> np.ndarray:
        a = np.array(a, dtype=np.uint64).reshape(-1, 4)
        if a.shape[0] != 1 << log_n:
            raise ValueError("a.len() != 1 << log_n")
        self._check(self.lib.dehalo_intt_scaled(self.handle, field, _ptr(a), log_n, _ptr(_u64(omega_inv, 4)), _ptr(_u64(n_inv, 4))))
        return a

    def coset_ntt(self, field: int, coeffs, log_n: int, log_ext: int, omega_ext, zeta) -> np.ndarray:
        c = _u64(coeffs, 4)
        if c.shape[0] != 1 << log_n:
            raise ValueError("coeffs.len() != 1 << log_n")
        out = np.zeros((1 << log_ext, 4), dtype=np.uint64)
        self._check(self.lib.dehalo_coset_ntt(self.handle, field, _ptr(c), log_n, _ptr(out), log_ext, _ptr(_u64(omega_ext, 4)), _ptr(_u64(zeta, 4))))
        return out

    def coset_intt(self, field: int, a, log_ext: int, omega_ext_inv, ext_n_inv, zeta) -> np.ndarray:
        a = np.array(a, dtype=np.uint64).reshape(-1, 4)
        if a.shape[0] != 1 << log_ext:
            raise ValueError("a.len() != 1 << log_ext")
        self._check(self.lib.dehalo_coset_intt(self.handle, field, _ptr(a), log_ext, _ptr(_u64(omega_ext_inv, 4)), _ptr(_u64(ext_n_inv, 4)),
                                               _ptr(_u64(zeta, 4))))
        return a

    # device-resident forms: raw device pointers (e.g. torch tensor .data_ptr()) and a hipStream_t
    def ntt_device(self, field: int, d_a: int, log_n: int, omega, batch: int = 1, stream: int = 0):
        self._check(self.lib.dehalo_ntt_device(self.handle, field, d_a, log_n, _ptr(_u64(omega, 4)), batch, stream or None))

    def intt_scaled_device(self, field: int, d_a: int, log_n: int, omega_inv, n_inv, batch: int = 1, stream: int = 0):
        self._check(self.lib.dehalo_intt_scaled_device(self.handle, field, d_a, log_n, _ptr(_u64(omega_inv, 4)), _ptr(_u64(n_inv, 4)), batch, stream or None))

    def lagrange_to_coeff_device(self, field: int, d_values: int, d_coeffs: int, log_n: int, omega_inv, n_inv, batch: int = 1, stream: int = 0):
        self._check(self.lib.dehalo_lagrange_to_coeff_device(self.handle, field, d_values, d_coeffs, log_n, _ptr(_u64(omega_inv, 4)), _ptr(_u64(n_inv, 4)), batch, stream or None))

    def coset_ntt_device(self, field: int, d_coeffs: int, log_n: int, d_ext: int, log_ext: int, omega_ext, zeta, batch: int = 1, stream: int = 0):
        self._check(self.lib.dehalo_coset_ntt_device(self.handle, field, d_coeffs, log_n, d_ext, log_ext, _ptr(_u64(omega_ext, 4)), _ptr(_u64(zeta, 4)), batch,
                                                     stream or None))

    # optional device-internal element form (dehalo.h): FORM_OUT_INTERNAL = 1, FORM_IN_INTERNAL = 2
    def convert_form_device(self, field: int, d_in: int, d_out: int, n: int, to_internal: bool, stream: int = 0):
        self._check(self.lib.dehalo_convert_form_device(self.handle, field, d_in, d_out, n, int(to_internal), stream or None))

    def coset_ntt_form_device(self, field: int, d_coeffs: int, log_n: int, d_ext: int, log_ext: int, omega_ext, zeta, batch: int, form_flags: int, stream: int = 0):
        self._check(self.lib.dehalo_coset_ntt_form_device(self.handle, field, d_coeffs, log_n, d_ext, log_ext, _ptr(_u64(omega_ext, 4)), _ptr(_u64(zeta, 4)), batch,
                                                          form_flags, stream or None))

    def coset_intt_form_device(self, field: int, d_a: int, log_ext: int, omega_ext_inv, ext_n_inv, zeta, batch: int, form_flags: int, stream: int = 0):
        self._check(self.lib.dehalo_coset_intt_form_device(self.handle, field, d_a, log_ext, _ptr(_u64(omega_ext_inv, 4)), _ptr(_u64(ext_n_inv, 4)),
                                                           _ptr(_u64(zeta, 4)), batch, form_flags, stream or None))

    def coset_intt_device(self, field: int, d_a: int, log_ext: int, omega_ext_inv, ext_n_inv, zeta, batch: int = 1, stream: int = 0):
        self._check(self.lib.dehalo_coset_intt_device(self.handle, field, d_a, log_ext, _ptr(_u64(omega_ext_inv, 4)), _ptr(_u64(ext_n_inv, 4)),
                                                      _ptr(_u64(zeta, 4)), batch, stream or None))

    # ---- element-wise field ops ----
    OPS = {"add": 0, "sub": 1, "mul": 2, "inv": 3, "to_mont": 4, "from_mont": 5, "mul29": 6}

    def field_op(self, field: int, op: str, a, b=None) -> np.ndarray:
        a = _u64(a, 4)
        out = np.empty_like(a)
        bp = _ptr(_u64(b, 4)) if b is not None else None
        self._check(self.lib.dehalo_field_op(self.handle, field, self.OPS[op], _ptr(a), bp, _ptr(out), a.shape[0]))
        return out

    def field_op_device(self, field: int, op: str, d_a: int, d_b: int, d_out: int, n: int, stream: int = 0):
        self._check(self.lib.dehalo_field_op_device(self.handle, field, self.OPS[op], d_a, d_b or None, d_out, n, stream or None))

    # ---- field-vector primitives around the path ----
    def eval_polynomial(self, field: int, coeffs, point) -> np.ndarray:
        c = _u64(coeffs, 4)
        out = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.dehalo_eval_polynomial(self.handle, field, _ptr(c) if c.shape[0] else None, c.shape[0], _ptr(_u64(point, 4)), _ptr(out)))
        return out

    def eval_polynomial_device(self, field: int, d_coeffs: int, length: int, stride: int, batch: int, point, d_out: int, stream: int = 0):
        self._check(self.lib.dehalo_eval_polynomial_device(self.handle, field, d_coeffs, length, stride, batch, _ptr(_u64(point, 4)), d_out, stream or None))

    def eval_polynomial_multi_device(self, field: int, d_polys: Sequence[int], length: int, points, d_out: int, stream: int = 0, wanted: Optional[Sequence[int]] = None):
        """d_out[point][polynomial] for device polynomial pointers (length coefficients each) and 1..4 points (k x 4 u64, host).  `wanted`: one
        bit mask per polynomial (bit i = point i); the pairs nobody wants are written as zero and skipped."""
        pts = _u64(points, 4)
        tbl = (C.c_void_p * max(1, len(d_polys)))(*d_polys)
        if wanted is None:
            self._check(self.lib.dehalo_eval_polynomial_multi_device(self.handle, field, tbl, len(d_polys), length, _ptr(pts), pts.shape[0], d_out, stream or None))
        else:
            assert len(wanted) == len(d_polys)
            self._check(self.lib.dehalo_eval_polynomial_multi_masked_device(self.handle, field, tbl, len(d_polys), length, _ptr(pts), pts.shape[0], bytes(wanted), d_out, stream or None))

    def batch_invert(self, field: int, values) -> np.ndarray:
        v = np.array(values, dtype=np.uint64).reshape(-1, 4)
        self._check(self.lib.dehalo_batch_invert(self.handle, field, _ptr(v) if v.shape[0] else None, v.shape[0]))
        return v

    def batch_invert_device(self, field: int, d_values: int, length: int, stream: int = 0):
        self._check(self.lib.dehalo_batch_invert_device(self.handle, field, d_values, length, stream or None))

    def prefix_product_device(self, field: int, d_in: int, length: int, d_out: int, stream: int = 0):
        self._check(self.lib.dehalo_prefix_product_device(self.handle, field, d_in, length, d_out, stream or None))

    def grand_product(self, field: int, num, den) -> np.ndarray:
        a, b = _u64(num, 4), _u64(den, 4)
        if a.shape[0] != b.shape[0]:
            raise ValueError("grand_product: num.len() != den.len()")
        z = np.zeros_like(a)
        if a.shape[0]:
            self._check(self.lib.dehalo_grand_product(self.handle, field, _ptr(a), _ptr(b), a.shape[0], _ptr(z)))
        return z

    def grand_product_batch_device(self, field: int, d_num: int, d_den: int, length: int, batch: int, stride: int, d_z: int, stream: int = 0):
        self._check(self.lib.dehalo_grand_product_batch_device(self.handle, field, d_num, d_den, length, batch, stride, d_z, stream or None))

    def grand_product_device(self, field: int, d_num: int, d_den: int, length: int, d_z: int, stream: int = 0):
        self._check(self.lib.dehalo_grand_product_device(self.handle, field, d_num, d_den, length, d_z, stream or None))

    def lincomb_device(self, field: int, d_cols: Sequence[int], coefs, length: int, d_out: int, sub_const=None, stream: int = 0):
        """out[i] = sum_j coefs[j] * cols[j][i], minus sub_const at i = 0 (device column pointers, host coefficients k x 4 u64)."""
        cf = _u64(coefs, 4) if len(d_cols) else np.zeros((0, 4), dtype=np.uint64)
        if cf.shape[0] != len(d_cols):
            raise ValueError("lincomb: one coefficient per column")
        tbl = (C.c_void_p * max(1, len(d_cols)))(*d_cols)
        sub = _ptr(_u64(sub_const, 4)) if sub_const is not None else None
        self._check(self.lib.dehalo_lincomb_device(self.handle, field, tbl, _ptr(cf) if cf.shape[0] else None, len(d_cols), length, d_out, sub, stream or None))

    def scale_device(self, field: int, d_a: int, length: int, pattern=None, d_factor: int = 0, stream: int = 0):
        """a[i] *= pattern[i mod len(pattern)] (host, 1/2/4/8 elements) and / or *= the element at device pointer d_factor."""
        pat = _u64(pattern, 4) if pattern is not None else None
        self._check(self.lib.dehalo_scale_device(self.handle, field, d_a, length, _ptr(pat) if pat is not None else None, pat.shape[0] if pat is not None else 0,
                                                 d_factor or None, stream or None))

    def kate_division(self, field: int, a, point) -> np.ndarray:
        a = _u64(a, 4)
        if a.shape[0] == 0:
            raise ValueError("kate_division: empty polynomial")   # upstream: a.len() - 1 underflows
        q = np.zeros((a.shape[0] - 1, 4), dtype=np.uint64)
        self._check(self.lib.dehalo_kate_division(self.handle, field, _ptr(a), a.shape[0], _ptr(_u64(point, 4)), _ptr(q) if q.shape[0] else None))
        return q

    def kate_division_batch_device(self, field: int, d_a: Sequence[int], length: int, points, d_q: Sequence[int], stream: int = 0):
        pts = _u64(points, 4)
        ta, tq = (C.c_void_p * max(1, len(d_a)))(*d_a), (C.c_void_p * max(1, len(d_q)))(*d_q)
        self._check(self.lib.dehalo_kate_division_batch_device(self.handle, field, ta, length, _ptr(pts), tq, len(d_a), stream or None))

    def kate_division_device(self, field: int, d_a: int, length: int, point, d_q: int, stream: int = 0):
        self._check(self.lib.dehalo_kate_division_device(self.handle, field, d_a, length, _ptr(_u64(point, 4)), d_q, stream or None))

    def permute_expression_pair(self, field: int, input_values, table_values, usable_rows: int):
        """-> (permuted_input, permuted_table), usable_rows x 4 u64 each; DehaloError(-6) when an input value is not in the table."""
        a, t = _u64(input_values, 4), _u64(table_values, 4)
        if a.shape[0] < usable_rows or t.shape[0] < usable_rows:
            raise ValueError("permute_expression_pair: fewer than usable_rows values")
        pi, pt = np.zeros((usable_rows, 4), dtype=np.uint64), np.zeros((usable_rows, 4), dtype=np.uint64)
        if usable_rows:
            self._check(self.lib.dehalo_permute_expression_pair(self.handle, field, _ptr(a), _ptr(t), usable_rows, _ptr(pi), _ptr(pt)))
        return pi, pt

    def permute_expression_pair_batch_device(self, field: int, d_inputs: int, d_tables: int, usable_rows: int, batch: int, stride: int, d_permuted_inputs: int,
                                             d_permuted_tables: int, stream: int = 0):
        self._check(self.lib.dehalo_permute_expression_pair_batch_device(self.handle, field, d_inputs, d_tables, usable_rows, batch, stride, d_permuted_inputs,
                                                                         d_permuted_tables, stream or None))

    def permute_expression_pair_ptrs_device(self, field: int, d_inputs: Sequence[int], d_tables: Sequence[int], usable_rows: int, d_permuted_inputs: Sequence[int],
                                            d_permuted_tables: Sequence[int], stream: int = 0):
        """columns as lists of device pointers; lookups whose table pointers are equal share one table sort"""
        arr = lambda xs: (C.c_void_p * max(1, len(xs)))(*xs)
        self._check(self.lib.dehalo_permute_expression_pair_ptrs_device(self.handle, field, arr(d_inputs), arr(d_tables), usable_rows, len(d_inputs), arr(d_permuted_inputs),
                                                                        arr(d_permuted_tables), stream or None))

    def permute_expression_pair_ptrs_deferred_device(self, field: int, d_inputs: Sequence[int], d_tables: Sequence[int], usable_rows: int, d_permuted_inputs: Sequence[int],
                                                     d_permuted_tables: Sequence[int], d_status: int, stream: int = 0):
        """the same without the synchronisation: d_status (device, one int32 per lookup) receives non-zero where an input value is not in the table"""
        arr = lambda xs: (C.c_void_p * max(1, len(xs)))(*xs)
        self._check(self.lib.dehalo_permute_expression_pair_ptrs_deferred_device(self.handle, field, arr(d_inputs), arr(d_tables), usable_rows, len(d_inputs), arr(d_permuted_inputs),
                                                                                 arr(d_permuted_tables), d_status, stream or None))

    def permute_expression_pair_distinct_device(self, field: int, d_inputs: Sequence[int], d_tables: Sequence[int], usable_rows: int, d_permuted_inputs: Sequence[int],
                                                d_permuted_tables: Sequence[int], d_rep_rows: Sequence[int], d_multiplicities: Sequence[int], distinct_counts: Sequence[int],
                                                d_status: int = 0, stream: int = 0):
        """tables of fixed columns given as distinct rows: per lookup a device array of representative row indices, one of multiplicities (uint32 each) and their number"""
        arr = lambda xs: (C.c_void_p * max(1, len(xs)))(*xs)
        cnt = (C.c_uint32 * max(1, len(distinct_counts)))(*distinct_counts)
        self._check(self.lib.dehalo_permute_expression_pair_distinct_device(self.handle, field, arr(d_inputs), arr(d_tables), usable_rows, len(d_inputs), arr(d_permuted_inputs),
                                                                            arr(d_permuted_tables), arr(d_rep_rows), arr(d_multiplicities), cnt, d_status or None, stream or None))

    def permute_expression_pair_device(self, field: int, d_input: int, d_table: int, usable_rows: int, d_permuted_input: int, d_permuted_table: int, stream: int = 0):
        self._check(self.lib.dehalo_permute_expression_pair_device(self.handle, field, d_input, d_table, usable_rows, d_permuted_input, d_permuted_table, stream or None))

    # ---- quotient numerator; see evaluation.py for the upstream-shaped wrapper ----
    def graph_create(self, field: int, constants, rotations, calcs, parts, num_intermediates: int):
        """calcs: [(op, (kind, index, rot), (kind, index, rot), parts_begin, parts_len, target)]; parts: [(kind, index, rot)]."""
        cst = _u64(constants, 4) if len(constants) else np.zeros((0, 4), dtype=np.uint64)
        rot = (C.c_int32 * max(1, len(rotations)))(*rotations)
        cc = (CCalculation * max(1, len(calcs)))()
        for i, (op, a, b, pb, pl, tgt) in enumerate(calcs):
            cc[i] = CCalculation(op, CSource(*a), CSource(*b), pb, pl, tgt)
        pp = (CSource * max(1, len(parts)))(*[CSource(*p) for p in parts])
        h = C.c_void_p()
        self._check(self.lib.dehalo_graph_create(self.handle, field, _ptr(cst) if cst.shape[0] else None, cst.shape[0], rot, len(rotations), cc, len(calcs), pp,
                                                 len(parts), num_intermediates, C.byref(h)))
        return h

    def graph_release(self, graph):
        self._check(self.lib.dehalo_graph_release(self.handle, graph))

    def _ptr_table(self, ptrs):
        """ctypes array of a list of device pointers; the table of a list OBJECT that is passed again (a prover hands the same
        column lists to every graph of a phase) is reused as long as the list has not changed."""
        key = id(ptrs)
        hit = self._tables.get(key)
        if hit is not None and hit[0] is ptrs and hit[1] == ptrs:
            return hit[2]
        table = (C.c_void_p * max(1, len(ptrs)))(*ptrs)
        if isinstance(ptrs, list):
            if len(self._tables) > 64:
                self._tables.clear()
            self._tables[key] = (ptrs, list(ptrs), table)
        return table

    def graph_evaluate_device(self, graph, fixed, advice, instance, challenges, beta, gamma, theta, y, log_rows: int, rot_scale: int, d_previous: int,
                              d_out: int, stream: int = 0, form_flags: int = 0):
        """fixed / advice / instance: lists of device pointers; challenges: k x 4 u64; beta..y: 4 u64 or None."""
        keep = [np.ascontiguousarray(v, dtype=np.uint64).reshape(4) if v is not None else None for v in (beta, gamma, theta, y)]
        ch = _u64(challenges, 4) if challenges is not None and len(challenges) else np.zeros((0, 4), dtype=np.uint64)
        tf, ta, ti = self._ptr_table(fixed), self._ptr_table(advice), self._ptr_table(instance)
        inp = CEvalInputs(tf, len(fixed), ta, len(advice), ti, len(instance), ch.ctypes.data if ch.shape[0] else None, ch.shape[0],
                          *[k.ctypes.data if k is not None else None for k in keep], form_flags)
        self._check(self.lib.dehalo_graph_evaluate_device(self.handle, graph, C.byref(inp), log_rows, rot_scale, d_previous or None, d_out, stream or None))

    def graph_evaluate_batch_device(self, graphs, fixed, advice, instance, challenges, beta, gamma, theta, y, log_rows: int, rot_scale: int, d_outs,
                                    stream: int = 0, form_flags: int = 0):
        """several programs over the same inputs, program i into d_outs[i] (one call)."""
        keep = [np.ascontiguousarray(v, dtype=np.uint64).reshape(4) if v is not None else None for v in (beta, gamma, theta, y)]
        ch = _u64(challenges, 4) if challenges is not None and len(challenges) else np.zeros((0, 4), dtype=np.uint64)
        tf, ta, ti = self._ptr_table(fixed), self._ptr_table(advice), self._ptr_table(instance)
        inp = CEvalInputs(tf, len(fixed), ta, len(advice), ti, len(instance), ch.ctypes.data if ch.shape[0] else None, ch.shape[0],
                          *[k.ctypes.data if k is not None else None for k in keep], form_flags)
        gs, outs = (C.c_void_p * len(graphs))(*graphs), (C.c_void_p * len(d_outs))(*d_outs)
        self._check(self.lib.dehalo_graph_evaluate_batch_device(self.handle, gs, len(graphs), C.byref(inp), log_rows, rot_scale, outs, stream or None))

    def permutation_h_device(self, field: int, z, columns, sigma, chunk_len: int, last_rotation: int, l0: int, l_last: int, l_active: int, beta, gamma, y,
                             delta, beta_zeta, extended_omega, log_rows: int, rot_scale: int, d_values: int, stream: int = 0, form_flags: int = 0):
        keep = [np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (beta, gamma, y, delta, beta_zeta, extended_omega)]
        tz, tc, ts = self._ptr_table(z), self._ptr_table(columns), self._ptr_table(sigma)
        inp = CPermInputs(tz, len(z), tc, ts, len(columns), chunk_len, last_rotation, l0, l_last, l_active, *[k.ctypes.data for k in keep], form_flags)
        self._check(self.lib.dehalo_permutation_h_device(self.handle, field, C.byref(inp), log_rows, rot_scale, d_values, stream or None))

    def lookup_h_device(self, field: int, product: int, permuted_input: int, permuted_table: int, table_value: int, l0: int, l_last: int, l_active: int,
                        beta, gamma, y, log_rows: int, rot_scale: int, d_values: int, stream: int = 0, form_flags: int = 0):
        keep = [np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (beta, gamma, y)]
        inp = CLookupInputs(product, permuted_input, permuted_table, table_value, l0, l_last, l_active, *[k.ctypes.data for k in keep], form_flags)
        self._check(self.lib.dehalo_lookup_h_device(self.handle, field, C.byref(inp), log_rows, rot_scale, d_values, stream or None))

    def lookup_h_batch_device(self, field: int, lookups: Sequence[Tuple[int, int, int, int]], l0: int, l_last: int, l_active: int, beta, gamma, y,
                              log_rows: int, rot_scale: int, d_values: int, stream: int = 0, form_flags: int = 0):
        """lookups: (product, permuted_input, permuted_table, table_value) device pointers per lookup, in upstream's order; one pass."""
        keep = [np.ascontiguousarray(v, dtype=np.uint64).reshape(4) for v in (beta, gamma, y)]
        arr = (CLookupInputs * len(lookups))(*[CLookupInputs(z, a, s_, tv, l0, l_last, l_active, *[k.ctypes.data for k in keep], form_flags)
                                               for z, a, s_, tv in lookups])
        self._check(self.lib.dehalo_lookup_h_batch_device(self.handle, field, arr, len(lookups), log_rows, rot_scale, d_values, stream or None))

    # ---- measurement ----
    def product_terms_device(self, field: int, columns: Sequence[int], sigma: Sequence[int], chunk_len: int, omega_powers: int, beta, gamma, delta, set_factors,
                             lookups: Sequence[Tuple[int, int, int, int]], n: int, d_num: int, d_den: int, stride: int, stream: int = 0):
        """every grand product's per-row numerator / denominator in one launch; lookups = (A, S, a', s') device pointers per lookup;
        beta / gamma / delta: 4 x u64 Montgomery; set_factors: (sets, 4) = beta * delta^(chunk_len * set)"""
        arr = lambda xs: (C.c_void_p * max(1, len(xs)))(*xs)
        keep = [np.ascontiguousarray(a, dtype=np.uint64) for a in (beta, gamma, delta, set_factors)]
        inp = CProductInputs()
        inp.columns, inp.sigma, inp.num_columns, inp.chunk_len, inp.omega_powers = arr(columns), arr(sigma), len(columns), chunk_len, omega_powers
        inp.beta, inp.gamma, inp.delta, inp.set_factors = (k.ctypes.data for k in keep)
        inp.compressed_input, inp.compressed_table = arr([l[0] for l in lookups]), arr([l[1] for l in lookups])
        inp.permuted_input, inp.permuted_table, inp.num_lookups = arr([l[2] for l in lookups]), arr([l[3] for l in lookups]), len(lookups)
        self._check(self.lib.dehalo_product_terms_device(self.handle, field, C.byref(inp), n, d_num, d_den, stride, stream or None))

    def timing_enable(self, on: bool = True):
        self._check(self.lib.dehalo_timing_enable(self.handle, int(on)))

    def timing_reset(self):
        self._check(self.lib.dehalo_timing_reset(self.handle))

    def timing_get(self, kernel_id: int):
        ms, cnt = C.c_double(), C.c_uint64()
        self._check(self.lib.dehalo_timing_get(self.handle, kernel_id, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


_TRANSFER = {}
_TRANSFER_LOCK = threading.RLock()


class _TransferUse:
    """`with transfer_context() as c:` -- the per-device transfer context, held under the module's lock for the duration of the copy: two threads cannot both create
    one (and leak a stream), and release_transfer_contexts() cannot close a context another thread is copying through."""

    def __enter__(self):
        import torch

        _TRANSFER_LOCK.acquire()
        try:
            dev = torch.cuda.current_device()
            c = _TRANSFER.get(dev)
            if c is None or c.handle is None:
                c = _TRANSFER[dev] = Context(dev, priority=-1)
            return c
        except BaseException:
            _TRANSFER_LOCK.release()
            raise

    def __exit__(self, *exc):
        _TRANSFER_LOCK.release()
        return False


def transfer_context() -> "_TransferUse":
    """One context per device whose only job is moving host arrays to / from HBM through the library's staged copies (Context.upload / download_tensor) for
    code that has no context at hand (keygen.to_device / to_host).  Its stream carries nothing else; both calls return with the copy complete.  Use as a
    context manager (see _TransferUse)."""
    return _TransferUse()


def release_transfer_contexts():
    """Closes the transfer contexts (their streams count against the runtime's pool of hardware queues: a caller about to keep several proving contexts in
    flight drops them first; the next to_device / to_host makes a new one).  Waits for copies in flight on other threads."""
    with _TRANSFER_LOCK:
        for dev in list(_TRANSFER):
            c = _TRANSFER.pop(dev)
            if c.handle is not None:
                c.close()
