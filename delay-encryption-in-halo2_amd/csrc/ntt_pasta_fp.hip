// NTT / field-op / field-vector kernels + drivers instantiated for PastaFp.
#include "ntt.cuh"
#include "poly.cuh"
DEFINE_NTT_ENTRY(pasta_fp, PastaFp)
DEFINE_POLY_ENTRY(pasta_fp, PastaFp)
