// NTT / field-op / field-vector / quotient-numerator kernels + drivers instantiated for PastaFp.
#include "ntt.cuh"
#include "poly.cuh"
#include "evalh.cuh"
DEFINE_NTT_ENTRY(pasta_fp, PastaFp)
DEFINE_POLY_ENTRY(pasta_fp, PastaFp)
DEFINE_EVALH_ENTRY(pasta_fp, PastaFp)
