// NTT / field-op / field-vector / quotient-numerator kernels + drivers instantiated for PastaFq.
#include "ntt.cuh"
#include "poly.cuh"
#include "evalh.cuh"
DEFINE_NTT_ENTRY(pasta_fq, PastaFq)
DEFINE_POLY_ENTRY(pasta_fq, PastaFq)
DEFINE_EVALH_ENTRY(pasta_fq, PastaFq)
