// NTT / field-op / field-vector / quotient-numerator kernels + drivers instantiated for Bn254Fq.
#include "ntt.cuh"
#include "poly.cuh"
#include "evalh.cuh"
DEFINE_NTT_ENTRY(bn254_fq, Bn254Fq)
DEFINE_POLY_ENTRY(bn254_fq, Bn254Fq)
DEFINE_EVALH_ENTRY(bn254_fq, Bn254Fq)
