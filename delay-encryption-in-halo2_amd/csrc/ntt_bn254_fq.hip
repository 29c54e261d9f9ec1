// NTT / field-op / field-vector kernels + drivers instantiated for Bn254Fq.
#include "ntt.cuh"
#include "poly.cuh"
DEFINE_NTT_ENTRY(bn254_fq, Bn254Fq)
DEFINE_POLY_ENTRY(bn254_fq, Bn254Fq)
