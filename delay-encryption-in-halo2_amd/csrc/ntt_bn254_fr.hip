// NTT / field-op / field-vector / quotient-numerator kernels + drivers instantiated for Bn254Fr.
#include "ntt.cuh"
#include "poly.cuh"
#include "evalh.cuh"
DEFINE_NTT_ENTRY(bn254_fr, Bn254Fr)
DEFINE_POLY_ENTRY(bn254_fr, Bn254Fr)
DEFINE_EVALH_ENTRY(bn254_fr, Bn254Fr)
