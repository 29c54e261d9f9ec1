// NTT / field-op / field-vector kernels + drivers instantiated for Bn254Fr.
#include "ntt.cuh"
#include "poly.cuh"
DEFINE_NTT_ENTRY(bn254_fr, Bn254Fr)
DEFINE_POLY_ENTRY(bn254_fr, Bn254Fr)
