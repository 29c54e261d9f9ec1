// NTT / field-op kernels + driver instantiated for Bn254Fr.
#include "ntt.cuh"
DEFINE_NTT_ENTRY(bn254_fr, Bn254Fr)
