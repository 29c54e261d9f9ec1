// NTT / field-op kernels + driver instantiated for Bn254Fq.
#include "ntt.cuh"
DEFINE_NTT_ENTRY(bn254_fq, Bn254Fq)
