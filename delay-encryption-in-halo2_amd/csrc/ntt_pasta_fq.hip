// NTT / field-op / field-vector kernels + drivers instantiated for PastaFq.
#include "ntt.cuh"
#include "poly.cuh"
DEFINE_NTT_ENTRY(pasta_fq, PastaFq)
DEFINE_POLY_ENTRY(pasta_fq, PastaFq)
