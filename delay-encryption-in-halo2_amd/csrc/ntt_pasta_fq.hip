// NTT / field-op kernels + driver instantiated for PastaFq.
#include "ntt.cuh"
DEFINE_NTT_ENTRY(pasta_fq, PastaFq)
