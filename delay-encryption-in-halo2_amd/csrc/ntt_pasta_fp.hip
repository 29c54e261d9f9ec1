// NTT / field-op kernels + driver instantiated for PastaFp.
#include "ntt.cuh"
DEFINE_NTT_ENTRY(pasta_fp, PastaFp)
