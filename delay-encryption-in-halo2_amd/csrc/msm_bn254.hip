// MSM kernels + driver instantiated for CurveBn254 (one translation unit per curve: parallel builds).
#include "msm.cuh"
DEFINE_MSM_ENTRY(bn254, CurveBn254)
