// MSM kernels + driver instantiated for CurvePallas (one translation unit per curve: parallel builds).
#include "msm.cuh"
DEFINE_MSM_ENTRY(pallas, CurvePallas)
