// MSM kernels + driver instantiated for CurveVesta (one translation unit per curve: parallel builds).
#include "msm.cuh"
DEFINE_MSM_ENTRY(vesta, CurveVesta)
