// MSM kernels + driver instantiated for CurveBn254 (one translation unit per curve: parallel builds); ParamsKZG::setup's kernels (BN254 is the pairing curve).
#include "msm.cuh"
#include "setup.cuh"
DEFINE_MSM_ENTRY(bn254, CurveBn254)
int kzg_setup_bn254(dehalo_ctx* ctx, uint32_t k, const uint64_t s[4], const uint64_t omega[4], const uint64_t cfac[4], affine_t* d_g, affine_t* d_gl, hipStream_t st) {
    return kzg_setup_t<CurveBn254>(ctx, k, s, omega, cfac, d_g, d_gl, st);
}
