// Build check for halo2_backend.hpp (compiled by `make host_example`; needs a GPU to run).
#include <cstdio>
#include "halo2_backend.hpp"
int main() {
    try {
        halo2_amd::Backend be(0);
        std::vector<halo2_amd::Fe> a(4, halo2_amd::Fe{0, 0, 0, 0});
        halo2_amd::Fe one_m{0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL};  // pasta::Fp R
        be.best_fft(DEHALO_FIELD_PASTA_FP, a, one_m, 2);
        std::printf("%s ok\n", dehalo_version());
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
