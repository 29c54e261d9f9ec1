// stream_concurrency.hip -- how many kernels of different streams does this device / runtime run at once?  N streams, each with K launches of a
// one-wave kernel that spins for ~T microseconds: wall time = K * T * N / (streams actually concurrent).
// hipcc --offload-arch=gfx950 -O2 -o tools/stream_concurrency tools/stream_concurrency.hip ; GPU_MAX_HW_QUEUES=8 tools/stream_concurrency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main() {
    const int K = 20;
    const long long cycles = 20000;      // wall_clock64 ticks at 100 MHz: 200 us
    for (int prio = 0; prio < 2; prio++)
    for (int N : {1, 2, 3, 4, 6, 8, 12, 16}) {
        std::vector<hipStream_t> s(N);
        for (int i = 0; i < N; i++) {
            if (prio) { int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi); hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, (i & 1) ? hi : lo); }
            else hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        }
        k_spin<<<1, 64, 0, s[0]>>>(100, nullptr);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; k++)
            for (int i = 0; i < N; i++) k_spin<<<1, 64, 0, s[i]>>>(cycles, nullptr);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%s streams %2d: %7.3f ms for %d x %d launches of 0.2 ms -> %.2f running at once\n", prio ? "mixed-priority" : "default", N, ms, N, K, N * K * 0.2 / ms);
        for (auto st : s) hipStreamDestroy(st);
    }
    return 0;
}
