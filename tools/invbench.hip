#include "../acvm_amd/csrc/fr_device.hpp"
#include <cstdio>
using namespace acvm;
__global__ void __launch_bounds__(64) kinv(uint32_t *out, uint32_t seed, int iters) {
    Fr a;
    for (int i = 0; i < 8; i++) a.v[i] = (threadIdx.x + 1 + blockIdx.x) * 2654435761u + seed + i;
    a.v[7] &= 0x0fffffffu;
    for (int i = 0; i < iters; i++) a = fr_inv(a);
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) kmul(uint32_t *out, uint32_t seed, int iters) {
    Fr a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = (threadIdx.x + 1 + blockIdx.x) * 2654435761u + seed + i; b.v[i] = a.v[i] ^ 0x5bd1e995u; }
    a.v[7] &= 0x0fffffffu; b.v[7] &= 0x0fffffffu;
    for (int i = 0; i < iters; i++) { a = fr_mul(a, b); b = fr_mul(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float t(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); float best = 1e30f; for (int r = 0; r < 3; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; }
int main() {
    uint32_t *out; hipMalloc(&out, (size_t)1 << 24);
    for (int waves : {1024, 8192}) {
        float mi = t([&] { kinv<<<waves, 64>>>(out, 1, 8); });
        float mm = t([&] { kmul<<<waves, 64>>>(out, 1, 100); });
        printf("waves %5d: fr_inv %8.1f us each (%.0f cycles)   fr_mul %6.2f us each (%.0f cycles)  [latency when waves=1024, throughput when 8192]\n", waves, mi * 1e3 / 8,
               mi * 1e-3 / 8 * 2.4e9 / (waves / 1024.0 > 1 ? waves / 1024.0 : 1), mm * 1e3 / 200, mm * 1e-3 / 200 * 2.4e9 / (waves / 1024.0 > 1 ? waves / 1024.0 : 1));
    }
    return 0;
}
