#include "../acvm_amd/csrc/fr_device.hpp"
#include <cstdio>
using namespace acvm;
__global__ void __launch_bounds__(256) k29(uint32_t *out, uint32_t seed, int iters) {
    Fr29 a, b;
    for (int i = 0; i < 9; i++) { a.v[i] = ((threadIdx.x + 1) * 2654435761u + seed + i) & 0x1fffffffu; b.v[i] = (a.v[i] ^ 0x5bd1e995u) & 0x1fffffffu; }
    a.v[8] &= 0xfffff; b.v[8] &= 0xfffff;
    for (int i = 0; i < iters; i++) { a = fr29_mul(a, b); b = fr29_mul(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 9; i++) s += a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k32(uint32_t *out, uint32_t seed, int iters) {
    Fr a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = (threadIdx.x + 1) * 2654435761u + seed + i; b.v[i] = a.v[i] ^ 0x5bd1e995u; }
    a.v[7] &= 0x0fffffffu; b.v[7] &= 0x0fffffffu;
    for (int i = 0; i < iters; i++) { a = fr_mul(a, b); b = fr_mul(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float t(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); float best = 1e30f; for (int r = 0; r < 3; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; }
int main() {
    uint32_t *out; const int blocks = 256 * 8 * 4, it = 500; hipMalloc(&out, (size_t)blocks * 256 * 4);
    float m29 = t([&] { k29<<<blocks, 256>>>(out, 1, it); }), m32 = t([&] { k32<<<blocks, 256>>>(out, 1, it); });
    double n = (double)blocks * 256 * it * 2;
    // history: the 8 x 32 product-scanning form with v_addc carries measured 1164 cycles/wave here (round 1, before the switch)
    printf("fr29_mul (9x29, no carries) %8.3f ms %8.2f G modmul/s  %.0f cycles/wave\n", m29, n / m29 / 1e6, 1024 * 2.4e9 / (n / 64 / (m29 / 1e3)));
    printf("fr_mul   (storage form)     %8.3f ms %8.2f G modmul/s  %.0f cycles/wave\n", m32, n / m32 / 1e6, 1024 * 2.4e9 / (n / 64 / (m32 / 1e3)));
    return 0;
}
