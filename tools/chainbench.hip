#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// OP 0: one dependent chain of (mad, addc) pairs as in fr_mul;  1: two independent chains interleaved;  2: four chains
// 3: dependent mad-only chain;  4: two independent mad-only chains
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 77, d = b + 1234567;
    uint64_t l0 = a, l1 = b, l2 = c, l3 = d;
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) {
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                             "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %5, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                             : "+&v"(l0), "+&v"(h0) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            } else if (OP == 1) {
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %2, vcc, %5, %6, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
                             "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %2, vcc, %7, %4, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc"
                             : "+&v"(l0), "+&v"(h0), "+&v"(l1), "+&v"(h1) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            } else if (OP == 2) {
                asm volatile("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n\tv_mad_u64_u32 %2, s[12:13], %9, %10, %2\n\tv_mad_u64_u32 %4, s[14:15], %10, %11, %4\n\tv_mad_u64_u32 %6, s[16:17], %11, %8, %6\n\t"
                             "v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n\tv_addc_co_u32 %3, s[12:13], 0, %3, s[12:13]\n\tv_addc_co_u32 %5, s[14:15], 0, %5, s[14:15]\n\tv_addc_co_u32 %7, s[16:17], 0, %7, s[16:17]"
                             : "+&v"(l0), "+&v"(h0), "+&v"(l1), "+&v"(h1), "+&v"(l2), "+&v"(h2), "+&v"(l3), "+&v"(h3) : "v"(a), "v"(b), "v"(c), "v"(d)
                             : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
            } else if (OP == 3) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %4, %1, %0"
                             : "+&v"(l0) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            } else {
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %4, %1\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %5, %2, %1"
                             : "+&v"(l0), "+&v"(l1) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(l0 + l1 + l2 + l3) + h0 + h1 + h2 + h3;
}
template <class F> float t(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); float best = 1e30f; for (int r = 0; r < 3; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; }
int main() {
    uint32_t *out; const int blocks = 256 * 8 * 4, iters = 1000; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const char *names[5] = {"1 chain (mad+addc) x4", "2 chains (mad+addc) x4", "4 chains, sgpr carries x4", "1 chain mad only x4", "2 chains mad only x4"};
    float ms[5];
    ms[0] = t([&] { k<0><<<blocks, 256>>>(out, 1, iters); }); ms[1] = t([&] { k<1><<<blocks, 256>>>(out, 1, iters); }); ms[2] = t([&] { k<2><<<blocks, 256>>>(out, 1, iters); });
    ms[3] = t([&] { k<3><<<blocks, 256>>>(out, 1, iters); }); ms[4] = t([&] { k<4><<<blocks, 256>>>(out, 1, iters); });
    double mads = (double)blocks * 256 * iters * 16 * 4;
    for (int i = 0; i < 5; i++) printf("%-28s %8.3f ms  %.2f cycles per wave64 mad(+addc)\n", names[i], ms[i], 1024.0 * 2.4e9 / (mads / 64 / (ms[i] / 1e3)));
    return 0;
}
