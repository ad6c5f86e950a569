#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 1234567;
    uint64_t x0 = a0, x1 = a1, x2 = a2, x3 = a3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { a0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a1), __builtin_bit_cast(u16x2, a2), a0, false); a1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a2), __builtin_bit_cast(u16x2, a3), a1, false);
                           a2 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a3), __builtin_bit_cast(u16x2, a0), a2, false); a3 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a0), __builtin_bit_cast(u16x2, a1), a3, false); }
            if (OP == 1) { a0 = __builtin_amdgcn_udot4(a1, a2, a0, false); a1 = __builtin_amdgcn_udot4(a2, a3, a1, false); a2 = __builtin_amdgcn_udot4(a3, a0, a2, false); a3 = __builtin_amdgcn_udot4(a0, a1, a3, false); }
            if (OP == 2) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x0) : "v"(a0), "v"(a1) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x1) : "v"(a1), "v"(a2) : "vcc");
                           asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x2) : "v"(a2), "v"(a3) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x3) : "v"(a3), "v"(a0) : "vcc"); }
            if (OP == 3) { a0 = a0 * a1 + a2; a1 = a1 * a2 + a3; a2 = a2 * a3 + a0; a3 = a3 * a0 + a1; }  // v_mad_u32_u24? no: mul_lo + add
            if (OP == 4) { a0 = __umul24(a0, a1) + a2; a1 = __umul24(a1, a2) + a3; a2 = __umul24(a2, a3) + a0; a3 = __umul24(a3, a0) + a1; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(x0 + x1 + x2 + x3);
}
template <class F> float t(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); float best = 1e30f; for (int r = 0; r < 3; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; }
int main() {
    uint32_t *out; const int blocks = 256 * 8 * 4, iters = 2000; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const char *names[5] = {"v_dot2_u32_u16", "v_dot4_u32_u8", "v_mad_u64_u32 (asm)", "v_mul_lo_u32+add", "v_mad_u32_u24"};
    float ms[5];
    ms[0] = t([&] { k<0><<<blocks, 256>>>(out, 1, iters); }); ms[1] = t([&] { k<1><<<blocks, 256>>>(out, 1, iters); }); ms[2] = t([&] { k<2><<<blocks, 256>>>(out, 1, iters); });
    ms[3] = t([&] { k<3><<<blocks, 256>>>(out, 1, iters); }); ms[4] = t([&] { k<4><<<blocks, 256>>>(out, 1, iters); });
    double ops = (double)blocks * 256 * iters * 16 * 4;
    for (int i = 0; i < 5; i++) printf("%-22s %8.3f ms %8.2f T lane-ops/s  (%.1f cycles per wave64 instr)\n", names[i], ms[i], ops / ms[i] / 1e9, 1024.0 * 2.4e9 / (ops / 64 / (ms[i] / 1e3)));
    return 0;
}
