// tools/microbench.hip -- instruction-rate and bandwidth probes used to choose the limb representation
// and to quote measured ceilings beside the spec numbers. Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include "../acvm_amd/csrc/fr_device.hpp"
using namespace acvm;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 1234567;
    uint64_t x0 = a0, x1 = a1, x2 = a2, x3 = a3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) {  // v_mad_u64_u32, 4 independent chains
                x0 = (uint64_t)(uint32_t)x0 * a0 + x0; x1 = (uint64_t)(uint32_t)x1 * a1 + x1;
                x2 = (uint64_t)(uint32_t)x2 * a2 + x2; x3 = (uint64_t)(uint32_t)x3 * a3 + x3;
            } else if (OP == 1) {  // v_mul_lo_u32
                a0 = a0 * a1 + 1; a1 = a1 * a2 + 1; a2 = a2 * a3 + 1; a3 = a3 * a0 + 1;
            } else if (OP == 2) {  // v_mul_hi_u32
                a0 = __umulhi(a0, a1) | 1; a1 = __umulhi(a1, a2) | 3; a2 = __umulhi(a2, a3) | 5; a3 = __umulhi(a3, a0) | 7;
            } else if (OP == 3) {  // v_fma_f64
                d0 = fma(d0, 1.0000001, d1); d1 = fma(d1, 0.9999999, d2); d2 = fma(d2, 1.0000001, d3); d3 = fma(d3, 0.9999999, d0);
            } else if (OP == 4) {  // v_add_u32 / xor
                a0 = (a0 + a1) ^ a2; a1 = (a1 + a2) ^ a3; a2 = (a2 + a3) ^ a0; a3 = (a3 + a0) ^ a1;
            } else if (OP == 5) {  // v_mad_u32_u24
                a0 = __umul24(a0, a1) + a2; a1 = __umul24(a1, a2) + a3; a2 = __umul24(a2, a3) + a0; a3 = __umul24(a3, a0) + a1;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(x0 + x1 + x2 + x3) + (uint32_t)(d0 + d1 + d2 + d3);
}

__global__ void __launch_bounds__(256) frmul_kernel(uint32_t *out, uint32_t seed, int iters) {
    Fr a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = (threadIdx.x + 1) * 2654435761u + seed + i; b.v[i] = a.v[i] ^ 0x5bd1e995u; }
    a.v[7] &= 0x0fffffffu; b.v[7] &= 0x0fffffffu;
    for (int i = 0; i < iters; i++) { a = fr_mul(a, b); b = fr_mul(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <class F>
static float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    if (argc > 1 && std::string(argv[1]) == "copy") {  // PMC calibration: a known byte count, 16 B/lane coalesced
        size_t bytes = 4ull << 30;
        uint4 *a, *b;
        CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 1, bytes));
        for (int r = 0; r < 3; r++) copy_kernel<<<256 * 8, 256>>>(a, b, bytes / 16);
        CHECK(hipDeviceSynchronize());
        printf("copy_kernel: 3 launches, %zu bytes read + %zu bytes written each\n", bytes, bytes);
        return 0;
    }
    uint32_t *out;
    const int blocks = 256 * 8 * 4, iters = 2000;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const char *names[6] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64", "v_add/xor_u32 (2 ops)", "v_mad_u32_u24"};
    for (int op = 0; op < 6; op++) {
        float ms = 0;
        switch (op) {
        case 0: ms = time_ms([&] { rate_kernel<0><<<blocks, 256>>>(out, 1, iters); }); break;
        case 1: ms = time_ms([&] { rate_kernel<1><<<blocks, 256>>>(out, 1, iters); }); break;
        case 2: ms = time_ms([&] { rate_kernel<2><<<blocks, 256>>>(out, 1, iters); }); break;
        case 3: ms = time_ms([&] { rate_kernel<3><<<blocks, 256>>>(out, 1, iters); }); break;
        case 4: ms = time_ms([&] { rate_kernel<4><<<blocks, 256>>>(out, 1, iters); }); break;
        case 5: ms = time_ms([&] { rate_kernel<5><<<blocks, 256>>>(out, 1, iters); }); break;
        }
        double ops = (double)blocks * 256 * iters * 16 * 4;
        printf("%-24s %8.3f ms  %8.2f Tinstr-lane/s\n", names[op], ms, ops / ms / 1e9);
    }
    {
        const int it = 500;
        float ms = time_ms([&] { frmul_kernel<<<blocks, 256>>>(out, 1, it); });
        double n = (double)blocks * 256 * it * 2;
        printf("fr_mul (8x32 CIOS)       %8.3f ms  %8.2f G modmul/s\n", ms, n / ms / 1e6);
    }
    {
        size_t bytes = 4ull << 30;
        uint4 *a, *b;
        CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 1, bytes));
        float ms = time_ms([&] { copy_kernel<<<256 * 8, 256>>>(a, b, bytes / 16); });
        printf("copy 4 GiB (uint4)       %8.3f ms  %8.1f GB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e6);
        float ms2 = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
        printf("hipMemcpyDtoD 4 GiB      %8.3f ms  %8.1f GB/s (read+write)\n", ms2, 2.0 * bytes / ms2 / 1e6);
    }
    return 0;
}
