// serial_mul_probe.hip -- three forms of the Montgomery product's column scan, back to back, at 1 / 2 / 4 / 8 waves per SIMD (NOTEBOOK 6.15):
//   (0) the compiler's: hipcc re-associates `acc = (acc >> 29) + products`, starts every column from 0 ahead of time and joins it with the carry of the
//       column before by v_lshl_add_u64 -- 17 of the product's 221 VALU instructions;
//   (1) one dependent chain of single-instruction asm statements (the carry is the addend of the column's first multiply-add: 205 instructions) -- hipcc
//       puts an s_nop behind every asm STATEMENT whose result the next instruction reads (154 per product): x0.98 - x1.02, x0.72 at one wave per SIMD;
//   (2) the product's form, fr29_mul_b (fr_blocks.inc, tools/gen_mul_blocks.py): the same chain in asm statements as long as the 30-operand budget
//       allows (17 wait states per product): x1.03 - x1.06 at 4 - 8 waves per SIMD, x1.00 at 2, x1.04 at 1.
// Same values in, same values out (checked). tools/mad_hazard_probe.hip: dependent v_mad_u64_u32 back to back need no wait state on this part.
// (1) in every kernel of the library: config 4 -9 % (two waves per SIMD, spills); (2) is used by gate_eval.hpp only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/serial_mul_probe tools/serial_mul_probe.hip && tools/serial_mul_probe
#include "../acvm_amd/csrc/fr_device.hpp"
#include <cstdio>
#include <vector>
using namespace acvm;

namespace acvm {
#if defined(__HIP_DEVICE_COMPILE__)
#define FR_SERIAL_MAD 1
template <bool UNIFORM_B>
__device__ __forceinline__ uint64_t fr_mad(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t d, carry_out;
    if (UNIFORM_B) asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry_out) : "v"(a), "s"(b), "v"(c));
    else asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry_out) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <bool UNIFORM_B>
__device__ __forceinline__ uint64_t fr_mad0(uint32_t a, uint32_t b) {  // a * b
    uint64_t d, carry_out;
    if (UNIFORM_B) asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry_out) : "v"(a), "s"(b));
    else asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry_out) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint64_t fr_add32(uint32_t a, uint64_t c) {  // a + c (a * 1 + c: one instruction, no zero-extended register pair)
    uint64_t d, carry_out;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(carry_out) : "v"(a), "v"(c));
    return d;
}
#endif

#if FR_SERIAL_MAD
// the same scan, one chain: column k = carry + sum a_i b_(k-i) + sum m_i p_(k-i). For kernels that keep four or more waves per SIMD (the gate
// kernels); at one or two waves per SIMD the chain's latency shows and fr29_mul is the faster form (the Grumpkin kernels: -9 % with the chain).
__device__ __forceinline__ Fr29 fr29_mul_chain(const Fr29 &a, const Fr29 &b) {
    constexpr uint32_t M = 0x1fffffffu;
    uint64_t acc = fr_mad0<false>(a.v[0], b.v[0]);
    uint32_t m[9];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++)
            if (k) acc = fr_mad<false>(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = 0; i < k; i++) acc = fr_mad<true>(m[i], fr_p29(k - i), acc);
        const uint32_t lo = (uint32_t)acc;
        m[k] = (((lo & 1u) << 28) - lo) & M;
        acc = fr_mad<true>(m[k], fr_p29(0), acc);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc = fr_mad<false>(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc = fr_mad<true>(m[i], fr_p29(k - i), acc);
        r.v[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
#endif

// the dot as one chain. UB: bit t set = every limb of b[t] is WAVE-UNIFORM (a gate's coefficient): it stays in scalar registers (fr_mad above)
#if FR_SERIAL_MAD
template <int N, bool ADD, unsigned UB>
__device__ __forceinline__ Fr29 fr29_dot_chain(const Fr29 (&a)[N], const Fr29 (&b)[N], const Fr29 *h) {
    static_assert(N >= 1 && N <= 6, "column accumulator budget");
    constexpr uint32_t M = 0x1fffffffu;
    uint64_t acc = (UB & 1u) ? fr_mad0<true>(a[0].v[0], b[0].v[0]) : fr_mad0<false>(a[0].v[0], b[0].v[0]);
    uint32_t m[9];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = 0; i <= k; i++)
                if (k || t) acc = ((UB >> t) & 1u) ? fr_mad<true>(a[t].v[i], b[t].v[k - i], acc) : fr_mad<false>(a[t].v[i], b[t].v[k - i], acc);
#pragma unroll
        for (int i = 0; i < k; i++) acc = fr_mad<true>(m[i], fr_p29(k - i), acc);
        const uint32_t lo = (uint32_t)acc;
        m[k] = (((lo & 1u) << 28) - lo) & M;
        acc = fr_mad<true>(m[k], fr_p29(0), acc);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc = ((UB >> t) & 1u) ? fr_mad<true>(a[t].v[i], b[t].v[k - i], acc) : fr_mad<false>(a[t].v[i], b[t].v[k - i], acc);
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc = fr_mad<true>(m[i], fr_p29(k - i), acc);
        if (ADD) acc = fr_add32(h->v[k - 9], acc);
        r.v[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    if (ADD) r.v[8] += h->v[8];
    return r;
}
#endif
#if !FR_SERIAL_MAD  // the host pass parses the kernels too
__device__ __forceinline__ Fr29 fr29_mul_chain(const Fr29 &a, const Fr29 &b) { return fr29_mul(a, b); }
template <int N, bool ADD, unsigned UB>
__device__ __forceinline__ Fr29 fr29_dot_chain(const Fr29 (&a)[N], const Fr29 (&b)[N], const Fr29 *h) { return fr29_dot_impl<N, ADD>(a, b, h); }
#endif
}  // namespace acvm

template <int V>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t *__restrict__ out, uint32_t seed, uint32_t iters) {
    Fr29 a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        a.v[i] = ((threadIdx.x + 1) * 2654435761u + seed + i) & 0x1fffffffu;
        b.v[i] = (a.v[i] ^ 0x5bd1e995u) & 0x1fffffffu;
    }
    a.v[8] &= 0xfffffu;
    b.v[8] &= 0xfffffu;
    for (uint32_t i = 0; i < iters; i++) {
        if (V == 0) {
            a = fr29_mul(a, b);
            b = fr29_mul(b, a);
        } else if (V == 1) {
            a = fr29_mul_chain(a, b);
            b = fr29_mul_chain(b, a);
        } else {
            a = fr29_mul_b(a, b);  // the product's asm-block form (fr_blocks.inc)
            b = fr29_mul_b(b, a);
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s += (a.v[i] ^ b.v[i]) * (i + 1);
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// two-product dot with a uniform coefficient, the gate kernel's common reduction
template <int V>
__global__ void __launch_bounds__(256) dot_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ coef, uint32_t iters) {
    Fr29 a, b, c, d, h;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        a.v[i] = ((threadIdx.x + 1) * 2654435761u + i) & 0x1fffffffu;
        b.v[i] = (a.v[i] ^ 0x5bd1e995u) & 0x1fffffffu;
        d.v[i] = (a.v[i] * 40503u + 17u) & 0x1fffffffu;
        c.v[i] = coef[i] & 0x1fffffffu;  // wave-uniform
        h.v[i] = i;
    }
    a.v[8] &= 0xfffffu;
    b.v[8] &= 0xfffffu;
    c.v[8] &= 0xfffffu;
    d.v[8] &= 0xfffffu;
    for (uint32_t i = 0; i < iters; i++) {
        const Fr29 l[2] = {a, d}, m[2] = {b, c};  // (four different operands: with one in common hipcc factors the two products into one)
        if (V == 0) a = fr29_dot_impl<2, true>(l, m, &h);
        else a = fr29_dot_chain<2, true, 2u>(l, m, &h);
        a.v[8] &= 0xfffffu;
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s += a.v[i] * (i + 1);
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float best_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const uint32_t iters = 2000;
    uint32_t *out0, *out1, *out2, *coef;
    const size_t cap = (size_t)256 * 8 * 256;
    hipMalloc(&out0, cap * 4);
    hipMalloc(&out1, cap * 4);
    hipMalloc(&out2, cap * 4);
    hipMalloc(&coef, 64);
    uint32_t hc[9] = {0x12345678u, 0x0badcafeu, 0x1eadbeefu, 0x07654321u, 0x11111111u, 0x02222222u, 0x13333333u, 0x04444444u, 0x00055555u};
    hipMemcpy(coef, hc, 36, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 4, 8}) {
        const uint32_t blocks = 256 * wps;  // 4 waves per block, 4 SIMDs per CU: wps waves per SIMD
        const float m0 = best_ms([&] { rate_kernel<0><<<blocks, 256>>>(out0, 1, iters); });
        const float m1 = best_ms([&] { rate_kernel<1><<<blocks, 256>>>(out1, 1, iters); });
        const float m2 = best_ms([&] { rate_kernel<2><<<blocks, 256>>>(out2, 1, iters); });
        std::vector<uint32_t> h0((size_t)blocks * 256), h1((size_t)blocks * 256), h2((size_t)blocks * 256);
        hipMemcpy(h0.data(), out0, h0.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), out2, h2.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0, bad2 = 0;
        for (size_t i = 0; i < h0.size(); i++) bad += h0[i] != h1[i], bad2 += h0[i] != h2[i];
        printf("mul  %d waves/SIMD: asm blocks %8.3f ms %7.1f G/s (x%.3f of the compiler form)  mismatches %zu\n", wps, m2, (double)blocks * 256 * iters * 2 / m2 / 1e6, m0 / m2, bad2);
        const double prods = (double)blocks * 256 * iters * 2;
        printf("mul  %d waves/SIMD: compiler form %8.3f ms %7.1f G/s | serial chain %8.3f ms %7.1f G/s  (x%.3f)  mismatches %zu\n", wps, m0, prods / m0 / 1e6, m1,
               prods / m1 / 1e6, m0 / m1, bad);
        const float d0 = best_ms([&] { dot_kernel<0><<<blocks, 256>>>(out0, coef, iters); });
        const float d1 = best_ms([&] { dot_kernel<1><<<blocks, 256>>>(out1, coef, iters); });
        hipMemcpy(h0.data(), out0, h0.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost);
        bad = 0;
        for (size_t i = 0; i < h0.size(); i++) bad += h0[i] != h1[i];
        const double dots = (double)blocks * 256 * iters;
        printf("dot2 %d waves/SIMD: compiler form %8.3f ms %7.1f G/s | serial chain %8.3f ms %7.1f G/s  (x%.3f)  mismatches %zu\n", wps, d0, dots / d0 / 1e6, d1,
               dots / d1 / 1e6, d0 / d1, bad);
    }
    return 0;
}
