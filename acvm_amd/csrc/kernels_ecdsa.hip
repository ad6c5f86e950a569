// kernels_ecdsa.hip -- EcdsaSecp256k1 / EcdsaSecp256r1 opcodes (device routines in ops_ecdsa.hpp / secp_device.hpp): the level kernel and
// the tables of the two generators.
#include "ops_ecdsa.hpp"
#include "ops_kernel.hpp"

namespace acvm {

struct EcdsaOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *, SlowResult *, const ExactLanes *, uint32_t) {
        return op_ecdsa(p, rec, dp.ecdsa_g);
    }
};

void launch_ecdsa_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n, uint32_t *event) {
    launch_record_level<EcdsaOp, 64>(s, W, Bp, B, dp, offsets, nullptr, n, event, nullptr);
}

// one thread per (curve, window j, digit d): 2 x SECP_GWINDOWS x 2^SECP_GWIN entries of 16 words, digit 0 left zero
__global__ void ecdsa_gtable_kernel(uint32_t *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_curve = SECP_GWINDOWS << SECP_GWIN, curve = t / per_curve, j = (t % per_curve) >> SECP_GWIN, d = t & ((1u << SECP_GWIN) - 1u);
    if (curve > 1u) return;
    SAff e{fr_zero(), fr_zero()};
    if (d) e = curve == 0u ? secp_gtable_entry<0>(j, d) : secp_gtable_entry<1>(j, d);
    uint32_t *o = out + (size_t)t * 16u;
#pragma unroll
    for (int k = 0; k < 8; k++) { o[k] = e.x.v[k]; o[8 + k] = e.y.v[k]; }
}

// The ALU roofline of the ECDSA kernels: every lane runs a chain of 2 * iters base-field products of one curve, a product and a square in
// turn (a verification is 54 % products, 46 % squares: DESIGN.md section 6); several waves per SIMD interleave their chains.
template <int C>
__global__ void __launch_bounds__(256) secp_rate_kernel(uint32_t *__restrict__ out, uint32_t seed, uint32_t iters) {
    Fr a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.v[i] = (threadIdx.x + 1) * 2654435761u + seed + i;
        b.v[i] = a.v[i] ^ 0x5bd1e995u;
    }
    a.v[7] >>= 1;  // < p
    b.v[7] >>= 1;
    for (uint32_t i = 0; i < iters; i++) {
        a = sp_mul<C>(a, b);
        b = sp_sqr<C>(a);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a.v[i] ^ b.v[i];
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void launch_secp_rate(hipStream_t s, uint32_t curve, uint32_t *out, uint32_t blocks, uint32_t iters) {
    if (curve == 0u) hipLaunchKernelGGL(secp_rate_kernel<0>, dim3(blocks), dim3(256), 0, s, out, 1u, iters);
    else hipLaunchKernelGGL(secp_rate_kernel<1>, dim3(blocks), dim3(256), 0, s, out, 1u, iters);
}

// the tables of both curves (grumpkin_host.cpp keeps them in the device's table set and builds them on the set's stream)
size_t ecdsa_gtable_bytes() { return (size_t)2 * SECP_GTABLE_WORDS * 4; }
void launch_ecdsa_gtable(hipStream_t s, uint32_t *out) {
    hipLaunchKernelGGL(ecdsa_gtable_kernel, dim3(2u * (SECP_GWINDOWS << SECP_GWIN) / 64u), dim3(64), 0, s, out);
}

}  // namespace acvm
