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

// (workgroups of four waves, like the Grumpkin records: kernels_grumpkin.hip)
void launch_ecdsa_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n, uint32_t *event) {
    launch_record_level<EcdsaOp, 256>(s, W, Bp, B, dp, offsets, nullptr, n, event, nullptr);
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
// turn (a verification is 54 % products, 46 % squares: NOTEBOOK.md section 6); several waves per SIMD interleave their chains.
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
    S29 x = sp_enter<C>(a), y = sp_enter<C>(b);  // the working form of the curve routines (secp_device.hpp)
    for (uint32_t i = 0; i < iters; i++) {
        x = s29_mul<C>(x, y);
        y = s29_sqr<C>(x);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s += x.v[i] ^ y.v[i];
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void launch_secp_rate(hipStream_t s, uint32_t curve, uint32_t *out, uint32_t blocks, uint32_t iters) {
    if (curve == 0u) hipLaunchKernelGGL(secp_rate_kernel<0>, dim3(blocks), dim3(256), 0, s, out, 1u, iters);
    else hipLaunchKernelGGL(secp_rate_kernel<1>, dim3(blocks), dim3(256), 0, s, out, 1u, iters);
}

// ---- component probe for the parity tests (acvm_debug_secp): one lane per item, canonical 8 x u32 little-endian integers in and out.
// (8..11: compositions the point formulas and the square root are made of, see the code)
// what: 0 a b -> a b | 1 a -> a^2 | 2 a b -> a + b | 3 a b -> a - b | 4 a -> 1 / a | 5 a -> a^((p + 1) / 4)      (mod p, one word out)
//       6 X Y Z -> 2 (X, Y, Z) | 7 X Y Z x y -> (X, Y, Z) + (x, y)                                                  (Jacobian, three words out)
template <int C>
__device__ void secp_probe_item(uint32_t what, const uint32_t *in, uint32_t *out) {
    auto ld = [&](uint32_t i) { Fr c; for (int k = 0; k < 8; k++) c.v[k] = in[8 * i + k]; return c; };
    auto st = [&](uint32_t i, const Fr &c) { for (int k = 0; k < 8; k++) out[8 * i + k] = c.v[k]; };
    if (what == 0) st(0, sp_mul<C>(ld(0), ld(1)));
    else if (what == 1) st(0, sp_sqr<C>(ld(0)));
    else if (what == 2) st(0, sp_add<C>(ld(0), ld(1)));
    else if (what == 3) st(0, sp_sub<C>(ld(0), ld(1)));
    else if (what == 4) st(0, sp_inv<C>(ld(0)));
    else if (what == 5) st(0, sp_sqrt_candidate<C>(ld(0)));
    else if (what == 8) st(0, sp_leave<C>(s29_sqr<C>(s29_sqr<C>(sp_enter<C>(ld(0))))));                                      // a -> a^4 (a product fed to a product)
    else if (what == 9) st(0, sp_leave<C>(s29_sqr<C>(s29_addl(sp_enter<C>(ld(0)), sp_enter<C>(ld(1))))));                   // a b -> (a + b)^2 (a lazy sum into a product)
    else if (what == 10) {                                                                                                      // a b -> a - 4 b through the 16 p constants
        const S29 d = s29_dbll(sp_enter<C>(ld(1)));
        st(0, sp_leave<C>(s29_out<C>(s29_subl<C>(s29_subl<C>(sp_enter<C>(ld(0)), d, 4), d, 4))));
    } else if (what == 11) st(0, sp_leave<C>(s29_sqr_n<C>(sp_enter<C>(ld(0)), 5)));                                             // a -> a^32 (the loop of the square root chains)
    else if (what == 6 || what == 7) {
        const SJac p{sp_enter<C>(ld(0)), sp_enter<C>(ld(1)), sp_enter<C>(ld(2))};
        const SJac r = what == 6 ? sj_dbl<C>(p) : sj_add_aff<C>(p, SAff{sp_store<C>(sp_enter<C>(ld(3))), sp_store<C>(sp_enter<C>(ld(4)))});
        st(0, sp_leave<C>(r.X)); st(1, sp_leave<C>(r.Y)); st(2, sp_leave<C>(r.Z));
    }
}
__global__ void __launch_bounds__(64) secp_probe_kernel(uint32_t curve, uint32_t what, const uint32_t *__restrict__ in, uint32_t n_items, uint32_t words_in, uint32_t words_out,
                                                        uint32_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_items) return;
    if (curve == 0u) secp_probe_item<0>(what, in + (size_t)t * words_in * 8, out + (size_t)t * words_out * 8);
    else secp_probe_item<1>(what, in + (size_t)t * words_in * 8, out + (size_t)t * words_out * 8);
}
void launch_secp_probe(hipStream_t s, uint32_t curve, uint32_t what, const uint32_t *in, uint32_t n_items, uint32_t words_in, uint32_t words_out, uint32_t *out) {
    if (n_items) hipLaunchKernelGGL(secp_probe_kernel, dim3((n_items + 63) / 64), dim3(64), 0, s, curve, what, in, n_items, words_in, words_out, out);
}

// the tables of both curves (grumpkin_host.cpp keeps them in the device's table set and builds them on the set's stream)
size_t ecdsa_gtable_bytes() { return (size_t)2 * SECP_GTABLE_WORDS * 4; }
void launch_ecdsa_gtable(hipStream_t s, uint32_t *out) {
    hipLaunchKernelGGL(ecdsa_gtable_kernel, dim3(2u * (SECP_GWINDOWS << SECP_GWIN) / 64u), dim3(64), 0, s, out);
}

}  // namespace acvm
