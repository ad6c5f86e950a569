// kernels.hip -- hand-written gfx950 kernels of the batched ACIR witness solver.
//
//  import_witness_kernel   ACVM::new's initial WitnessMap (pwg/mod.rs:146-156): canonical big-endian
//                          -> Montgomery SoA (from_be_bytes_reduce, generic_ark.rs:281-283)
//  arith_level_kernel      ArithmeticSolver::solve (pwg/arithmetic.rs:27-127) for one dependency level,
//                          one lane per witness instance, generic-instance plan from plan.cpp
//  (the exact in-order kernels and every non-arithmetic opcode live in kernels_ops.hip / kernels_hash.hip /
//   kernels_grumpkin.hip / kernels_brillig.hip)
//  export_witness_kernel   FieldElement::to_be_bytes (generic_ark.rs:269-277) for witness_map()/finalize()
//
// Wave64 throughout; no LDS is needed by the streaming kernels (every operand is read once per lane);
// gate records and circuit constants are wave-uniform and travel through the scalar cache.
#include "ops_common.hpp"
#include "kernels.hpp"
#include "tuning.hpp"

namespace acvm {

// ------------------------------------------------------------------------------------------ import
// in: [B][n_in][32] big-endian. A block converts 64 instances x 4 consecutive inputs: the four inputs of an instance are 128 contiguous
// bytes of `in` (one line per four lanes, where one lane per (instance, input) had every lane fetch its own 32 bytes from a line of its own),
// and the values reach the table through LDS, 64 lanes to a row (1 KiB contiguous per half). ALIGNED: `in` is 16-byte aligned (every
// buffer of the library is; a caller's device pointer may not be) and the bytes travel as two 16-byte loads.
// gate: null, or a device word that must be zero for the import to happen (the count of instances that left the generic path in the solve
// the import was enqueued behind: batch.cpp "the next tile's import behind the solve").
template <bool ALIGNED>
__global__ void __launch_bounds__(256) import_witness_kernel(uint4 *__restrict__ W, uint64_t Bp, uint32_t B,
                                                             const uint8_t *__restrict__ in, const uint32_t *__restrict__ ids,
                                                             uint32_t n_in, const uint32_t *__restrict__ gate, const uint32_t *__restrict__ plane_of_input,
                                                             uint32_t *__restrict__ plane, uint32_t *__restrict__ event_reset) {
    __shared__ uint4 tile[4][2][65];
    __shared__ uint32_t tile_low[4][64];  // byte planes (plan.hpp): low 29 bits of the canonical value | is-byte << 31
    if (gate && *gate != 0u) return;  // (block-uniform)
    const uint32_t t = threadIdx.x;
    const uint64_t j0 = (uint64_t)blockIdx.x * 64u;
    const uint32_t k0 = blockIdx.y * 4u;
    {
        const uint32_t ji = t >> 2, kk = t & 3u;
        const uint64_t j = j0 + ji;
        const uint32_t k = k0 + kk;
        if (j < B && k < n_in) {
            const uint8_t *p = in + (j * n_in + k) * 32u;
            Fr x;
            if (ALIGNED) {
                const uint4 *q = (const uint4 *)p;  // read once (nontemporal): bytes [0, 16) are the most significant
                x.v[7] = __builtin_bswap32(__builtin_nontemporal_load(&q[0].x)); x.v[6] = __builtin_bswap32(__builtin_nontemporal_load(&q[0].y));
                x.v[5] = __builtin_bswap32(__builtin_nontemporal_load(&q[0].z)); x.v[4] = __builtin_bswap32(__builtin_nontemporal_load(&q[0].w));
                x.v[3] = __builtin_bswap32(__builtin_nontemporal_load(&q[1].x)); x.v[2] = __builtin_bswap32(__builtin_nontemporal_load(&q[1].y));
                x.v[1] = __builtin_bswap32(__builtin_nontemporal_load(&q[1].z)); x.v[0] = __builtin_bswap32(__builtin_nontemporal_load(&q[1].w));
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint8_t *q = p + 28 - 4 * i;  // limb i = bytes [28-4i, 32-4i) big-endian
                    x.v[i] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | (uint32_t)q[3];
                }
            }
            // from_be_bytes_reduce: x mod p for any x < 2^256 (2^256 / p < 6). The quotient is estimated from the top limb -- q = how many multiples of
            // p7 + 1 fit into x7, p7 = p's top limb: floor(x / p) is q or q + 1 (checked exhaustively at the multiples of p and on 2 x 10^5 random values) --
            // so x - q p followed by ONE conditional subtraction replaces round 1's loop of five (100 instructions of the ~400 per value: the import
            // of config 3's 64 inputs per instance was bound by instruction issue at 40 % of the HBM rate)
            {
                const uint32_t p7 = fr_p(7) + 1u;
                const uint32_t q = (x.v[7] >= p7) + (x.v[7] >= 2u * p7) + (x.v[7] >= 3u * p7) + (x.v[7] >= 4u * p7) + (x.v[7] >= 5u * p7);
                int64_t carry = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int64_t tt = (int64_t)x.v[i] - (int64_t)((uint64_t)q * fr_p(i)) + carry;  // q p_i < 2^35
                    x.v[i] = (uint32_t)tt;
                    carry = tt >> 32;
                }
                Fr d;
                uint64_t br = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    uint64_t tt = (uint64_t)x.v[i] - fr_p(i) - br;
                    d.v[i] = (uint32_t)tt;
                    br = (tt >> 32) & 1;
                }
                if (!br) x = d;
            }
            const bool is_byte = !(x.v[1] | x.v[2] | x.v[3] | x.v[4] | x.v[5] | x.v[6] | x.v[7]) && x.v[0] < 256u;
            // into Montgomery form: a byte (message bytes, digits, flags: the usual initial witness of the hash circuits) by the closed form of
            // ops_common.hpp fr_mont_of_byte (~64 instructions, no product) when every value the wave converts is one; else the product with R^2
            Fr m;
            if (__builtin_amdgcn_ballot_w64(!is_byte) == 0) m = fr_mont_of_byte(x.v[0]);
            else m = fr_mul(x, fr_r2());
            tile_low[kk][ji] = (x.v[0] & 0x1fffffffu) | (is_byte ? 0x80000000u : 0u);
            tile[kk][0][ji] = make_uint4(m.v[0], m.v[1], m.v[2], m.v[3]);
            tile[kk][1][ji] = make_uint4(m.v[4], m.v[5], m.v[6], m.v[7]);
        }
    }
    __syncthreads();
    {
        const uint32_t kk = t >> 6, ji = t & 63u;
        const uint64_t j = j0 + ji;
        const uint32_t k = k0 + kk;
        if (j < B && k < n_in) {
            // (nontemporal like the level kernels' stores: the rows are read by later launches, not by this one)
            const uint4 lo = tile[kk][0][ji], hi = tile[kk][1][ji];
            const Fr m = {{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
            fr_store_nt(W, ids[k], Bp, j, m);
            if (plane_of_input) {  // (block-uniform per kk: a scalar load)
                const uint32_t pl = plane_of_input[k];
                if (pl != 0xFFFFFFFFu) plane[(uint64_t)pl * Bp + j] = tile_low[kk][ji];
            }
            // ACVM::new: nobody has left the generic path yet. The import of a tile leaves the event words ready for its solve (one launch less in
            // front of every solve: a config-3 step is five launches). The blocks of the first four inputs do it for their 64 instances.
            if (event_reset && k == 0) {
                event_reset[j] = 0xFFFFFFFFu;
                if (j == 0) { event_reset[-4] = 0u; event_reset[-3] = 0u; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ export
// out: [n][n_sel][32] big-endian for instances [first, first+n) and witness list sel. A witness the level kernels keep scaled
// (plan.cpp "projective witnesses") is multiplied by 1 / scale here, unless the instance went through the exact path, whose
// columns unscale_slow_kernel already restored.
__global__ void __launch_bounds__(256) export_witness_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t first,
                                                             uint32_t n, const uint32_t *__restrict__ sel, uint32_t n_sel,
                                                             uint8_t *__restrict__ out, const Unscale u, uint32_t k0, const uint32_t *__restrict__ row_of) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = k0 + blockIdx.y;
    if (t >= n) return;
    Fr one = fr_zero();
    one.v[0] = 1;
    // row_of: the table's rows under slot reuse (plan.cpp); a witness nothing produces has no row (0xFFFFFFFF) and exports as zero
    // (a selector 0xFFFFFFFF -- the host's stand-in for a witness index beyond the circuit -- has no row either)
    const uint32_t w = sel[k];
    const uint32_t row = w == 0xFFFFFFFFu ? w : row_of ? row_of[w] : w;
    Fr x = row == 0xFFFFFFFFu ? fr_zero() : fr_load(W, row, Bp, first + t);
    const uint32_t ui = u.index && w != 0xFFFFFFFFu ? u.index[w] : 0xFFFFFFFFu;
    // out of Montgomery form; a scaled column leaves it through the canonical integer 1 / scale instead of 1
    x = fr_mul(x, ui != 0xFFFFFFFFu && u.event[first + t] == 0xFFFFFFFFu ? fr_const(u.consts_plain, ui) : one);
    uint8_t *p = out + ((uint64_t)t * n_sel + k) * 32;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t *q = p + 28 - 4 * i;
        q[0] = (uint8_t)(x.v[i] >> 24);
        q[1] = (uint8_t)(x.v[i] >> 16);
        q[2] = (uint8_t)(x.v[i] >> 8);
        q[3] = (uint8_t)x.v[i];
    }
}
// columns of the instances that continue on the exact path: every scaled witness the lane keeps back to its plain Montgomery value. The lane keeps
// what init_assigned_kernel marks -- outputs of opcodes in front of its event; the rows behind it are written by the exact kernels before anything
// reads them, so an instance that fails early costs next to nothing here (with relaxed rows nearly every gate output is "scaled").
__global__ void __launch_bounds__(256) unscale_slow_kernel(uint4 *__restrict__ W, uint64_t Bp, const uint32_t *__restrict__ slow_ids,
                                                           uint32_t n_slow, const uint32_t *__restrict__ scaled_ids, uint32_t n_scaled,
                                                           const uint32_t *__restrict__ consts, const uint32_t *__restrict__ producer,
                                                           const uint32_t *__restrict__ start_opcode) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_slow * n_scaled) return;
    const uint32_t k = (uint32_t)(i / n_slow), t = (uint32_t)(i % n_slow);
    const uint64_t j = slow_ids[t];
    const uint32_t w = scaled_ids[k];
    if (producer[w] >= start_opcode[t]) return;  // (a scaled witness is a gate's output, never an initial one)
    fr_store(W, w, Bp, j, fr_mul(fr_load(W, w, Bp, j), fr_const(consts, k)));
}

// ------------------------------------------------------------------------------------------ level kernel (body: ops_common.hpp arith_level_body)
#ifndef ARITH_BLOCK
#define ARITH_BLOCK 256
#endif
#ifndef ARITH_WAVES_MIN  // (tools/build_variant.sh: 7 waves per SIMD = 72 VGPRs + 20 B of scratch measured 5.72-5.76 against 5.79-5.81 M witnesses/s, 8 = 64 VGPRs + 60 B: 5.10-5.14)
#define ARITH_WAVES_MIN 6
#endif
__global__ void __launch_bounds__(ARITH_BLOCK) __attribute__((amdgpu_waves_per_eu(ARITH_WAVES_MIN, 8)))
arith_level_kernel(uint4 *__restrict__ W, uint64_t Bp, uint32_t B, const uint32_t *__restrict__ gate_stream, const uint32_t *__restrict__ gate_offset,
                   const uint32_t *__restrict__ consts, uint32_t *__restrict__ event, const uint4 *__restrict__ Inv) {
    arith_level_body(W, Bp, B, gate_stream, gate_offset, consts, event, Inv, blockIdx.y);
}

// Denominators of the gates whose unknown is multiplied by a known witness (arithmetic.rs:68-91): 1 / partner for a batch of
// inversion jobs (plan.cpp schedules them ahead of their gates). One wave = 64 instances x up to inv_chunk jobs (tuning.hpp); the
// inversions of a lane are batched with Montgomery's trick, so a lane pays one field inversion per chunk plus 3
// multiplications per job. The prefix products are parked in the jobs' own rows of the inverse table (laid out like W,
// [slot][half][instance], coalesced) and replaced by the inverses on the way back. Values stay in the 29-bit working form
// between products (< 1.06p, never repacked); the table holds representatives < 2^256, not necessarily < p.
__global__ void __launch_bounds__(64) inverse_batch_kernel(const uint4 *__restrict__ W, uint4 *__restrict__ Inv, uint64_t Bp, uint32_t B,
                                                           const uint32_t *__restrict__ gate_stream, const uint32_t *__restrict__ job_offset,
                                                           uint32_t n_jobs, uint32_t chunk, uint32_t *__restrict__ event) {
    const uint64_t j = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= B) return;
    const uint32_t first = blockIdx.y * chunk;
    const uint32_t n = n_jobs - first < chunk ? n_jobs - first : chunk;
    Fr29 prefix = fr29_from(fr_one());
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *__restrict__ g = gate_stream + job_offset[first + i];
        Fr den = fr_load(W, g[0], Bp, j);
        if (fr_is_zero(den)) {  // zero-coefficient drop (arithmetic.rs:217-221): this instance leaves the generic path at the gate
            flag_instance(event, j, g[1]);
            den = fr_one();
        }
        prefix = fr29_mul(prefix, fr29_from(den));
        fr_store(Inv, g[2], Bp, j, fr29_pack(prefix));
    }
    Fr29 inv = fr29_from(fr_inv(fr29_pack(fr29_cond_sub_p(prefix))));  // 1 / (den_0 ... den_{n-1})
    for (uint32_t i = n; i-- > 0;) {
        const uint32_t *__restrict__ g = gate_stream + job_offset[first + i];
        Fr den = fr_load_nt(W, g[0], Bp, j);  // second and last read of the row by this launch
        if (fr_is_zero(den)) den = fr_one();
        // (the first job's "prefix before it" is 1: one product more per wave, and no second path for the compiler to merge with 126 register moves per job)
        const Fr prev = i > 0 ? fr_load(Inv, gate_stream[job_offset[first + i - 1] + 2], Bp, j) : fr_one();
        const Fr29 inv_i = fr29_mul(inv, fr29_from(prev));
        inv = fr29_mul(inv, fr29_from(den));
        fr_store_nt(Inv, g[2], Bp, j, fr29_pack(inv_i));  // read once, by a gate levels later (the three nontemporal accesses of this path: 5.54 -> 5.565 M witnesses/s)
    }
}

// ------------------------------------------------------------------------------------------ self test
// Cross-checks the hand-scheduled field routines against the portable ones on pseudo-random operands:
// asm fr_mul == portable CIOS, a * inv(a) == 1, (a + b) - b == a, a + (-a) == 0. Counts mismatching lanes.
__device__ __forceinline__ uint64_t st_mix(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) fr_selftest_kernel(uint64_t seed, uint32_t n, uint32_t *mismatches) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + 0x1234567ULL * i;
    Fr a, b;
    for (int k = 0; k < 4; k++) {
        uint64_t x = st_mix(s), y = st_mix(s);
        a.v[2 * k] = (uint32_t)x; a.v[2 * k + 1] = (uint32_t)(x >> 32);
        b.v[2 * k] = (uint32_t)y; b.v[2 * k + 1] = (uint32_t)(y >> 32);
    }
    a.v[7] &= 0x1fffffffu;  // < 2^253 < p
    b.v[7] &= 0x1fffffffu;
    if (i == 0) a = fr_zero();
    if (i == 1) { a = fr_modulus(); a.v[0] -= 1; }  // p - 1
    if (i == 2) b = fr_one();
    uint32_t bad = 0;
    if (!fr_eq(fr_mul(a, b), fr_mul_portable(a, b))) bad |= 1;
    if (!fr_eq(fr_mul(a, a), fr_mul_portable(a, a))) bad |= 2;
    Fr ia = fr_inv(a);
    if (fr_is_zero(a) ? !fr_is_zero(ia) : !fr_eq(fr_mul(a, ia), fr_one())) bad |= 4;
    if (!fr_eq(ia, fr_inv_eea(a))) bad |= 32;
    // constants with zero limbs as asm operands (the early-clobber regression): Montgomery form of 2^192 - 1 both ways
    {
        Fr c = {{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u}};
        Fr r2 = fr_r2();
        if (!fr_eq(fr_mul(c, r2), fr_mul_portable(c, r2)) || !fr_eq(fr_mul(r2, c), fr_mul_portable(r2, c))) bad |= 64;
    }
    if (!fr_eq(fr_sub(fr_add(a, b), b), a)) bad |= 8;
    if (!fr_is_zero(fr_add(a, fr_neg(a)))) bad |= 16;
    // byte-sized values (ops_common.hpp): the table, its arithmetic twin and the recognition of a stored form agree for every byte, and a
    // value that is not a byte is told apart and still yields its canonical low limb
    {
        const uint32_t d = i & 0xffu;
        uint32_t got = 0xffffffffu;
        const Fr md = fr_from_byte(d);
        bool isb;
        if (!fr_eq(md, fr_mont_of_byte(d)) || !fr_eq(md, fr_from_u32(d)) || !fr_is_byte(md, got) || got != d || fr_low_limb(md, isb) != d || !isb) bad |= 128;
        const uint32_t low = fr_low_limb(a, isb);  // a: a random reduced value (or 0, p - 1)
        const Fr ca = fr_to_canonical(a);
        const bool small = (ca.v[0] < 256u) && !(ca.v[1] | ca.v[2] | ca.v[3] | ca.v[4] | ca.v[5] | ca.v[6] | ca.v[7]);
        if (isb != small || low != (small ? ca.v[0] : (ca.v[0] & 0x1fffffffu))) bad |= 256;
    }
    // the asm-block column scans of the gate kernel (fr_blocks.inc) against the C forms, limb for limb: any normalised operands (representatives up
    // to 2^256, a lazy sum h with limbs up to 2^32), zero and all-ones limbs, and a WAVE-UNIFORM second factor for the scalar-register forms (u is
    // made of the seed alone, so every lane hands the same value to the "s" operands)
    {
        Fr29 x = fr29_from(a), y = fr29_from(b), z, u, w, h;
        uint64_t su = seed ^ 0x5DEECE66DULL;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            z.v[k] = (uint32_t)st_mix(s) & 0x1fffffffu;
            h.v[k] = (uint32_t)st_mix(s);
            u.v[k] = (uint32_t)st_mix(su) & 0x1fffffffu;
            w.v[k] = (uint32_t)st_mix(su) & 0x1fffffffu;
        }
        x.v[8] |= (uint32_t)(i & 7u) << 24;  // unreduced: up to 2^256 (gate_eval.hpp relaxed rows)
        z.v[8] &= 0x7ffffffu;
        u.v[8] &= 0xffffffu;
        w.v[8] &= 0xffffffu;
        h.v[8] &= 0xffffu;
        if (i == 3) {
#pragma unroll
            for (int k = 0; k < 9; k++) { x.v[k] = 0x1fffffffu; y.v[k] = 0x1fffffffu; }
            x.v[8] = y.v[8] = 0x7ffffffu;
        }
        if (i == 4) {
#pragma unroll
            for (int k = 0; k < 9; k++) { x.v[k] = 0; h.v[k] = 0xffffffffu; }
            h.v[8] = 0xffffu;
        }
        auto same = [](const Fr29 &p, const Fr29 &q) {
            uint32_t d = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) d |= p.v[k] ^ q.v[k];
            return d == 0;
        };
        if (!same(fr29_mul_b(x, y), fr29_mul(x, y)) || !same(fr29_mul_b(y, x), fr29_mul(y, x))) bad |= 512;
        const Fr29 l1[1] = {x}, m1v[1] = {y}, m1u[1] = {u};
        if (!same(fr29_dot_add_b<1, 0u>(l1, m1v, h), fr29_dot_add<1>(l1, m1v, h))) bad |= 1024;
        if (!same(fr29_dot_add_b<1, 1u>(l1, m1u, h), fr29_dot_add<1>(l1, m1u, h))) bad |= 1024;
        const Fr29 l2[2] = {x, z}, m2vv[2] = {y, x}, m2vu[2] = {y, u}, m2uu[2] = {w, u};
        if (!same(fr29_dot_add_b<2, 0u>(l2, m2vv, h), fr29_dot_add<2>(l2, m2vv, h))) bad |= 2048;
        if (!same(fr29_dot_add_b<2, 2u>(l2, m2vu, h), fr29_dot_add<2>(l2, m2vu, h))) bad |= 2048;
        if (!same(fr29_dot_add_b<2, 3u>(l2, m2uu, h), fr29_dot_add<2>(l2, m2uu, h))) bad |= 2048;
    }
    if (bad) atomicAdd(mismatches, 1u);
}

// ------------------------------------------------------------------------------------------ modmul rate probe
// The ALU roofline of the integer-bound kernels (Grumpkin / Pedersen / ECDSA; SURVEY 8d: "report modmul/s against a measured
// back-to-back fr_mul microbenchmark peak"): every lane runs a chain of 2 * iters Montgomery products in the 29-bit working form;
// with several waves per SIMD the chains of different waves interleave and the rate is the part's modmul throughput.
__global__ void __launch_bounds__(256) modmul_rate_kernel(uint32_t *__restrict__ out, uint32_t seed, uint32_t iters) {
    Fr29 a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        a.v[i] = ((threadIdx.x + 1) * 2654435761u + seed + i) & 0x1fffffffu;
        b.v[i] = (a.v[i] ^ 0x5bd1e995u) & 0x1fffffffu;
    }
    a.v[8] &= 0xfffffu;
    b.v[8] &= 0xfffffu;
    for (uint32_t i = 0; i < iters; i++) {
        a = fr29_mul(a, b);
        b = fr29_mul(b, a);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s += a.v[i] ^ b.v[i];
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// The streaming ceiling of the part for the gate kernel's access shape (tools/copybench.hip "2 reads + 1 write, one tile per block": a wave =
// 64 consecutive instances = 1 KiB per access, two operand rows read and one output row written per lane, the grid covers the data
// exactly like a level launch): n 16-byte units per row. bench.py reports it beside the 8 TB/s spec peak of the roofline.
__global__ void __launch_bounds__(256) stream_rate_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint4 x = a[i], y = b[i];
    out[i] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
}
void launch_stream_rate(hipStream_t s, const uint4 *a, const uint4 *b, uint4 *out, uint64_t n) {
    hipLaunchKernelGGL(stream_rate_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, s, a, b, out);
}
void launch_modmul_rate(hipStream_t s, uint32_t *out, uint32_t blocks, uint32_t iters) {
    hipLaunchKernelGGL(modmul_rate_kernel, dim3(blocks), dim3(256), 0, s, out, 1u, iters);
}

// slot reuse: the exact path re-solves a flagged instance from its initial witnesses in a table of its own (row = witness index, lane t
// = the t-th flagged instance): copy the initial witnesses over from their (never recycled) rows of the level table
__global__ void gather_initial_kernel(uint4 *__restrict__ Wx, uint64_t Bpx, const uint4 *__restrict__ W, uint64_t Bp, const uint32_t *__restrict__ init_ids,
                                      const uint32_t *__restrict__ init_rows, uint32_t n_init, const uint32_t *__restrict__ slow_ids, uint32_t n_slow) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_init * n_slow) return;
    const uint32_t k = (uint32_t)(i / n_slow), t = (uint32_t)(i % n_slow);
    fr_store(Wx, init_ids[k], Bpx, t, fr_load(W, init_rows[k], Bp, slow_ids[t]));
}
// asynchronous exact path without slot reuse (batch.cpp): the column of every flagged instance moves into the side table (lane t = the t-th
// flagged instance), scaled witnesses back to plain values on the way, so that the exact kernels can resume at the instance's event exactly as
// they would in place while the level table is handed to the next tile. rows = witnesses (or memory cells: u.index and producer null). Only
// the rows the lane's assigned set will hold move (init_assigned_kernel below: initial witnesses and outputs of opcodes in front of the lane's
// event): what the level path wrote behind the event means nothing, the exact kernels write those rows before they read them, and an
// instance that fails at its first constraint -- the common failure -- costs a column of zero stores instead of a column of scattered reads.
__global__ void __launch_bounds__(256) gather_columns_kernel(uint4 *__restrict__ Wx, uint64_t Bpx, const uint4 *__restrict__ W, uint64_t Bp, uint32_t n_rows,
                                                             const uint32_t *__restrict__ slow_ids, uint32_t n_slow, const uint32_t *__restrict__ unscale_index,
                                                             const uint32_t *__restrict__ unscale_consts, const uint32_t *__restrict__ producer,
                                                             const uint32_t *__restrict__ start_opcode) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_rows * n_slow) return;
    const uint32_t w = (uint32_t)(i / n_slow), t = (uint32_t)(i % n_slow);
    if (producer) {
        const uint32_t pr = producer[w];
        if (pr != 0xFFFFFFFEu && pr >= start_opcode[t]) {
            fr_store(Wx, w, Bpx, t, fr_zero());
            return;
        }
    }
    Fr x = fr_load(W, w, Bp, slow_ids[t]);
    const uint32_t ui = unscale_index ? unscale_index[w] : 0xFFFFFFFFu;
    if (ui != 0xFFFFFFFFu) x = fr_mul(x, fr_const(unscale_consts, ui));
    fr_store(Wx, w, Bpx, t, x);
}
void launch_gather_columns(hipStream_t s, uint4 *Wx, uint64_t Bpx, const uint4 *W, uint64_t Bp, uint32_t n_rows, const uint32_t *slow_ids, uint32_t n_slow,
                           const uint32_t *unscale_index, const uint32_t *unscale_consts, const uint32_t *producer, const uint32_t *start_opcode) {
    const uint64_t n = (uint64_t)n_rows * n_slow;
    if (!n) return;
    hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wx, Bpx, W, Bp, n_rows, slow_ids, n_slow, unscale_index, unscale_consts,
                       producer, start_opcode);
}
void launch_gather_initial(hipStream_t s, uint4 *Wx, uint64_t Bpx, const uint4 *W, uint64_t Bp, const uint32_t *init_ids, const uint32_t *init_rows, uint32_t n_init,
                           const uint32_t *slow_ids, uint32_t n_slow) {
    const uint64_t n = (uint64_t)n_init * n_slow;
    if (!n) return;
    hipLaunchKernelGGL(gather_initial_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wx, Bpx, W, Bp, init_ids, init_rows, n_init, slow_ids, n_slow);
}

// ------------------------------------------------------------------------------------------ helpers
__global__ void fill_u32_kernel(uint32_t *p, uint32_t v, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void min_u32_kernel(uint32_t *p, uint32_t v, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] > v) p[i] = v;
}
// assigned-set initialisation for the exact kernel. Opcodes before an instance's first event behaved exactly
// like the generic plan on exact data, so their outputs (producer[w] < start) are kept; the in-order kernel
// resumes at the event. producer: 0xFFFFFFFE = initial witness, 0xFFFFFFFF = never assigned.
__global__ void init_assigned_kernel(uint32_t *assigned, uint32_t n_slow, uint32_t n_words, uint32_t n_witnesses,
                                     const uint32_t *__restrict__ producer, const uint32_t *__restrict__ start_opcode) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_slow * n_words) return;
    const uint32_t word = (uint32_t)(i / n_slow), t = (uint32_t)(i % n_slow);
    const uint32_t start = start_opcode[t];
    uint32_t bits = 0;
    for (uint32_t k = 0; k < 32; k++) {
        const uint32_t w = word * 32 + k;
        if (w < n_witnesses) {
            const uint32_t pr = producer[w];
            if (pr == 0xFFFFFFFEu || pr < start) bits |= 1u << k;
        }
    }
    assigned[i] = bits;
}

// ------------------------------------------------------------------------------------------ launchers
bool launch_import(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint8_t *in, const uint32_t *ids, uint32_t n_in, const uint32_t *gate,
                   const uint32_t *plane_of_input, uint32_t *plane, uint32_t *event_reset) {
    if (!B || !n_in) return false;
    const dim3 grid((B + 63u) / 64u, (n_in + 3u) / 4u);
    if (((uintptr_t)in & 15u) == 0) hipLaunchKernelGGL(import_witness_kernel<true>, grid, dim3(256), 0, s, W, Bp, B, in, ids, n_in, gate, plane_of_input, plane, event_reset);
    else hipLaunchKernelGGL(import_witness_kernel<false>, grid, dim3(256), 0, s, W, Bp, B, in, ids, n_in, gate, plane_of_input, plane, event_reset);
    return event_reset != nullptr;
}
void launch_export(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, const uint32_t *sel, uint32_t n_sel, uint8_t *out,
                   const Unscale &u, const uint32_t *row_of) {
    if (!n || !n_sel) return;
    // gridDim.y is limited to 65535
    for (uint32_t done = 0; done < n_sel; done += 65535u) {
        const uint32_t m = n_sel - done > 65535u ? 65535u : n_sel - done;
        hipLaunchKernelGGL(export_witness_kernel, dim3((n + 255) / 256, m), dim3(256), 0, s, W, Bp, first, n, sel, n_sel, out, u, done, row_of);
    }
}
void launch_unscale_slow(hipStream_t s, uint4 *W, uint64_t Bp, const uint32_t *slow_ids, uint32_t n_slow, const Unscale &u, const uint32_t *producer,
                         const uint32_t *start_opcode) {
    const uint64_t n = (uint64_t)n_slow * u.n_scaled;
    if (!n) return;
    hipLaunchKernelGGL(unscale_slow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, Bp, slow_ids, n_slow, u.scaled_ids, u.n_scaled,
                       u.consts, producer, start_opcode);
}
void launch_arith_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint32_t *gate_stream, const uint32_t *gate_offset,
                        uint32_t n_gates, const uint32_t *consts, uint32_t *event, const uint4 *inv) {
    // gridDim.y is limited to 65535
    for (uint32_t done = 0; done < n_gates;) {
        uint32_t n = n_gates - done > 65535u ? 65535u : n_gates - done;
        hipLaunchKernelGGL(arith_level_kernel, dim3((B + ARITH_BLOCK - 1) / ARITH_BLOCK, n), dim3(ARITH_BLOCK), 0, s, W, Bp, B, gate_stream, gate_offset + done, consts, event, inv);
        done += n;
    }
}
void launch_inverse_batch(hipStream_t s, const uint4 *W, uint4 *inv, uint64_t Bp, uint32_t B, const uint32_t *gate_stream,
                          const uint32_t *job_offset, uint32_t n_jobs, uint32_t *event, uint32_t inv_chunk) {
    if (!n_jobs || !B) return;
    // jobs per wave: at most inv_chunk (the PLAN's snapshot of the tuning, like every other knob of a handle: one field inversion, ~13 500
    // instructions, is shared by a wave's jobs: ~1 000 each), spread evenly over the waves
    const uint32_t cap = std::min<uint32_t>(std::max<uint32_t>(inv_chunk, 1), 65536), n_chunks = (n_jobs + cap - 1) / cap;
    const uint32_t chunk = (n_jobs + n_chunks - 1) / n_chunks;
    for (uint32_t done = 0; done < n_jobs;) {  // (grid.y holds 65 535 chunks)
        const uint32_t n = (uint32_t)std::min<uint64_t>(n_jobs - done, (uint64_t)chunk * 65535u);
        hipLaunchKernelGGL(inverse_batch_kernel, dim3((B + 63) / 64, (n + chunk - 1) / chunk), dim3(64), 0, s, W, inv, Bp, B, gate_stream, job_offset + done, n, chunk, event);
        done += n;
    }
}
void launch_fr_selftest(hipStream_t s, uint64_t seed, uint32_t n, uint32_t *mismatches) {
    if (!n) return;
    hipLaunchKernelGGL(fr_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, s, seed, n, mismatches);
}
// event words of a batch (ops_common.hpp flag_instance): [0, B) = per instance the first opcode that left the generic path (0xFFFFFFFF: none); in front
// of them the count of flagged instances, kept by the kernels that flag
__global__ void __launch_bounds__(256) event_reset_kernel(uint32_t *__restrict__ event, uint32_t B) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) event[i] = 0xFFFFFFFFu;
    else if (i < B + 2) event[(int)(i - B) - 4] = 0u;  // the count of flagged instances and the spare word in front (ops_common.hpp flag_instance)
}
// a plan the level kernels do not cover entirely: every instance continues on the exact path from `opcode` at the latest
__global__ void __launch_bounds__(256) event_truncate_kernel(uint32_t *__restrict__ event, uint32_t B, uint32_t opcode) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) event[i] = min(event[i], opcode);
    if (i == 0) event[-4] = B;
}
void launch_event_reset(hipStream_t s, uint32_t *event, uint32_t B) {
    hipLaunchKernelGGL(event_reset_kernel, dim3((B + 2) / 256 + 1), dim3(256), 0, s, event, B);
}
void launch_event_truncate(hipStream_t s, uint32_t *event, uint32_t B, uint32_t opcode) {
    if (B) hipLaunchKernelGGL(event_truncate_kernel, dim3((B + 255) / 256), dim3(256), 0, s, event, B, opcode);
}
void launch_fill_u32(hipStream_t s, uint32_t *p, uint32_t v, uint64_t n) {
    if (!n) return;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
}
void launch_min_u32(hipStream_t s, uint32_t *p, uint32_t v, uint64_t n) {
    if (!n) return;
    hipLaunchKernelGGL(min_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
}
void launch_init_assigned(hipStream_t s, uint32_t *assigned, uint32_t n_slow, uint32_t n_words, uint32_t n_witnesses,
                          const uint32_t *producer, const uint32_t *start_opcode) {
    uint64_t n = (uint64_t)n_slow * n_words;
    if (!n) return;
    hipLaunchKernelGGL(init_assigned_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, assigned, n_slow, n_words, n_witnesses,
                       producer, start_opcode);
}

}  // namespace acvm
