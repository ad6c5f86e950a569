// kernels_grumpkin.hip -- FixedBaseScalarMul / Pedersen / SchnorrVerify opcodes on Grumpkin (device routines in
// ops_grumpkin.hpp), level kernel + exact kernel. Integer-ALU bound, one wave per SIMD at 2^16 instances. Workgroups of FOUR waves (round 4;
// one wave per workgroup before): the four waves of a CU start together and run the same straight-line code -- 0.6-1 MB of it against a 64 KiB
// instruction cache per two CUs -- in step. Measured on config 4 (tools/gpu_r04k.sh, gpu_r04l.sh, ten runs each, interleaved on one box): one wave per
// workgroup 2.44-2.51 ms per step in its fast mode and 2.84-3.06 ms in every third run (profiles/r04_import_effect.txt); four waves 2.44-2.63 ms, every run.
// A barrier per ladder window on top of it: 2 % slower.
#include "ops_grumpkin.hpp"
#include "ops_kernel.hpp"
#include "tuning.hpp"

namespace acvm {

struct GrumpkinOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *scratch, SlowResult *, const ExactLanes *, uint32_t) {
        return dispatch_grumpkin(p, rec, dp.grumpkin, scratch);
    }
};

void launch_grumpkin_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                           const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch) {
    launch_record_level<GrumpkinOp, 256>(s, W, Bp, B, dp, offsets, scratch_off, n, event, scratch);
}

// ---------------------------------------------------------------------------------------------- Pedersen, 4 waves per instance group
// One Pedersen record is a chain of (n + 1) hash_pairs, each 2 x 29 dependent table additions plus a normalisation: a single
// wave per 64 instances is issue-bound on that chain and a level rarely holds enough records to fill 1024 SIMDs. Here
//  * the pair table (GrumpkinTables::ped2, 503 MB of the 288 GB) holds, per generator, the sum of the two points that two
//    consecutive 9-bit slices select (the even slice already through the endomorphism): 2 x 15 additions per hash_pair, and
//  * a workgroup of four waves serves 64 instances: wave w accumulates generators [0, 8) or [8, 15) (w & 1) of the left or
//    right operand (w >> 1), the partial points meet in LDS, one wave adds them, normalises (one inversion) and
//    publishes the x coordinate that seeds the next hash_pair.
// FastPolicy only (level schedule); flagged instances take the one-lane exact kernel on the small tables (same group
// elements, so the affine results are identical).
// WAVES = 1: the same chain on ONE wave per 64 instances, for launches of more than 512 groups (half the SIMDs hold a wave of it): no
// partial points to combine -- the three Jacobian additions of every step are 12 % of a commitment's products -- no LDS, no barriers.
// Alone on the device (tools/t_pedersen_sweep.py): 8 records x 2^16 instances 3.41 -> 2.61 ms, 1 record x 2^16 0.56 -> 0.44 ms, equal at 512
// groups, and below that the four-wave form wins on latency (64 groups: 0.22 against 0.37 ms).
// Table additions of one operand of a hash_pair onto acc: windows [i0, i1) of the value's GRUMPKIN_PEDW_WINDOWS (the window table: GRUMPKIN_PEDW_BITS bits of
// the scalar per addition), or -- without that table -- generators [i0, i1) of its 15 through the pair table (one entry per 18 bits: even slice through the
// endomorphism + odd slice). parity 0: the left operand's half of the tables. Shared by the level Pedersen kernels below.
__device__ __forceinline__ GJac pedersen_walk(const GrumpkinTables &T, GJac acc, const Fr &src, uint32_t parity, uint32_t i0, uint32_t i1) {
    const Fr v = fr_to_canonical(src);
    if (T.pedw != nullptr) {
        for (uint32_t jw = i0; jw < i1; jw++)
            acc = gj_add_aff(acc, gaff_load(T.pedw, ((parity * GRUMPKIN_PEDW_WINDOWS + jw) << GRUMPKIN_PEDW_BITS) | bits_at(v, GRUMPKIN_PEDW_BITS * jw, GRUMPKIN_PEDW_BITS)));
        return acc;
    }
    const uint32_t gen0 = parity ? 15u : 0u;
    for (uint32_t i = i0; i < i1; i++) {
        const uint32_t a = bits_at(v, 18u * i, 9), b = i < 14u ? bits_at(v, 18u * i + 9u, 9) : 0u;
        acc = gj_add_aff(acc, gaff_load(T.ped2, ((gen0 + i) << GRUMPKIN_PED2_LOG2) | a << 9 | b));
    }
    return acc;
}
// (WAVES = 1 is launched in workgroups of PED1_GROUP independent waves)
constexpr int PED1_GROUP = 4;
template <int WAVES>
__global__ void __launch_bounds__(WAVES == 1 ? 64 * PED1_GROUP : 64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8)))
pedersen_quad_level_kernel(uint4 *W, uint64_t Bp, uint32_t B, DeviceProgram dp, const uint32_t *__restrict__ offsets, uint32_t *__restrict__ event,
                           const uint32_t *__restrict__ prog, const uint32_t *__restrict__ slot_of, uint32_t prio) {
    __shared__ uint32_t lds_acc[WAVES == 4 ? 4 : 1][WAVES == 4 ? 27 : 1][64];  // [wave][limb of X, Y, Z (9 x 29-bit each)][lane]
    __shared__ uint32_t lds_r[WAVES == 4 ? 16 : 1][64];                         // affine result of the step (Montgomery limbs of x, y)
    if (prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t wave_in_group = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;  // a scalar: loop bounds and record words indexed by it stay scalar
    const uint32_t wave = WAVES == 1 ? 0u : wave_in_group;
    const uint64_t j = WAVES == 1 ? ((uint64_t)blockIdx.x * PED1_GROUP + wave_in_group) * 64 + lane : (uint64_t)blockIdx.x * 64 + lane;
    const bool active = j < B;
    const uint32_t *__restrict__ rec = prog + offsets[blockIdx.y];  // (noalias arguments: scalar loads, see ops_kernel.hpp)
    const uint32_t n = rec[3];
    const uint32_t *ws = rec + 8;
    const GrumpkinTables &T = dp.grumpkin;
    FastPolicy p{W, Bp, j, slot_of};
    if (n == 0) {  // the point at infinity is reported as (0, 0)
        if (wave == 0 && active && (!p.insert(rec[4], fr_zero(), rec[5]) || !p.insert(rec[6], fr_zero(), rec[7]))) flag_instance(event, j, rec[1]);
        return;
    }
    // The first link of the chain, hash_pair(IV[domain separator], n), is the same for every instance, and so is the left half
    // of the second one, hash_single(x of the first link): the seed table (pedersen_seed_kernel, once per batch) holds that
    // point. Step 1 therefore only walks the 15 generators of the first input (4, 4, 4, 3 per wave) and adds the seed point;
    // from step 2 on, wave w takes generators [0, 8) or [8, 15) (w & 1) of the running value or of the next input (w >> 1).
    Fr r = fr_zero(), y = fr_zero();
    const uint32_t n_units = T.pedw != nullptr ? GRUMPKIN_PEDW_WINDOWS : 15u;  // table additions per operand (pedersen_walk)
    auto walk = [&](GJac acc, const Fr &src, uint32_t parity, uint32_t i0, uint32_t i1) { return pedersen_walk(T, acc, src, parity, i0, i1); };
    for (uint32_t step = 1; step <= n; step++) {
        if constexpr (WAVES == 1) {
            // (ONE call site of the walk: two make the compiler keep it as a function, whose by-value points travel through the lane's scratch)
            GJac s = gj_inf();
#pragma unroll 1
            for (uint32_t pass = 0; pass < 2; pass++) {
                if (pass == 1u && step == 1u) {
                    s = gj_add_aff(s, GAff{fr_const(dp.ped_seed, 2 * ws[n]), fr_const(dp.ped_seed, 2 * ws[n] + 1)});
                    break;
                }
                Fr src = r;
                if (pass == 0u) src = active ? p.load(ws[step - 1]) : fr_one();
                s = walk(s, src, pass ^ 1u, 0u, n_units);
            }
            bool inf;
            const GAff a = gj_to_aff(s, &inf);
            r = a.x;
            y = a.y;
        } else {
            const uint32_t parity = step == 1 ? 1u : wave >> 1;
            const uint32_t q4 = (n_units + 3u) / 4u, h2 = (n_units + 1u) / 2u;  // 15 generators: 4, 4, 4, 3 and 8, 7; 12 windows: 3 each and 6, 6
            const uint32_t i0 = step == 1 ? q4 * wave : (wave & 1u ? h2 : 0u), i1 = step == 1 ? (wave == 3 ? n_units : q4 * wave + q4) : (wave & 1u ? n_units : h2);
            Fr src = r;
            if (parity) src = active ? p.load(ws[step - 1]) : fr_one();
            const GJac acc = walk(gj_inf(), src, parity, i0, i1);
#pragma unroll
            for (int k = 0; k < 9; k++) {
                lds_acc[wave][k][lane] = acc.X.v[k];
                lds_acc[wave][9 + k][lane] = acc.Y.v[k];
                lds_acc[wave][18 + k][lane] = acc.Z.v[k];
            }
            __syncthreads();
            // the serial tail of the step (three additions, one inversion) rotates over the four waves
            if (wave == ((blockIdx.x + blockIdx.y + step) & 3u)) {
                GJac s = acc;
                if (step == 1) s = gj_add_aff(s, GAff{fr_const(dp.ped_seed, 2 * ws[n]), fr_const(dp.ped_seed, 2 * ws[n] + 1)});
                for (uint32_t dw = 1; dw < 4; dw++) {
                    const uint32_t w2 = (wave + dw) & 3u;
                    GJac o;
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        o.X.v[k] = lds_acc[w2][k][lane];
                        o.Y.v[k] = lds_acc[w2][9 + k][lane];
                        o.Z.v[k] = lds_acc[w2][18 + k][lane];
                    }
                    s = gj_add(s, o);
                }
                bool inf;
                const GAff a = gj_to_aff(s, &inf);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    lds_r[k][lane] = a.x.v[k];
                    lds_r[8 + k][lane] = a.y.v[k];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; k++) {
                r.v[k] = lds_r[k][lane];
                y.v[k] = lds_r[8 + k][lane];
            }
            __syncthreads();  // lds_r / lds_acc are rewritten by the next step
        }
    }
    if (wave == 0 && active && (!p.insert(rec[4], r, rec[5]) || !p.insert(rec[6], y, rec[7]))) flag_instance(event, j, rec[1]);
}

// ---------------------------------------------------------------------------------------------- Pedersen, K records per wave, ONE inversion per step
// A link of the chain ends in one field inversion (safegcd: ~13 500 instructions, against ~2 700 for each of its 12-24 table additions): a sixth of the
// link, a fifth of a two-input commitment. A wave pays it whatever its lanes do, so batching across LANES buys nothing; batching across RECORDS does: here
// a wave walks up to PED_BUNDLE independent Pedersen records of the launch in lock-step -- step s of every record, then ONE inversion of the product of
// their Z coordinates (Montgomery's trick: 3 products per record), then step s + 1. The Jacobian sums, the prefix products and the x coordinate that
// seeds the next step wait in the record's scratch rows ([word][instance], coalesced; PEDERSEN_PARK_WORDS per instance and record). Records of
// different input counts may share a bundle: the shorter ones simply sit out the later steps. Same table additions, same affine results bit for bit
// (the inverse of a field element is unique). launch_pedersen_level picks the records per wave so that the launch keeps its share of the chip.
constexpr uint32_t PED_BUNDLE = 8;
constexpr uint32_t PED_PARK_PREFIX = 27, PED_PARK_X = 36;
static_assert(PEDERSEN_PARK_WORDS == PED_PARK_X + 8, "scratch map of pedersen_bundle_level_kernel");
__global__ void __launch_bounds__(64 * PED1_GROUP) __attribute__((amdgpu_waves_per_eu(4, 8)))
pedersen_bundle_level_kernel(uint4 *W, uint64_t Bp, uint32_t B, DeviceProgram dp, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ soff,
                             uint32_t n_records, uint32_t K, uint32_t *__restrict__ event, const uint32_t *__restrict__ prog, const uint32_t *__restrict__ slot_of,
                             uint32_t prio, uint32_t *__restrict__ scratch) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t wave_in_group = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint64_t j0 = ((uint64_t)blockIdx.x * PED1_GROUP + wave_in_group) * 64;
    if (j0 >= B) return;  // (a whole wave beyond the batch: its lanes have no scratch rows; the kernel has no barrier)
    const uint64_t j = j0 + lane;  // < Bp: rows are padded to a multiple of 64 instances
    const bool active = j < B;
    const uint32_t b0 = blockIdx.y * K, nb = min(K, n_records - b0);  // K <= PED_BUNDLE records per wave
    const GrumpkinTables &T = dp.grumpkin;
    FastPolicy p{W, Bp, j, slot_of};
    const uint32_t n_units = T.pedw != nullptr ? GRUMPKIN_PEDW_WINDOWS : 15u;
    auto walk = [&](GJac acc, const Fr &src, uint32_t parity) { return pedersen_walk(T, acc, src, parity, 0u, n_units); };
    auto park = [&](uint32_t i, uint32_t w) -> uint32_t * { return scratch + ((uint64_t)soff[2u * (b0 + i)] + w) * Bp + j; };
    auto park_get29 = [&](uint32_t i, uint32_t w0) { Fr29 r; for (int k = 0; k < 9; k++) r.v[k] = *park(i, w0 + k); return r; };
    auto park_put29 = [&](uint32_t i, uint32_t w0, const Fr29 &v) { for (int k = 0; k < 9; k++) *park(i, w0 + k) = v.v[k]; };
    uint32_t n_max = 0;
    for (uint32_t i = 0; i < nb; i++) {
        const uint32_t *__restrict__ rec = prog + offsets[b0 + i];
        n_max = max(n_max, rec[3]);
        if (rec[3] == 0u && active && (!p.insert(rec[4], fr_zero(), rec[5]) || !p.insert(rec[6], fr_zero(), rec[7]))) flag_instance(event, j, rec[1]);  // the point at infinity is reported as (0, 0)
    }
    for (uint32_t step = 1; step <= n_max; step++) {
        // ---- forward: the step's sum of every record that still runs, and the running product of their Z
        Fr29 prefix = g29_one();
#pragma unroll 1
        for (uint32_t i = 0; i < nb; i++) {
            const uint32_t *__restrict__ rec = prog + offsets[b0 + i];
            const uint32_t n = rec[3];
            if (step > n) continue;
            const uint32_t *ws = rec + 8;
            Fr r = fr_zero();
            if (step > 1u)
                for (int k = 0; k < 8; k++) r.v[k] = *park(i, PED_PARK_X + k);
            // (ONE call site of the walk, as in pedersen_quad_level_kernel<1>)
            GJac s = gj_inf();
#pragma unroll 1
            for (uint32_t pass = 0; pass < 2; pass++) {
                if (pass == 1u && step == 1u) {
                    s = gj_add_aff(s, GAff{fr_const(dp.ped_seed, 2 * ws[n]), fr_const(dp.ped_seed, 2 * ws[n] + 1)});
                    break;
                }
                Fr src = r;
                if (pass == 0u) src = active ? p.load(ws[step - 1]) : fr_one();
                s = walk(s, src, pass ^ 1u);
            }
            park_put29(i, 0, s.X);
            park_put29(i, 9, s.Y);
            park_put29(i, 18, s.Z);
            prefix = fr29_mul(prefix, gj_is_inf(s) ? g29_one() : s.Z);  // (a sum at infinity leaves the product alone and reports (0, 0))
            park_put29(i, PED_PARK_PREFIX, prefix);
        }
        // ---- one inversion for all of them, handed back record by record
        Fr29 inv = fr29_from(fr_inv(fr29_pack(fr29_canon(prefix))));
#pragma unroll 1
        for (uint32_t i = nb; i-- > 0u;) {
            const uint32_t *__restrict__ rec = prog + offsets[b0 + i];
            const uint32_t n = rec[3];
            if (step > n) continue;
            uint32_t ip = 0xFFFFFFFFu;  // the record in front of this one that runs this step too
            for (uint32_t q = i; q-- > 0u;)
                if (step <= prog[offsets[b0 + q] + 3]) { ip = q; break; }
            const Fr29 X = park_get29(i, 0), Y = park_get29(i, 9), Z = park_get29(i, 18);
            const bool inf = fr29_is_zero_mod_p(Z);
            const Fr29 zi = fr29_mul(inv, ip == 0xFFFFFFFFu ? g29_one() : park_get29(ip, PED_PARK_PREFIX));
            inv = fr29_mul(inv, inf ? g29_one() : Z);
            const Fr29 zi2 = fr29_sqr(zi);
            Fr x = fr29_pack(fr29_canon(fr29_mul(X, zi2))), y = fr29_pack(fr29_canon(fr29_mul(Y, fr29_mul(zi2, zi))));
            if (inf) { x = fr_zero(); y = fr_zero(); }
            for (int k = 0; k < 8; k++) *park(i, PED_PARK_X + k) = x.v[k];
            if (step == n && active && (!p.insert(rec[4], x, rec[5]) || !p.insert(rec[6], y, rec[7]))) flag_instance(event, j, rec[1]);
        }
    }
}

// seed table of the level Pedersen kernel: one lane per Pedersen record, keys = (n, domain separator) pairs; a row is the
// affine point hash_single(x of hash_pair(IV, n), parity 0), 16 x u32
__global__ void __launch_bounds__(64) pedersen_seed_kernel(GrumpkinTables T, const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t n_in = keys[2 * i], hash_index = keys[2 * i + 1];
    const GAff first = pedersen_hash_pair(T, pedersen_iv_x(T, hash_index), fr_from_u32(n_in));
    bool inf;
    const GAff a = gj_to_aff(pedersen_hash_single(T, fr_to_canonical(first.x), 0), &inf);  // left half of the second link
#pragma unroll
    for (int k = 0; k < 8; k++) {
        out[16 * i + k] = a.x.v[k];
        out[16 * i + 8 + k] = a.y.v[k];
    }
}
void launch_pedersen_seeds(hipStream_t s, const GrumpkinTables &T, const uint32_t *keys, uint32_t n, uint32_t *out) {
    if (n) hipLaunchKernelGGL(pedersen_seed_kernel, dim3((n + 63) / 64), dim3(64), 0, s, T, keys, n, out);
}

// pair table of the level Pedersen kernel (grumpkin_host.hpp GrumpkinTables::ped2): one lane per entry
__global__ void __launch_bounds__(64) pedersen_pair_table_kernel(GrumpkinTables T, uint4 *__restrict__ out) {
    const uint32_t e = blockIdx.x * 64 + threadIdx.x;
    const uint32_t gen = e >> GRUMPKIN_PED2_LOG2, a = (e >> 9) & 511u, b = e & 511u;
    GAff pa = gaff_load(T.ped, gen * GRUMPKIN_PED_ENTRIES + a);
    pa.x = fr_mul(pa.x, grumpkin_beta());  // (x, y) -> (beta x, y)
    GAff r = pa;
    if (gen % 15u != 14u) {
        bool inf;
        r = gj_to_aff(gj_add_aff(gj_add_aff(gj_inf(), pa), gaff_load(T.ped, gen * GRUMPKIN_PED_ENTRIES + b)), &inf);
    }
    uint4 *p = out + (uint64_t)e * 4;
    p[0] = make_uint4(r.x.v[0], r.x.v[1], r.x.v[2], r.x.v[3]);
    p[1] = make_uint4(r.x.v[4], r.x.v[5], r.x.v[6], r.x.v[7]);
    p[2] = make_uint4(r.y.v[0], r.y.v[1], r.y.v[2], r.y.v[3]);
    p[3] = make_uint4(r.y.v[4], r.y.v[5], r.y.v[6], r.y.v[7]);
}
// 16-bit window tables (GrumpkinTables::win16): entry d of window w = d * 2^(16 w) * P = the sum of the two 8-bit window entries of its
// bytes; one lane per entry, affine by one inversion
__global__ void __launch_bounds__(64) grumpkin_win16_table_kernel(GrumpkinTables T, uint4 *__restrict__ out) {
    const uint64_t e = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (e >= (uint64_t)GRUMPKIN_N_WINDOW_BASES * GRUMPKIN_WIN16_STRIDE) return;
    const uint32_t base = (uint32_t)(e / GRUMPKIN_WIN16_STRIDE), rem = (uint32_t)(e % GRUMPKIN_WIN16_STRIDE);
    const uint32_t w = rem / 65535u, d = rem % 65535u + 1u, lo = d & 255u, hi = d >> 8;
    const uint4 *tbl = T.win + (uint64_t)base * GRUMPKIN_WIN_STRIDE * 4;
    GJac acc = gj_inf();
    if (lo) acc = gj_add_aff(acc, gaff_load(tbl, (2u * w) * 255u + lo - 1u));
    if (hi) acc = gj_add_aff(acc, gaff_load(tbl, (2u * w + 1u) * 255u + hi - 1u));
    bool inf;
    const GAff r = gj_to_aff(acc, &inf);
    uint4 *p = out + e * 4;
    p[0] = make_uint4(r.x.v[0], r.x.v[1], r.x.v[2], r.x.v[3]);
    p[1] = make_uint4(r.x.v[4], r.x.v[5], r.x.v[6], r.x.v[7]);
    p[2] = make_uint4(r.y.v[0], r.y.v[1], r.y.v[2], r.y.v[3]);
    p[3] = make_uint4(r.y.v[4], r.y.v[5], r.y.v[6], r.y.v[7]);
}
void launch_grumpkin_win16_table(hipStream_t s, const GrumpkinTables &T, uint4 *out) {
    const uint64_t n = (uint64_t)GRUMPKIN_N_WINDOW_BASES * GRUMPKIN_WIN16_STRIDE;
    hipLaunchKernelGGL(grumpkin_win16_table_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, T, out);
}
// window table of the level Pedersen kernel (grumpkin_host.hpp GrumpkinTables::pedw): one lane per entry [parity][window j][v]
__global__ void __launch_bounds__(64) pedersen_window_table_kernel(GrumpkinTables T, uint4 *__restrict__ out, uint32_t *__restrict__ n_infinite) {
    const uint64_t e = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint32_t v = (uint32_t)e & ((1u << GRUMPKIN_PEDW_BITS) - 1u), pj = (uint32_t)(e >> GRUMPKIN_PEDW_BITS), parity = pj / GRUMPKIN_PEDW_WINDOWS, j = pj % GRUMPKIN_PEDW_WINDOWS;
    const uint32_t w_lo = GRUMPKIN_PEDW_BITS * j, w_hi = w_lo + GRUMPKIN_PEDW_BITS;
    GJac acc = gj_inf();
    for (uint32_t s = w_lo / 9u; s < 29u && 9u * s < w_hi; s++) {  // the slices (9 bits each: a_0, b_0, a_1, ..., a_14) the window touches
        const uint32_t lo = max(9u * s, w_lo), hi = min(9u * s + 9u, w_hi);
        const uint32_t piece = ((v >> (lo - w_lo)) & ((1u << (hi - lo)) - 1u)) << (lo - 9u * s);  // its bits at their place in the slice
        const bool starts_here = lo == 9u * s;                                                     // then the slice's `+ 1` rides in this window
        if (!starts_here && piece == 0u) continue;
        GAff pt = gaff_load(T.ped, (15u * parity + s / 2u) * GRUMPKIN_PED_ENTRIES + (starts_here ? piece : piece - 1u));  // (k + 1) D at index k
        if ((s & 1u) == 0u) pt.x = fr_mul(pt.x, grumpkin_beta());  // the even slices go through the endomorphism (x, y) -> (beta x, y)
        acc = gj_add_aff(acc, pt);
    }
    bool inf;
    const GAff r = gj_to_aff(acc, &inf);
    if (inf) atomicAdd(n_infinite, 1u);
    uint4 *p = out + e * 4;
    p[0] = make_uint4(r.x.v[0], r.x.v[1], r.x.v[2], r.x.v[3]);
    p[1] = make_uint4(r.x.v[4], r.x.v[5], r.x.v[6], r.x.v[7]);
    p[2] = make_uint4(r.y.v[0], r.y.v[1], r.y.v[2], r.y.v[3]);
    p[3] = make_uint4(r.y.v[4], r.y.v[5], r.y.v[6], r.y.v[7]);
}
void launch_pedersen_window_table(hipStream_t s, const GrumpkinTables &T, uint4 *out, uint32_t *n_infinite) {
    const uint64_t n = (uint64_t)2 * GRUMPKIN_PEDW_WINDOWS << GRUMPKIN_PEDW_BITS;
    hipLaunchKernelGGL(pedersen_window_table_kernel, dim3((unsigned)(n / 64)), dim3(64), 0, s, T, out, n_infinite);
}
void launch_pedersen_pair_table(hipStream_t s, const GrumpkinTables &T, uint4 *out) {
    hipLaunchKernelGGL(pedersen_pair_table_kernel, dim3((30u << GRUMPKIN_PED2_LOG2) / 64), dim3(64), 0, s, T, out);
}

void launch_pedersen_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                           uint32_t *event, const uint32_t *scratch_off, uint32_t *scratch) {
    if (!n || !B) return;
    const uint64_t groups = (uint64_t)((B + 63u) / 64u) * n;  // one per 64 instances of a record
    const int64_t mode = tuning().pedersen_waves;               // 0: by the size of the launch; 1 / 4: forced (A/B measurements)
    const uint32_t prio = (uint32_t)tuning().pedersen_prio;
    const bool one = mode == 1 || (mode != 4 && groups > 512u);  // (measured alone, tools/t_pedersen_sweep.py: equal at 512 groups, one wave 21-24 % faster from 1 024 on)
    // several records in the launch: K of them per wave share the inversions, K as large as leaves pedersen_bundle_waves waves (the launch runs
    // beside the gate kernel: fewer instructions in total, but each wave's walk is K times as long)
    const int64_t bundle = tuning().pedersen_bundle;  // 0: never; 1: where it pays; 2: every launch of two records or more (tests)
    uint32_t K = 1;
    if (n >= 2u && scratch && bundle == 2) K = PED_BUNDLE;
    else if (n >= 2u && scratch && bundle == 1 && one) K = (uint32_t)std::min<uint64_t>(PED_BUNDLE, groups / (uint64_t)std::max<int64_t>(tuning().pedersen_bundle_waves, 1));
    if (K >= 2u) {
        K = (n + (n + K - 1) / K - 1) / ((n + K - 1) / K);  // the same number of waves, records spread evenly (9 records: 5 + 4, not 8 + 1)
        for (uint32_t done = 0; done < n; done += 65535u * K) {
            const uint32_t m = n - done > 65535u * K ? 65535u * K : n - done;
            hipLaunchKernelGGL(pedersen_bundle_level_kernel, dim3((B + 64 * PED1_GROUP - 1) / (64 * PED1_GROUP), (m + K - 1) / K), dim3(64 * PED1_GROUP), 0, s, W, Bp, B, dp,
                               offsets + done, scratch_off + 2 * (size_t)done, m, K, event, dp.prog, dp.slot_of, prio, scratch);
        }
        return;
    }
    for (uint32_t done = 0; done < n;) {
        const uint32_t m = n - done > 65535u ? 65535u : n - done;
        if (one) hipLaunchKernelGGL(pedersen_quad_level_kernel<1>, dim3((B + 64 * PED1_GROUP - 1) / (64 * PED1_GROUP), m), dim3(64 * PED1_GROUP), 0, s, W, Bp, B, dp, offsets + done, event, dp.prog, dp.slot_of, prio);
        else hipLaunchKernelGGL(pedersen_quad_level_kernel<4>, dim3((B + 63) / 64, m), dim3(256), 0, s, W, Bp, B, dp, offsets + done, event, dp.prog, dp.slot_of, prio);
        done += m;
    }
}

// ---- component probes for the parity tests (acvm_debug_grumpkin): in / out are canonical 8 x u32 little-endian
static __device__ uint32_t g_probe_window_table[GRUMPKIN_VARBASE_SCRATCH_WORDS];
__global__ void grumpkin_probe_kernel(GrumpkinTables T, uint32_t what, uint32_t param, const uint32_t *in, uint32_t n_in, uint32_t *out) {
    if (threadIdx.x || blockIdx.x) return;
    auto ld = [&](uint32_t i) { Fr c; for (int k = 0; k < 8; k++) c.v[k] = in[8 * i + k]; return c; };
    auto st = [&](uint32_t i, const Fr &m) { Fr c = fr_to_canonical(m); for (int k = 0; k < 8; k++) out[8 * i + k] = c.v[k]; };
    bool inf;
    if (what == 1) {
        GAff a = gj_to_aff(pedersen_hash_single(T, ld(0), param), &inf);
        st(0, a.x); st(1, a.y);
    } else if (what == 2) {
        GJac acc = gj_inf();
        for (uint32_t j = 0; j < n_in && j < 3; j++) acc = gj_add(acc, ladder_term(T, ld(j), j));
        st(0, gj_to_aff(acc, &inf).x);
    } else if (what == 3) {
        GAff a = gj_to_aff(fixed_base_mul(T, param, ld(0)), &inf);
        st(0, a.x); st(1, a.y);
    } else if (what == 5) {
        GJac acc0 = gj_inf();
        const Fr v = ld(0);
        for (uint32_t i = 0; i < 15; i++) acc0 = gj_add_aff(acc0, gaff_load(T.ped, (param * 15u + i) * GRUMPKIN_PED_ENTRIES + bits_at(v, 18u * i, 9)));
        GAff a = gj_to_aff(acc0, &inf);
        st(0, a.x); st(1, a.y);
    } else if (what == 6) {
        st(0, fr_mul(fr_from_canonical(ld(0)), grumpkin_beta()));
        st(1, grumpkin_beta());
    } else if (what == 7) {
        GAff p0 = gaff_load(T.ped, 0), p1 = gaff_load(T.ped, 512);
        GJac a = gj_add_aff(gj_inf(), p0), b = gj_add_aff(gj_add_aff(gj_inf(), p1), p0);
        GAff r = gj_to_aff(gj_add(a, b), &inf);
        st(0, r.x); st(1, r.y);
    } else if (what == 8 || what == 9) {  // in[0] * (in[1], in[2]): 8 = GLV + window table (SchnorrVerify's path), 9 = double-and-add
        const GAff pt{fr_from_canonical(ld(1)), fr_from_canonical(ld(2))};
        GAff a = gj_to_aff(grumpkin_var_base_mul(pt, ld(0), what == 8 ? g_probe_window_table : nullptr, 1, 0), &inf);
        st(0, a.x); st(1, a.y);
    } else if (what == 4) {
        GAff a = gaff_load(param >> 24 == 0 ? T.ped : param >> 24 == 1 ? T.win : param >> 24 == 2 ? T.small : T.skew, param & 0xffffffu);
        st(0, a.x); st(1, a.y);
    }
}
void launch_grumpkin_probe(hipStream_t s, const GrumpkinTables &T, uint32_t what, uint32_t param, const uint32_t *in, uint32_t n_in, uint32_t *out) {
    hipLaunchKernelGGL(grumpkin_probe_kernel, dim3(1), dim3(64), 0, s, T, what, param, in, n_in, out);
}

}  // namespace acvm
