// kernels_grumpkin.hip -- FixedBaseScalarMul / Pedersen / SchnorrVerify opcodes on Grumpkin (device routines in
// ops_grumpkin.hpp), level kernel + exact kernel. Integer-ALU bound; one wave per workgroup so that the scheduler can
// spread the long-running lanes over all SIMDs.
#include "ops_grumpkin.hpp"
#include "ops_kernel.hpp"

namespace acvm {

struct GrumpkinOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *scratch, SlowResult *, const ExactLanes *, uint32_t) {
        return dispatch_grumpkin(p, rec, dp.grumpkin, scratch);
    }
};

void launch_grumpkin_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                           const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch) {
    launch_record_level<GrumpkinOp, 64>(s, W, Bp, B, dp, offsets, scratch_off, n, event, scratch);
}
void launch_exact_grumpkin(hipStream_t s, uint4 *W, uint64_t Bp, const DeviceProgram &dp, const ExactLanes &L, uint32_t opcode, uint32_t *scratch) {
    launch_record_exact<GrumpkinOp, 64>(s, W, Bp, dp, L, opcode, scratch);
}

// ---- component probes for the parity tests (acvm_debug_grumpkin): in / out are canonical 8 x u32 little-endian
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
    } else if (what == 4) {
        GAff a = gaff_load(param >> 24 == 0 ? T.ped : param >> 24 == 1 ? T.win : param >> 24 == 2 ? T.small : T.skew, param & 0xffffffu);
        st(0, a.x); st(1, a.y);
    }
}
void launch_grumpkin_probe(hipStream_t s, const GrumpkinTables &T, uint32_t what, uint32_t param, const uint32_t *in, uint32_t n_in, uint32_t *out) {
    hipLaunchKernelGGL(grumpkin_probe_kernel, dim3(1), dim3(64), 0, s, T, what, param, in, n_in, out);
}

}  // namespace acvm
