// kernels.hpp -- host-visible launchers of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acvm {

struct SlowResult {
    uint32_t status, err, opcode_index, aux0, aux1;
};

void launch_import(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint8_t *in, const uint32_t *ids, uint32_t n_in);
void launch_export(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, const uint32_t *sel, uint32_t n_sel,
                   uint8_t *out);
void launch_arith_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint32_t *gate_stream, const uint32_t *gate_offset,
                        uint32_t n_gates, const uint32_t *consts, uint32_t *event);
void launch_arith_dyn_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint32_t *gate_stream, const uint32_t *dyn_offset,
                            uint32_t n_dyn, const uint32_t *consts, uint32_t *event, uint4 *scratch);
void launch_arith_inorder(hipStream_t s, uint4 *W, uint64_t Bp, const uint32_t *slow_ids, uint32_t n_slow, const uint32_t *stream,
                          const uint32_t *offset, uint32_t n_opcodes, const uint32_t *consts, uint32_t *assigned,
                          const uint32_t *start_opcode, SlowResult *results);
void launch_fr_selftest(hipStream_t s, uint64_t seed, uint32_t n, uint32_t *mismatches);
void launch_fill_u32(hipStream_t s, uint32_t *p, uint32_t v, uint64_t n);
void launch_min_u32(hipStream_t s, uint32_t *p, uint32_t v, uint64_t n);
void launch_init_assigned(hipStream_t s, uint32_t *assigned, uint32_t n_slow, uint32_t n_words, uint32_t n_witnesses,
                          const uint32_t *producer, const uint32_t *start_opcode);

}  // namespace acvm
