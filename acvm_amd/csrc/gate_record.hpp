// gate_record.hpp -- the format of a gate record in the u32 stream of the Arithmetic level program, shared by the planner that writes
// it (plan.cpp, plain C++) and the evaluation that reads it (gate_eval.hpp, host and device).
//  w0 = kind | np_mac << 8 | nl_mac << 16 | flags   w1 = opcode index (program order)   w2 = output witness row (SOLVE*)
//  w3 = constant term (index into the constant pool, or GATE_COEF_ZERO)   w4 = row of 1 / denominator in the inverse table (SOLVE_DYN)
//  w5 = np_pos | np_neg << 8 | nl_pos << 16 | nl_neg << 24, then the term lists:
//    np_mac x (coef[8], a, b), nl_mac x (coef[8], w), np_pos x (a, b), np_neg x (a, b), nl_pos x (w), nl_neg x (w)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GATE_HD __host__ __device__
#else
#define GATE_HD
#endif

namespace acvm {

// The record stream and the constant pool as the evaluation reads them. On the device they are read through the CONSTANT address space:
// nothing writes them while a kernel runs, and a load from that space with a wave-uniform address is a scalar load by definition -- not
// only where the compiler can prove that no store of the kernel reaches it (that proof is budgeted: in a kernel with many stores some
// term loops fell back to vector loads of record words, each with a vmcnt(0) behind it; tools/stream_kernel.patch).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const uint32_t __attribute__((address_space(4))) *GateWords;
#define GATE_WORDS(p) ((::acvm::GateWords)(uintptr_t)(p))
#else
typedef const uint32_t *GateWords;
#define GATE_WORDS(p) ((::acvm::GateWords)(p))
#endif

// operand row GATE_LOCAL: the output of the record that ran before this one in the same wave (plan.cpp "gate pairs")
static constexpr uint32_t GATE_LOCAL = 0xFFFFFFFFu;
static constexpr uint32_t GATE_TAIL_FLAG = 1u << 24;      // another record follows in the same wave
static constexpr uint32_t GATE_SETLOCAL_FLAG = 1u << 25;  // this record's output becomes GATE_LOCAL of the records behind it
// w0 bits 26-27: what happens to the record's result before it is stored
static constexpr uint32_t GATE_OUT_SHIFT = 26;
static constexpr uint32_t GATE_OUT_ASIS = 0;   // the normalised column scan as it is (planner: bound < 2^256)
static constexpr uint32_t GATE_OUT_WEAK = 1;   // fr29_weak: < 1.03 p
static constexpr uint32_t GATE_OUT_CANON = 2;  // canonical [0, p)
// w0 bits 28-29: the subtracted linear terms are taken from 2^k p (k = 1..3: the largest operand bound among them)
static constexpr uint32_t GATE_SUBK_SHIFT = 28;
// w0 bit 30 (SOLVE_DYN): the sum may reach 8 p: reduce it before it meets the inverse
static constexpr uint32_t GATE_PRESUM_WEAK = 1u << 30;
static constexpr uint32_t GATE_COEF_ZERO = 0xFFFFFFFDu;

// Limb weights of the lazy side sum h (units of 2^25 per limb): an added normalised operand 16, a product 17, a subtracted term 33
// (it is added as 2^k p - x with limbs below 2^30); past GATE_H_MAX the sum is reduced (fr29_weak) before the next term.
static constexpr uint32_t GATE_H_MAX = 111;
static constexpr uint32_t GATE_H_AFTER_WEAK = 17;
// the running sum of a record is reduced (fr29_weak) after every so many Montgomery reductions (each adds up to 1.34 p)
static constexpr uint32_t GATE_REDUCTIONS_PER_WEAK = 16;

// Bounds of stored values in units of p / 256 (plan.cpp kbound): canonical rows, rows after fr29_weak (1.03 p), the most a row of
// eight 32-bit words can hold (2^256 / p = 5.2903: values below 1354 p / 256 fit), the inverse table's rows (below 1.4 p).
static constexpr uint32_t GATE_K_CANON = 256, GATE_K_WEAK = 264, GATE_K_ROW_MAX = 1354, GATE_K_INVERSE = 359;
// ceil(256 * (ka / 256) (kb / 256) p / 2^261): what a product of operands below ka p / 256 and kb p / 256 adds to the result of the
// Montgomery reduction it shares (fr29_dot_impl: result < p + sum a_t b_t / 2^261; p / 2^261 < 0.0059073 < 1549 / 2^18)
GATE_HD inline uint32_t gate_k_product(uint32_t ka, uint32_t kb) {
    return (uint32_t)(((uint64_t)ka * kb * 1549u + ((1ull << 26) - 1)) >> 26);
}

// words of a gate record
GATE_HD inline uint32_t gate_record_words(GateWords g) {
    const uint32_t w0 = g[0], w5 = g[5];
    return 6u + 10u * ((w0 >> 8) & 0xff) + 9u * ((w0 >> 16) & 0xff) + 2u * ((w5 & 0xff) + ((w5 >> 8) & 0xff)) + ((w5 >> 16) & 0xff) + (w5 >> 24);
}

}  // namespace acvm
