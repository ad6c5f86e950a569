// plan.hpp -- static plan of one circuit against one set of initially-assigned witnesses.
//
// The reference interprets opcodes in program order per instance (acvm/src/pwg/mod.rs:236-303). Which
// witness an Arithmetic opcode solves depends only on WHICH witnesses are assigned, except for the
// zero-coefficient drop in ArithmeticSolver::evaluate (arithmetic.rs:217-221). The planner replays the
// assigned-set bookkeeping once ("generic" instance: no known multiplicand of an unknown is zero), folds
// every constant divisor (-(sum / coeff), arithmetic.rs:56,86,120) into the gate's coefficients and
// levelises the dependency DAG. Instances that leave the generic path are detected on the device
// (event word) and re-solved by the exact in-order kernel.
#pragma once
#include "circuit.hpp"
#include <vector>

namespace acvm {

// coefficient encoding inside the gate stream
static constexpr uint32_t COEF_ONE = 0xFFFFFFFFu;
static constexpr uint32_t COEF_MINUS_ONE = 0xFFFFFFFEu;
static constexpr uint32_t COEF_ZERO = 0xFFFFFFFDu;  // only for the constant term

enum GateKind : uint32_t { GATE_ASSERT = 0, GATE_SOLVE = 1, GATE_SOLVE_DYN = 2 };

// Gate record in the u32 stream:
//  w0 = kind | n_prod << 8 | n_lin << 16
//  w1 = opcode index (program order)      w2 = output witness slot (SOLVE*)
//  w3 = constant term (coef encoding)     w4 = denominator witness slot (SOLVE_DYN)
//  then n_prod x {coef, a, b}, n_lin x {coef, a}
static constexpr uint32_t GATE_HDR_WORDS = 5;

struct Plan {
    uint32_t n_witnesses = 0;
    uint32_t n_opcodes = 0;
    std::vector<uint32_t> initial_ids;
    // device program
    std::vector<uint32_t> gate_stream;          // all gate records
    std::vector<uint32_t> gate_offset;          // per scheduled ASSERT/SOLVE gate: offset into gate_stream (level-major)
    std::vector<uint32_t> level_start;          // size n_levels + 1, indexes gate_offset
    std::vector<uint32_t> dyn_offset;           // per scheduled SOLVE_DYN gate (needs a per-instance inversion), level-major
    std::vector<uint32_t> dyn_level_start;      // size n_levels + 1, indexes dyn_offset
    std::vector<FrH> constants;                 // Montgomery-form circuit constants
    // bookkeeping for export / failure masking
    std::vector<uint32_t> producer;             // per witness: opcode index that assigns it, 0xFFFFFFFF if none,
                                                // 0xFFFFFFFE if initial
    // first opcode that the level kernels cannot execute for the generic instance (static failure or an opcode
    // kind not yet on the fast path); 0xFFFFFFFF if the whole circuit is covered
    uint32_t truncated_at = 0xFFFFFFFFu;
    // statistics
    uint32_t n_fast_gates = 0, n_dyn_gates = 0, max_level_width = 0;
    uint64_t algorithmic_bytes = 0, arith_algorithmic_bytes = 0, dyn_algorithmic_bytes = 0;
    double plan_ms = 0;
    // in-order program for the exact kernel: per opcode offset into `slow_stream`
    std::vector<uint32_t> slow_stream;
    std::vector<uint32_t> slow_offset;
    std::string unsupported;  // non-empty: circuit holds an opcode no kernel implements yet
};

Plan build_plan(const Circuit &c, const uint32_t *initial_ids, uint32_t n_initial);

}  // namespace acvm
