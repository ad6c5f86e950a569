// scratch_layout.hpp -- sizes of per-lane device scratch that the planner (plan.cpp, plain C++: also built without HIP for the sanitizer
// runs) reserves and the device routines lay out.
#pragma once
#include <stdint.h>

namespace acvm {

// per-lane scratch words of SchnorrVerify's window table of e * pk (ops_grumpkin.hpp grumpkin_var_base_mul: 16 chain entries of 27 words,
// 16 finished rows of 16 words, 4 words of alignment slack); the planner reserves them behind the message words of the record
static constexpr uint32_t GRUMPKIN_VARBASE_SCRATCH_WORDS = 16 * 27 + 16 * 16 + 4;

// per-instance scratch words of a Pedersen record in the level schedule (kernels_grumpkin.hip pedersen_bundle_level_kernel): X, Y, Z of the
// step's sum (9 limbs each), the running product of the bundle's Z (9), the affine x that seeds the next step (8)
static constexpr uint32_t PEDERSEN_PARK_WORDS = 44;

}  // namespace acvm
