// grumpkin_host.hpp -- lookup tables of the Grumpkin kernels (built once on the host, see grumpkin_host.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scratch_layout.hpp"

namespace acvm {

static constexpr uint32_t GRUMPKIN_N_GENERATORS = 30;
static constexpr uint32_t GRUMPKIN_PED_ENTRIES = 512;     // k * D[i], k = 1..512
static constexpr uint32_t GRUMPKIN_N_WINDOW_BASES = 4;    // G, D[0], D[3], D[6]
static constexpr uint32_t GRUMPKIN_WIN_STRIDE = 32 * 255; // points per base: T[w][d-1] = d * 2^(8w) * P

// device pointers; one affine point = 4 x uint4 (x limbs 0..7, y limbs 0..7, Montgomery form)
struct GrumpkinTables {
    const uint4 *ped;    // [30][512]
    const uint4 *win;    // [4][32][255]
    const uint4 *small;  // [3][15]: k * D[3j+1], k = 1..15
    const uint4 *skew;   // [3]: D[3j+2]
    const uint4 *ped2;   // [30][512][512] pair table of the level Pedersen kernel (grumpkin_pair_table), else nullptr
    const uint4 *win16;  // [4][16][65535] 16-bit windows of the same four bases: T[w][d-1] = d * 2^(16w) * P (built on the device: 268 MB)
    const uint4 *pedw;   // [2][11][2^24] window table of the level Pedersen kernel (grumpkin_window_table), else nullptr
};
static constexpr uint32_t GRUMPKIN_WIN16_STRIDE = 16 * 65535;  // points per base
static constexpr uint32_t GRUMPKIN_PED2_LOG2 = 18;  // entries per generator

// Tables of the CURRENT device, built on first use and kept in a per-device set (grumpkin_host.cpp): a copy of the set's pointers into
// *out, false on failure. Builds take the device's own lock and synchronise the set's build stream (never the device).
bool grumpkin_tables(GrumpkinTables *out);
// the same tables with ped2 built (503 MB of HBM, generated on the device on first use): entry [g][a][b] is
// beta((a + 1) D[g]) + (b + 1) D[g], i.e. the contribution of two consecutive 9-bit slices of a plookup Pedersen
// hash_single (the even slice goes through the endomorphism), so that the level kernel pays one mixed addition per 18 bits
// instead of two. For the last generator of a value (g % 15 == 14, one slice only) the entry is beta((a + 1) D[g]).
bool grumpkin_pair_table(GrumpkinTables *out);
// ... and with pedw built (23.6 GB, generated on the device on first use; 6.4 GB with round 3's 22-bit windows). The slices of hash_single are linear in the bits of the scalar -- a slice
// a of generator D contributes (a + 1) D = a_hi 2^k D + a_lo D + D -- so the 29 slices (261 bits) of a value can be cut at ANY bit: entry [parity][j][v]
// is the joint contribution of bits [24 j, 24 j + 24) of the scalar (the pieces of the two to four slices the window touches, the even slices through
// the endomorphism, plus the `+ 1` of every slice that starts inside the window): 11 mixed additions per hash_single instead of the pair table's 15 (12 with 22-bit windows).
#ifndef GRUMPKIN_PEDW_BITS_V  // (tools/build_variant.sh -DGRUMPKIN_PEDW_BITS_V=22: round 3's 12 windows of 22 bits, a 6.4 GB table; the A/B is in NOTEBOOK.md section 9)
#define GRUMPKIN_PEDW_BITS_V 24
#endif
static constexpr uint32_t GRUMPKIN_PEDW_BITS = GRUMPKIN_PEDW_BITS_V, GRUMPKIN_PEDW_WINDOWS = (261 + GRUMPKIN_PEDW_BITS - 1) / GRUMPKIN_PEDW_BITS;
static_assert(GRUMPKIN_PEDW_BITS >= 9 && GRUMPKIN_PEDW_BITS <= 26, "window of the level Pedersen kernel");
// (tuning.cpp accepts pedersen_window_bits = 0 or 24: a build with another width changes that line too)
bool grumpkin_window_table(GrumpkinTables *out);
bool grumpkin_host_point(uint32_t which, uint32_t index, uint8_t out_be[64]);
// generator tables of the two ECDSA curves on the current device (kernels_ecdsa.hip builds them), nullptr on failure
const uint32_t *ecdsa_generator_tables();
// Lifetime of a device's set: every batch handle whose circuit reads a table holds a reference from its creation to its destruction
// (device_tables_retain / _unref); device_tables_free releases the device memory of a set nobody holds (acvm_device_release_tables:
// 0 freed, 1 still in use, -1 device error); with tuning tables_keep = 0 the last handle's destruction frees it.
void device_tables_retain(int device);
void device_tables_unref(int device);
int device_tables_free(int device, size_t *bytes_freed);
// device memory the fixed tables of a circuit would still add to the current device (memory sizing of acvm_node_new)
size_t device_tables_missing_bytes(bool grumpkin, bool pedersen_level, bool window_table, bool ecdsa);

}  // namespace acvm
