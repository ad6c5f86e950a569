// tuning.hpp -- every planner / scheduler mode of the library in ONE place (process-wide defaults; a plan keeps the snapshot it was
// built with). The defaults are the measured optimum of NOTEBOOK.md section 9; the other values exist so that the A/B measurements of
// that section stay reproducible and so that the parity tests can sweep the planner's modes (tests/test_gpu_planner_modes.py). Set
// through the ABI (acvm_tuning_set, include/acvm_amd.h) or, for command-line tools, once at first use from the environment variable
// ACVM_TUNING="key=value,key=value". No mode changes any result: every one of them is compared bit for bit against the oracle.
#pragma once
#include <stdint.h>

namespace acvm {

struct Tuning {
    // ---- planner (plan.cpp)
    int64_t scale = 1;             // projective witnesses (a gate's most expensive coefficient becomes 1)
    int64_t relax = 1;             // relaxed rows: a scaled witness is stored as any representative below 2^256 (gate_eval.hpp); needs scale
    int64_t pairs = 1;             // wave programs: a gate runs behind its producer in the same wave
    int64_t chains = 1;            // ... and behind a tail of that wave
    int64_t max_tails = 5;         // records behind the host of a wave program
    int64_t inv_epoch = 4;         // levels per batch of denominator inversions
    int64_t byte_plane = 1;        // initial witnesses that byte-message hashes read get a second, 4-byte copy (low limb + is-byte flag) written by the import: the hash kernel reads 4 bytes per input instead of 32
    int64_t inv_chunk = 128;       // denominators per wave of an inversion batch at most (they share ONE field inversion; 64 -> 128: +0.9 % on the headline, NOTEBOOK.md section 9)
    int64_t inv_latency = 1;       // levels of slack between an inversion batch and the first gate that reads it
    int64_t heavy_epoch = 1;       // heavy records launched every K-th level only
    int64_t heavy_latency = 0;     // levels the main stream waits before it reads a heavy output
    // Pedersen records are launched every 8th level only and the main stream reads their outputs 4 levels behind the launch at the earliest: a
    // launch of two or three records is a handful of long waves, eight levels' worth fill the lane. Round 6, 10^6-opcode config-5 tile of 8 192
    // (profiles/r06_config5_sweep.txt, four boxes): epoch / latency 2 / 1 (the default until then) 138.5-141.7 ms per 4 096 instances, 4 / 2 139.8, 8 / 3 135.5,
    // 8 / 4 132.4-136.4, 8 / 6 138.7, 12 / 6 139.4, 16 / 8 137.4, 32 / 16 137.9; the north-star shape (8 Pedersen, consumers right behind) 5.228 M witnesses/s
    // at 2 / 1, 5.225 M at 8 / 4, 5.275 M at 16 / 8. (A slack-aware variant -- only records whose readers are far away move to the epoch boundary -- was
    // measured SLOWER than leaving everything alone, 140.8-142.8 against 138.5: removed.)
    int64_t pedersen_latency = 4;  // the same for Pedersen outputs alone
    int64_t pedersen_epoch = 8;    // Pedersen records launched every K-th level only
    int64_t digest_epoch = 8;      // levels per batch of folded digest leaves
    int64_t range_fuse = 1;        // byte RANGE checks run inside the hash that reads the byte
    int64_t range_merge = 1;       // RANGE opcodes of a level, eight to a record
    int64_t hash_chain = 1;        // a byte-message hash of another one's digest runs behind it in the same workgroup
    int64_t brillig_inline = 1;    // straight-line Brillig programs compiled into light records of the level schedule
    int64_t sl_lane = 0;           // 1: ... which run on the Brillig lane beside the gate levels instead of on the main stream between them (round 6, 10^6-opcode tile of 8 192,
                                   // A-B-A-B on one box: 130.6 / 135.3 ms per 4 096 instances on the main stream, 134.2 / 135.0 on the lane: no difference, the default stays)
    int64_t pedersen_waves = 0;    // waves per 64 instances of the level Pedersen kernel: 0 = four, or one when the launch fills the chip anyhow; 1 / 4 force
    int64_t pedersen_bundle = 1;   // up to eight Pedersen records of a launch per wave, ONE inversion per chain step for all of them: 0 never, 1 in launches that fill the chip anyhow, 2 always (tests)
    int64_t pedersen_bundle_waves = 2048;  // ... as many records per wave as leave the launch this many waves (two per SIMD; measured, NOTEBOOK.md section 9: at 1 024 the north-star shape in tiles of 2^16 loses 2 %, at 512 a config-5 tile of 4 096 gets slower)
    int64_t pedersen_prio = 1;     // s_setprio 3 in the level Pedersen kernel: its few long waves win the issue arbitration against the gate kernel's many
    int64_t light_fuse = 1;        // the light records of a level ride in its gate launch
    int64_t plan_validate = 0;     // 1: every pass of the planner checks what it promises (plan.cpp check_*; a violation is an error of the call); 100 + k: break invariant k first (tests)
    int64_t brillig_mem_cells = 0; // lower bound of the per-lane Brillig memory of the level kernels (0: the planner's estimate)
    // ---- driver (batch.cpp)
    int64_t overlap = 1;           // inversion batches and heavy lanes on streams of their own
    int64_t heavy_streams = 1;     // heavy lanes beside the main stream (0: everything on the main stream)
    int64_t heavy_only_streams = 1; // a circuit of heavy records only: its lanes on their own streams too (0: one stream). A record kernel that follows a
                                   // large launch -- the import of its tile -- on the SAME stream runs 16-44 % longer (profiles/r04_import_effect.txt):
                                   // config 4 import + solve 3.27 -> 2.15 ms, ECDSA 3.83 -> 3.32 ms per 2^16
    int64_t fc_relevel = -1;       // answered foreign calls re-enter the level schedule: 1 always, 0 never, -1 when >= 1/16 of the batch was answered
    int64_t exact_async = 1;       // the exact path of tile k runs beside the level schedule of tile k + 1 (acvm_node_*)
    // Brillig VM limits of the device (the reference has none: brillig_vm/src/{memory.rs:27-39, lib.rs:154-307}). The level kernels run
    // with the first value; an instance that reaches it continues on the exact path, which retries with the limit raised step by
    // step up to the second value; beyond that THAT instance ends with ACVM_ERR_DEVICE_LIMIT (include/acvm_amd.h: not a reference
    // outcome, the other instances keep their results).
    int64_t brillig_steps_log2 = 22, brillig_steps_max_log2 = 26;
    int64_t brillig_call_depth = 64, brillig_call_depth_max = 1 << 16;
    int64_t brillig_mem_max_log2 = 22;  // cells of one lane's memory on the exact path at most (32 B each)
    // ---- tables (grumpkin_host.cpp)
    int64_t pedersen_window_bits = 24; // nonzero: the level Pedersen kernel reads 24-bit windows of the scalar (GRUMPKIN_PEDW_BITS: 23.6 GB of tables, 11 additions per hash_single) instead of slice pairs (0: 503 MB, 15)
    int64_t win16 = 1;             // 16-bit window tables of the four fixed bases
    int64_t tables_keep = 1;       // 1: a device's lookup tables stay until acvm_device_release_tables; 0: the last handle of the device frees them
};

Tuning &tuning();                                   // the process-wide defaults
bool tuning_set(const char *key, int64_t value);    // false: unknown key
bool tuning_get(const char *key, int64_t *value);
const char *tuning_key(unsigned index);             // enumeration for documentation / tests, nullptr past the end

}  // namespace acvm
