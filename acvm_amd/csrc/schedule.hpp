// schedule.hpp -- the level schedule of one solve as DATA: which launch goes to which stream, which events it waits for and which it
// records. A pure function of the plan (and of the handle's instance stride, which only decides where a class's records are cut into
// launches by their scratch): no HIP call, no handle. Two consumers read the same list --
//   * batch_schedule.cpp enqueues it (one HIP call per step), so what runs on the device is exactly this list;
//   * schedule_check.cpp walks it on the host, derives every launch's reads and writes from the RECORD WORDS (not from the planner's
//     dependency tables) and proves that stream order + events order every conflicting pair (tests/test_schedule_hazards.py, no GPU).
// The reference executes opcodes strictly in order (acvm/src/pwg/mod.rs:236-303); a missing edge here would be a timing-dependent wrong
// witness, which no parity test on one box is sure to see.
#pragma once
#include "plan.hpp"
#include <cstdint>
#include <vector>

namespace acvm {

// records [first, first + count) of a class's level-major list (plan.cls_offset[k]); coop: CLS_HASH byte-message records (PLAN_HASH_COOP_FLAG)
// with lds_words message words per instance, or CLS_LIGHT straight-line Brillig records (a kernel of their own)
struct LaunchChunk { uint32_t first, count; bool coop = false; uint32_t lds_words = 0; };

// where the records of every class are cut into launches, and where each record's per-instance scratch lies inside its class's buffer
struct LaunchLayout {
    std::vector<std::vector<LaunchChunk>> cls_chunks[N_CLS];  // [class][level]
    std::vector<uint32_t> scratch_off[N_CLS];                 // per record of cls_offset[k]: (offset, words) in u32 words per instance
    uint64_t scratch_words[N_CLS] = {0, 0, 0, 0, 0, 0, 0, 0};    // u32 words per instance the class's buffer must hold (level launches and exact path)
    uint64_t cls_exact_words[N_CLS] = {0, 0, 0, 0, 0, 0, 0, 0};  // largest per-lane scratch of a single record of the class (exact path)
};
LaunchLayout layout_launches(const Plan &p, uint64_t Bp);

enum SchedStream : uint8_t { SS_MAIN = 0, SS_INV = 1, SS_LANE0 = 2, SS_LANE1 = 3, SS_LANE2 = 4, SS_LANE3 = 5, N_SCHED_STREAMS = 6 };
enum SchedKind : uint8_t { SK_LAUNCH = 0, SK_RECORD = 1, SK_WAIT = 2 };
enum SchedOp : uint8_t {
    SO_EVENT_RESET = 0,  // event words <- "solved"
    SO_GATES,            // gate wave programs [first, first + count) of gate_offset
    SO_GATES_LIGHT,      // the same plus light records [first2, first2 + count2) of cls_offset[CLS_LIGHT] in one launch
    SO_LIGHT, SO_LIGHT_SL, SO_HASH_COOP, SO_HASH, SO_GRUMPKIN, SO_BRILLIG, SO_PEDERSEN, SO_ECDSA, SO_DIGEST,  // records [first, first + count) of cls_offset[cls]
    SO_HOSTBB,           // the same, executed by host callbacks between two small kernels on the main stream (caller-supplied solver)
    SO_INVERSE,          // inversion jobs [first, first + count) of dyn_offset
    SO_TRUNCATE,         // event words <- min(event, truncated_at): a plan the level kernels do not cover entirely
    N_SCHED_OPS
};
const char *sched_op_name(uint32_t op);

struct SchedStep {
    uint8_t kind = SK_LAUNCH, stream = SS_MAIN, op = 0, cls = 0;
    uint32_t level = 0;                // 0-based level of the launch
    uint32_t first = 0, count = 0;     // launch: its records (see SchedOp)
    uint32_t first2 = 0, count2 = 0;   // SO_GATES_LIGHT: the light records
    uint32_t lds_words = 0;            // SO_HASH_COOP
    uint32_t event = 0;                // SK_RECORD / SK_WAIT: event id (sched_sync_event / sched_heavy_event)
};
// event ids: [0, 2 n_levels]: the handle's ev_sync (2L: "main levels < L are done", 2L + 1: the inversion batch of level L, 2 n_levels: the
// start of the solve); then ev_heavy[4 L + q]: the records of heavy lane q at level L have run
inline uint32_t sched_sync_event(uint32_t i) { return i; }
inline uint32_t sched_heavy_event(const Plan &p, uint32_t i) { return 2 * p.n_levels + 1 + i; }

struct LevelSchedule {
    std::vector<SchedStep> steps;  // in the order the host enqueues them
    bool one_stream = false;       // every launch on the main stream (tuning overlap / heavy_streams, or a circuit of byte hashes only)
};
LevelSchedule level_schedule(const Plan &p, const LaunchLayout &lay);

// ---- the hazard checker (schedule_check.cpp)
struct ScheduleReport {
    bool ok = true;
    uint64_t n_launches = 0, n_waits = 0, n_accesses = 0, n_records = 0;
    uint32_t n_findings = 0;
    std::string text;  // one line per finding (the first 32)
};
// drop_wait: index (among the SK_WAIT steps) of a wait to leave out -- the mutation of tests/test_schedule_hazards.py; 0xFFFFFFFF = none
ScheduleReport check_level_schedule(const Plan &p, const LaunchLayout &lay, const LevelSchedule &s, uint32_t drop_wait = 0xFFFFFFFFu);

}  // namespace acvm
