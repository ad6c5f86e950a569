// fuzz_driver.cpp -- the host side that parses UNTRUSTED bytes (circuit.cpp: Circuit::read and the WitnessMap reader; plan.cpp: the planner
// that indexes by what the bytes say), built with -fsanitize=address,undefined for the CPU (Makefile target `asan`). The reference's
// reader returns Err on malformed input, never UB (acir/src/circuit/mod.rs:154-161, native_types/witness_map.rs:108-146); this driver
// asserts the same of ours: every input must end in "parsed" or "refused", never in a sanitizer report.
//
//   fuzz_driver FILE     FILE = a sequence of [u32 length][bytes] blobs; each blob is fed to the circuit reader (then planned against the
//                        first private parameters, the plain and the slot-reuse plan) and to the WitnessMap reader.
// Every plan that builds is laid out, scheduled and hazard-checked too (schedule.cpp, schedule_check.cpp) for a small and a large tile.
// Prints one summary line: blobs, circuits parsed, plans built, plans refused, witness maps parsed, exceptions, schedules checked, schedules with a finding
// (exit code 3 if any).
#include "../../acvm_amd/csrc/schedule.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <vector>

using namespace acvm;

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: fuzz_driver FILE\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    unsigned long n_blobs = 0, n_parsed = 0, n_planned = 0, n_refused = 0, n_maps = 0, n_thrown = 0, n_checked = 0, n_hazards = 0, n_skipped = 0;
    for (;;) {
        uint32_t len = 0;
        if (fread(&len, 4, 1, f) != 1) break;
        if (len > (64u << 20)) { fprintf(stderr, "blob too long\n"); return 2; }
        std::vector<uint8_t> buf(len);
        if (len && fread(buf.data(), 1, len, f) != len) break;
        n_blobs++;
        try {
            std::string err;
            auto c = circuit_from_bytes(buf.data(), buf.size(), err);
            if (c) {
                n_parsed++;
                std::vector<uint32_t> ids = c->private_parameters;
                ids.insert(ids.end(), c->public_parameters.begin(), c->public_parameters.end());
                std::sort(ids.begin(), ids.end());
                ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
                if (ids.size() > 4096) ids.resize(4096);
                for (int mode = 0; mode < 2; mode++) {
                    PlanOpts opts;
                    opts.fold_digest = opts.reuse_slots = mode == 1;
                    opts.keep = c->return_values;
                    Plan p = build_plan(*c, ids.data(), (uint32_t)ids.size(), opts);
                    (p.unsupported.empty() ? n_planned : n_refused)++;
                    if (!p.unsupported.empty()) continue;
                    // what a batch handle of this plan would enqueue, proved ordered (schedule_check.cpp): a mutant that the reader accepts and the planner
                    // plans is a circuit like any other -- its schedule must be as free of hazards as the corpus's (and the walk as free of UB)
                    // (the checker keeps ~72 bytes per row: a mutant that DECLARES 2^26 witnesses -- the reader's bound is 2^27 -- is walked by the entry point on a
                    // real host, not under the sanitizer's 4 GB allocation cap)
                    if ((uint64_t)p.n_witnesses + p.n_inverse_slots + p.mem_cells > (1u << 22)) { n_skipped++; continue; }
                    for (uint64_t Bp : {(uint64_t)64, (uint64_t)1 << 17}) {
                        const LaunchLayout lay = layout_launches(p, Bp);
                        const LevelSchedule sched = level_schedule(p, lay);
                        const ScheduleReport rep = check_level_schedule(p, lay, sched, 0xFFFFFFFFu);
                        n_checked++;
                        if (!rep.ok) {
                            n_hazards++;
                            fprintf(stderr, "blob %lu mode %d Bp %llu: %s\n", n_blobs, mode, (unsigned long long)Bp, rep.text.c_str());
                        }
                    }
                }
            }
            std::vector<uint32_t> wid;
            std::vector<uint8_t> wval;
            if (witness_map_from_bytes(buf.data(), buf.size(), wid, wval, err)) n_maps++;
        } catch (const std::bad_alloc &) {  // what the ABI turns into ACVM_E_NOMEM
            n_thrown++;
        } catch (const std::exception &) {  // ... into ACVM_E_INVALID
            n_thrown++;
        }
    }
    fclose(f);
    printf("blobs %lu parsed %lu planned %lu refused %lu witness_maps %lu thrown %lu schedules %lu hazards %lu too_large_to_check %lu\n", n_blobs, n_parsed, n_planned, n_refused, n_maps,
           n_thrown, n_checked, n_hazards, n_skipped);
    return n_hazards ? 3 : 0;
}
