// fuzz_driver.cpp -- the host side that parses UNTRUSTED bytes (circuit.cpp: Circuit::read and the WitnessMap reader; plan.cpp: the planner
// that indexes by what the bytes say), built with -fsanitize=address,undefined for the CPU (Makefile target `asan`). The reference's
// reader returns Err on malformed input, never UB (acir/src/circuit/mod.rs:154-161, native_types/witness_map.rs:108-146); this driver
// asserts the same of ours: every input must end in "parsed" or "refused", never in a sanitizer report.
//
//   fuzz_driver FILE     FILE = a sequence of [u32 length][bytes] blobs; each blob is fed to the circuit reader (then planned against the
//                        first private parameters, the plain and the slot-reuse plan) and to the WitnessMap reader.
// Prints one summary line: blobs, circuits parsed, plans built, plans refused, witness maps parsed.
#include "../../acvm_amd/csrc/plan.hpp"
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
    unsigned long n_blobs = 0, n_parsed = 0, n_planned = 0, n_refused = 0, n_maps = 0, n_thrown = 0;
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
    printf("blobs %lu parsed %lu planned %lu refused %lu witness_maps %lu thrown %lu\n", n_blobs, n_parsed, n_planned, n_refused, n_maps, n_thrown);
    return 0;
}
