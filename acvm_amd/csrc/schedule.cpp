// schedule.cpp -- launch layout and level schedule of one solve as pure functions of the plan (schedule.hpp). Host only: no HIP call.
#include "schedule.hpp"
#include <algorithm>

namespace acvm {

const char *sched_op_name(uint32_t op) {
    static const char *names[N_SCHED_OPS] = {"event_reset", "gates", "gates+light", "light", "light_sl", "hash_coop", "hash", "grumpkin", "brillig",
                                             "pedersen", "ecdsa", "digest", "host_blackbox", "inverse", "truncate"};
    return op < N_SCHED_OPS ? names[op] : "?";
}

// Per level and class: launch chunks whose per-instance scratch fits the class's scratch buffer (1 GiB per class). The order of a level's
// records inside a class is the planner's (plan.cpp order_level_records): byte-message hashes come in two groups of their own (no scratch),
// straight-line Brillig records close the light class.
LaunchLayout layout_launches(const Plan &p, uint64_t Bp) {
    LaunchLayout lay;
    const uint64_t scratch_cap_words = std::max<uint64_t>(1, (1ull << 30) / (std::max<uint64_t>(Bp, 64) * 4));  // (an empty batch is allowed)
    const size_t n_levels = p.n_levels;
    for (int k = 0; k < (int)N_CLS; k++) {
        lay.cls_chunks[k].assign(n_levels, {});
        lay.scratch_off[k].assign(2 * p.cls_offset[k].size(), 0);
        uint64_t need = 0;
        for (size_t L = 0; L < n_levels; L++) {
            uint32_t lo = p.cls_level_start[k][L];
            const uint32_t hi = p.cls_level_start[k][L + 1];
            if (k == CLS_HASH) {  // byte-message hashes first: their own kernel, no scratch (the records of a level are independent)
                for (int group = 0; group < 2; group++) {
                    uint32_t n = 0, words = 0;
                    while (lo + n < hi && hash_launch_group(p.prog, p.cls_offset[k][lo + n]) == group) {
                        words = std::max(words, hash_record_lds_words(p.prog, p.cls_offset[k][lo + n]));
                        n++;
                    }
                    if (n) lay.cls_chunks[k][L].push_back({lo, n, true, words});
                    lo += n;
                }
            }
            // straight-line Brillig records close the list of their class (the light class on the main stream, or the Brillig lane: tuning sl_lane):
            // a kernel of their own, no scratch
            uint32_t n_sl = 0;
            if (k == CLS_LIGHT || k == CLS_BRILLIG)
                while (n_sl < hi - lo && p.prog[p.cls_offset[k][hi - 1 - n_sl]] == PK_BRILLIG_SL) n_sl++;
            const uint32_t hi_scratch = hi - n_sl;
            uint32_t first = lo;
            uint64_t used = 0;
            for (uint32_t r = lo; r < hi_scratch; r++) {
                const uint64_t w = p.cls_scratch[k][r];
                if (r > first && used + w > scratch_cap_words) {
                    lay.cls_chunks[k][L].push_back({first, r - first});
                    first = r;
                    used = 0;
                }
                lay.scratch_off[k][2 * (size_t)r] = (uint32_t)used;
                lay.scratch_off[k][2 * (size_t)r + 1] = (uint32_t)w;
                used += w;
                need = std::max(need, used);
            }
            if (hi_scratch > first) lay.cls_chunks[k][L].push_back({first, hi_scratch - first});
            if (n_sl) lay.cls_chunks[k][L].push_back({hi_scratch, n_sl, true});
        }
        // the exact kernels use slot 0 of the same buffer: it must hold the largest single record
        for (uint32_t oi = 0; oi < p.n_opcodes; oi++)
            if (p.prog_class[oi] == (uint32_t)k) {
                need = std::max<uint64_t>(need, p.prog_scratch[oi]);
                lay.cls_exact_words[k] = std::max<uint64_t>(lay.cls_exact_words[k], p.prog_scratch[oi]);
            }
        lay.scratch_words[k] = need;
    }
    return lay;
}

// The level schedule: the gate levels and the light records on the main stream, the inversion batches on a second one, the heavy record
// classes on lanes of their own (plan.hpp heavy_lane: hashes, Grumpkin, ECDSA | Pedersen | Brillig | digest leaves), each a stream in order.
//   * Level L + 1 of the main stream follows level L in stream order. It waits for an inversion batch only if one of its gates reads that
//     batch's rows (plan.level_needs_inverse: the planner put those gates after the batch, usually several levels after), and for a heavy
//     lane only up to the level whose outputs it reads (plan.level_needs_heavy[lane]): a level that reads a hash output does not wait for
//     the Pedersen launch beside it. Under slot reuse the same two tables carry the readers and asynchronous writers of a recycled row.
//   * The records of lane q at level L wait for the main stream only as far as they read it (plan.lane_needs_main: a hash of initial
//     witnesses and of other hashes never waits for the range checks launched beside it) and for the levels of the OTHER lanes whose
//     outputs they read (plan.lane_needs_lane).
//   * The inversion batch of level L waits for the main levels < L (its denominators; and the gates that read the rows of the inverse
//     table it overwrites ran there) and for the heavy levels that produced a denominator (plan.inv_needs_heavy).
//   * "Main levels < L are done" is an event recorded only where another stream is about to wait for it (an event between two gate launches
//     costs more than the launch gap itself).
// (One hipGraph of the whole schedule was measured in round 2 at -2 % on the 250 k-opcode circuit and ROCm 7.2's hipStreamEndCapture
// recursed without bound on the five-stream schedule of larger ones: removed.)
LevelSchedule level_schedule(const Plan &p, const LaunchLayout &lay) {
    LevelSchedule out;
    std::vector<SchedStep> &st = out.steps;
    const size_t n_levels = p.n_levels;
    const uint32_t NO_EVENT = 0xFFFFFFFFu;
    auto launch = [&](uint8_t stream, uint8_t op, uint32_t level, uint32_t first, uint32_t count, uint8_t cls = 0) -> SchedStep & {
        SchedStep s;
        s.kind = SK_LAUNCH; s.stream = stream; s.op = op; s.level = level; s.first = first; s.count = count; s.cls = cls;
        st.push_back(s);
        return st.back();
    };
    auto record = [&](uint8_t stream, uint32_t event) { SchedStep s; s.kind = SK_RECORD; s.stream = stream; s.event = event; st.push_back(s); };
    auto wait = [&](uint8_t stream, uint32_t event) { SchedStep s; s.kind = SK_WAIT; s.stream = stream; s.event = event; st.push_back(s); };
    const uint8_t s_main = SS_MAIN;
    const uint8_t s_inv = p.tune.overlap ? SS_INV : SS_MAIN;  // (overlap = 0, a measurement aid, serialises the two level kernels)
    launch(s_main, SO_EVENT_RESET, 0, 0, 0);
    auto heavy_cls = [](int k) { return k == CLS_HASH || k == CLS_GRUMPKIN || k == CLS_BRILLIG || k == CLS_PEDERSEN || k == CLS_ECDSA || k == CLS_DIGEST; };
    bool any_heavy = false;
    const bool any_main = !p.gate_offset.empty() || !p.cls_offset[CLS_LIGHT].empty();
    bool lane_any[N_HEAVY_LANES] = {false, false, false, false};  // a lane without records never joins the schedule
    for (int k = 0; k < (int)N_CLS; k++)
        if (heavy_cls(k) && !p.cls_offset[k].empty()) { any_heavy = true; lane_any[heavy_lane(k)] = true; }
    // A circuit of heavy records only kept everything on one stream until round 4 (round 2 had measured config 4 at 2.62 ... 3.03 ms from run to run
    // with the lanes side by side against a steady 2.84-2.89 on one stream, and config 3 0.18 instead of 0.15 ms). But a record kernel of the
    // integer-bound classes that follows a large launch -- the import of its tile -- on the SAME stream runs 16-44 % longer than on a stream of its
    // own (profiles/r04_import_effect.txt: config 4 import + solve 3.27 -> 2.15 ms, ECDSA 3.83 -> 3.32 ms per 2^16): such circuits take the lanes'
    // streams too (tuning heavy_only_streams); a circuit of byte-message hashes alone stays on the main stream.
    const bool integer_bound = !p.cls_offset[CLS_GRUMPKIN].empty() || !p.cls_offset[CLS_PEDERSEN].empty() || !p.cls_offset[CLS_ECDSA].empty() || !p.cls_offset[CLS_BRILLIG].empty();
    const bool one_stream = !p.tune.overlap || !p.tune.heavy_streams || (!any_main && !(p.tune.heavy_only_streams && integer_bound));
    out.one_stream = one_stream;
    const uint8_t lane_stream[N_HEAVY_LANES] = {one_stream ? s_main : (uint8_t)SS_LANE0, one_stream ? s_main : (uint8_t)SS_LANE1, one_stream ? s_main : (uint8_t)SS_LANE2,
                                                one_stream ? s_main : (uint8_t)SS_LANE3};
    const bool any_dyn = !p.dyn_offset.empty();
    if (any_dyn || any_heavy) {
        record(s_main, sched_sync_event(2 * (uint32_t)n_levels));
        if (any_dyn && s_inv != s_main) wait(s_inv, sched_sync_event(2 * (uint32_t)n_levels));
        if (any_heavy && !one_stream)
            for (int q = 0; q < N_HEAVY_LANES; q++)
                if (lane_any[q]) wait(lane_stream[q], sched_sync_event(2 * (uint32_t)n_levels));
    }
    uint32_t last_reg = NO_EVENT, last_dyn = NO_EVENT, last_lane[N_HEAVY_LANES] = {NO_EVENT, NO_EVENT, NO_EVENT, NO_EVENT};
    bool main_dirty = false;  // the main stream has launches behind last_reg
    std::vector<std::pair<uint32_t, uint32_t>> main_marks;  // (L, event): "main levels < L are done", in order
    uint32_t lane_main_waited[N_HEAVY_LANES] = {0, 0, 0, 0};
    uint32_t waited_inverse_level = 0, waited_heavy[N_HEAVY_LANES] = {0, 0, 0, 0}, lane_waited[N_HEAVY_LANES][N_HEAVY_LANES] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (size_t L = 0; L < n_levels; L++) {
        const uint32_t n = p.level_start[L + 1] - p.level_start[L];
        const uint32_t nd = p.dyn_level_start[L + 1] - p.dyn_level_start[L];
        bool s_work = n != 0, h_work = false;
        bool lane_used[N_HEAVY_LANES] = {false, false, false, false};
        for (int k = 0; k < (int)N_CLS; k++) {
            (heavy_cls(k) ? h_work : s_work) |= !lay.cls_chunks[k][L].empty();
            if (heavy_cls(k) && !lay.cls_chunks[k][L].empty()) lane_used[heavy_lane(k)] = true;
        }
        if ((nd || h_work) && main_dirty) {
            record(s_main, sched_sync_event(2 * (uint32_t)L));
            last_reg = sched_sync_event(2 * (uint32_t)L);
            main_marks.push_back({(uint32_t)L, last_reg});
            main_dirty = false;
        }
        const uint32_t prev_reg = last_reg;
        const uint32_t need = p.level_needs_inverse[L + 1];  // 1-based inversion level, 0 = none
        if (s_work && need > waited_inverse_level) {
            if (s_inv != s_main) wait(s_main, sched_sync_event(2 * (need - 1) + 1));
            waited_inverse_level = need;
        }
        for (int q = 0; q < N_HEAVY_LANES; q++) {
            const uint32_t need_h = p.level_needs_heavy[q][L + 1];  // 1-based level of the lane's records, 0 = none
            if (s_work && need_h > waited_heavy[q]) {
                if (!one_stream) wait(s_main, sched_heavy_event(p, 4 * (need_h - 1) + q));
                waited_heavy[q] = need_h;
            }
        }
        // the level's light records (not the straight-line Brillig ones: a kernel of their own) ride in the gate launch when there is one
        const LaunchChunk *fused_light = nullptr;
        if (n && p.tune.light_fuse)
            for (const LaunchChunk &ch : lay.cls_chunks[CLS_LIGHT][L])
                if (!ch.coop && (uint64_t)n + ch.count <= 65535u) { fused_light = &ch; break; }
        if (n) {
            SchedStep &g = launch(s_main, fused_light ? SO_GATES_LIGHT : SO_GATES, (uint32_t)L, p.level_start[L], n);
            if (fused_light) { g.first2 = fused_light->first; g.count2 = fused_light->count; }
        }
        // what the lanes wait for: levels < L of the main stream as far as they read them, and the other lanes likewise
        if (!one_stream)
            for (int q = 0; q < N_HEAVY_LANES; q++) {
                if (!lane_used[q]) continue;
                if (const uint32_t need_m = p.lane_needs_main[q][L + 1]; need_m > lane_main_waited[q]) {
                    // the earliest mark behind main level need_m (1-based): "levels < mark" with mark >= need_m
                    auto it = std::lower_bound(main_marks.begin(), main_marks.end(), need_m, [](const std::pair<uint32_t, uint32_t> &mk, uint32_t v) { return mk.first < v; });
                    if (it != main_marks.end()) {
                        wait(lane_stream[q], it->second);
                        lane_main_waited[q] = it->first;
                    }
                }
                for (int q2 = 0; q2 < N_HEAVY_LANES; q2++) {
                    const uint32_t need_l = p.lane_needs_lane[q][q2][L + 1];
                    if (q2 != q && lane_stream[q2] != lane_stream[q] && need_l > lane_waited[q][q2]) {
                        wait(lane_stream[q], sched_heavy_event(p, 4 * (need_l - 1) + q2));
                        lane_waited[q][q2] = need_l;
                    }
                }
            }
        for (int k = 0; k < (int)N_CLS; k++)
            for (const LaunchChunk &ch : lay.cls_chunks[k][L]) {
                if (&ch == fused_light) continue;  // went with the gates
                const uint8_t sk = heavy_cls(k) ? lane_stream[heavy_lane(k)] : s_main;
                uint8_t op = SO_LIGHT;
                switch (k) {
                case CLS_LIGHT: op = ch.coop ? SO_LIGHT_SL : SO_LIGHT; break;
                case CLS_HASH: op = ch.coop ? SO_HASH_COOP : SO_HASH; break;
                case CLS_GRUMPKIN: op = SO_GRUMPKIN; break;
                case CLS_BRILLIG: op = ch.coop ? SO_LIGHT_SL : SO_BRILLIG; break;  // (coop: straight-line records on the Brillig lane)
                case CLS_PEDERSEN: op = SO_PEDERSEN; break;
                case CLS_ECDSA: op = SO_ECDSA; break;
                case CLS_DIGEST: op = SO_DIGEST; break;
                case CLS_HOSTBB:  // host callbacks: everything launched so far on any stream must have finished
                    op = SO_HOSTBB;
                    if (last_dyn != NO_EVENT && s_inv != s_main) wait(s_main, last_dyn);
                    for (int q = 0; q < N_HEAVY_LANES; q++)
                        if (last_lane[q] != NO_EVENT) wait(s_main, last_lane[q]);
                    break;
                }
                launch(sk, op, (uint32_t)L, ch.first, ch.count, (uint8_t)k).lds_words = ch.lds_words;
            }
        if (!one_stream)
            for (int q = 0; q < N_HEAVY_LANES; q++)
                if (lane_used[q]) {
                    record(lane_stream[q], sched_heavy_event(p, 4 * (uint32_t)L + q));
                    last_lane[q] = sched_heavy_event(p, 4 * (uint32_t)L + q);
                }
        // (heavy records that share the main stream are launches of the main stream: until round 6 they left main_dirty alone, and with tuning
        // heavy_streams = 0 an inversion batch whose denominator a Pedersen record had written on the main stream waited for nothing --
        // found by schedule_check.cpp on tests/circuit_corpus.py "inverse_behind_pedersen", never by a parity test)
        main_dirty |= s_work || (one_stream && h_work);
        if (nd) {
            if (prev_reg != NO_EVENT && s_inv != s_main) wait(s_inv, prev_reg);
            if (!one_stream)
                for (int q = 0; q < N_HEAVY_LANES; q++)
                    if (p.inv_needs_heavy[q][L + 1]) wait(s_inv, sched_heavy_event(p, 4 * (p.inv_needs_heavy[q][L + 1] - 1) + q));
            launch(s_inv, SO_INVERSE, (uint32_t)L, p.dyn_level_start[L], nd);
            if (s_inv != s_main) {
                record(s_inv, sched_sync_event(2 * (uint32_t)L + 1));
                last_dyn = sched_sync_event(2 * (uint32_t)L + 1);
            }
        }
    }
    for (int q = 0; q < N_HEAVY_LANES; q++)
        if (last_lane[q] != NO_EVENT) wait(s_main, last_lane[q]);
    if (last_dyn != NO_EVENT) wait(s_main, last_dyn);
    if (p.truncated_at != 0xFFFFFFFFu) launch(s_main, SO_TRUNCATE, (uint32_t)n_levels, 0, 0);
    return out;
}

}  // namespace acvm
