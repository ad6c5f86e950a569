// batch_schedule.cpp -- ACVM::solve for the batch (acvm/src/pwg/mod.rs:236-241): the level schedule of one solve enqueued on the handle's
// streams (gate levels and light records on the main stream, inversion batches on a second one, the heavy record classes on lanes of their
// own), the count of instances that left the generic path, and the hand-over to the exact path (batch_exact.cpp).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include "batch_internal.hpp"

struct LaunchTimers {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> reg_pairs, dyn_pairs, cls_pairs[N_CLS];
    size_t ev_used = 0;
};

// The level schedule of one solve (schedule.cpp level_schedule: a pure function of the plan, built once per handle) enqueued on the handle's
// streams: one HIP call per step, nothing decided here. What the hazard checker proves about that list (schedule_check.cpp) is therefore
// true of what the device is given.
static int enqueue_level_schedule(acvm_batch *b, LaunchTimers *tm) {
    const Plan &p = b->plan();
    auto next_event = [&]() -> hipEvent_t {
        if (tm->ev_used == b->ev_pool.size()) {
            hipEvent_t e;
            hipEventCreate(&e);
            b->ev_pool.push_back(e);
        }
        return b->ev_pool[tm->ev_used++];
    };
    const bool prof = tm != nullptr;
    hipStream_t streams[N_SCHED_STREAMS] = {b->stream, b->stream_dyn, b->stream_heavy, b->stream_heavy2, b->stream_heavy3, b->stream_digest};
    const uint32_t n_sync = 2 * p.n_levels + 1;
    auto event_of = [&](uint32_t id) { return id < n_sync ? b->ev_sync[id] : b->ev_heavy[id - n_sync]; };
    for (const SchedStep &st : b->schedule.steps) {
        hipStream_t s = streams[st.stream];
        if (st.kind == SK_RECORD) { HIPCHK(hipEventRecord(event_of(st.event), s)); continue; }
        if (st.kind == SK_WAIT) { HIPCHK(hipStreamWaitEvent(s, event_of(st.event), 0)); continue; }
        const int k = st.cls;
        const uint32_t *off = b->d_cls_offset[k] + st.first, *soff = b->d_cls_scratch_off[k] + 2 * (size_t)st.first;
        const bool timed = prof && st.op != SO_EVENT_RESET && st.op != SO_TRUNCATE;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) { e0 = next_event(); hipEventRecord(e0, s); }
        switch (st.op) {
        case SO_EVENT_RESET:  // (the import of this tile usually left the event words ready: kernels.hip import_witness_kernel)
            if (!b->events_fresh) launch_event_reset(s, b->d_event, b->B);
            b->events_fresh = false;
            break;
        case SO_GATES:
            launch_arith_level(s, b->d_W, b->Bp, b->B, b->d_gate_stream, b->d_gate_offset + st.first, st.count, b->d_consts, b->d_event, b->d_inv);
            break;
        case SO_GATES_LIGHT:
            launch_arith_light_level(s, b->d_W, b->Bp, b->B, b->d_gate_stream, b->d_gate_offset + st.first, st.count, b->d_inv, b->dp,
                                     b->d_cls_offset[CLS_LIGHT] + st.first2, st.count2, b->d_event);
            break;
        case SO_LIGHT: launch_light_level(s, b->d_W, b->Bp, b->B, b->dp, off, st.count, b->d_event); break;
        case SO_LIGHT_SL: launch_light_sl_level(s, b->d_W, b->Bp, b->B, b->dp, off, st.count, b->d_event); break;
        case SO_HASH_COOP: launch_hash_coop_level(s, b->d_W, b->Bp, b->B, b->dp, off, st.count, b->d_event, st.lds_words); break;
        case SO_HASH: launch_hash_level(s, b->d_W, b->Bp, b->B, b->dp, off, soff, st.count, b->d_event, b->d_cls_scratch[k]); break;
        case SO_GRUMPKIN: launch_grumpkin_level(s, b->d_W, b->Bp, b->B, b->dp, off, soff, st.count, b->d_event, b->d_cls_scratch[k]); break;
        case SO_BRILLIG: launch_brillig_level(s, b->d_W, b->Bp, b->B, b->dp, off, soff, st.count, b->d_event, b->d_cls_scratch[k]); break;
        case SO_PEDERSEN: launch_pedersen_level(s, b->d_W, b->Bp, b->B, b->dp, off, st.count, b->d_event, soff, b->d_cls_scratch[k]); break;
        case SO_ECDSA: launch_ecdsa_level(s, b->d_W, b->Bp, b->B, b->dp, off, st.count, b->d_event); break;
        case SO_DIGEST: launch_digest_fold_level(s, b->d_W, b->Bp, b->B, b->dp, off, st.count, b->fp, b->d_leaves); break;
        case SO_HOSTBB:  // host callbacks (the schedule put the waits for every other stream in front)
            for (uint32_t r = 0; r < st.count; r++)
                if (int rc = run_host_blackbox(b, p.prog[p.cls_offset[k][st.first + r] + 1], false, 0)) return rc;
            break;
        case SO_INVERSE:
            launch_inverse_batch(s, b->d_W, b->d_inv, b->Bp, b->B, b->d_gate_stream, b->d_dyn_offset + st.first, st.count, b->d_event, (uint32_t)p.tune.inv_chunk);
            break;
        case SO_TRUNCATE: launch_event_truncate(s, b->d_event, b->B, p.truncated_at); break;
        }
        if (timed) {
            e1 = next_event();
            hipEventRecord(e1, s);
            (st.op == SO_GATES || st.op == SO_GATES_LIGHT ? tm->reg_pairs : st.op == SO_INVERSE ? tm->dyn_pairs : tm->cls_pairs[k]).push_back({e0, e1});
        }
        if (st.op == SO_GATES || st.op == SO_GATES_LIGHT) b->n_launches += (st.count + 65534) / 65535;
        else if (st.op != SO_EVENT_RESET && st.op != SO_TRUNCATE) b->n_launches++;
    }
    return 0;
}

// the cross-stream events of the schedule exist before anything is enqueued (nothing is created under stream capture)
static int ensure_level_events(acvm_batch *b) {
    const size_t n_levels = b->plan().n_levels;
    while (b->ev_sync.size() < 2 * n_levels + 1 || b->ev_heavy.size() < 4 * n_levels) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        (b->ev_sync.size() < 2 * n_levels + 1 ? b->ev_sync : b->ev_heavy).push_back(e);
    }
    return 0;
}
// The side table of the exact path (slot reuse, and the asynchronous jobs of the node driver): all witnesses x `n_lanes` flagged instances,
// padded to 64 lanes, the memory blocks beside it and -- for a job that runs beside the next tile's level kernels, which use the class
// buffers meanwhile -- per-class scratch of its own. Grow-only. 0 = ready, negative = an error (ACVM_E_DEVICE: no room).

int batch_solve_impl(acvm_batch *b, const void *next_inputs) {
    if (!b->inputs_set && !b->plan().initial_ids.empty()) return set_err(ACVM_E_STATE, "initial witness not set");
    HIPCHK(hipSetDevice(b->device));
    const Plan &p = b->plan();
    b->next_imported = false;
    b->next_inputs = nullptr;
    if (b->solved) {  // only resolved foreign calls can change anything
        if (b->stepping) return solve_stepping(b, false);
        // A few resumed instances continue on the exact in-order kernels from their Brillig opcode on. When a sizeable part of the
        // batch was answered (the usual case: every instance reaches the same oracle call), the whole LEVEL schedule runs again
        // instead: the answers are in the result store, the Brillig level kernel finds them, and the opcodes behind the call
        // run level-parallel for everybody (instances still waiting, or waiting at the next call, are flagged again there).
        uint32_t n_resolved = 0;
        for (uint32_t t = 0; t < b->slow_ids.size() && t < b->fc_lane.size(); t++)
            n_resolved += b->slow_res[t].status == ACVM_STATUS_REQUIRES_FOREIGN_CALL && b->fc_lane[t].resolved_new;
        const int64_t mode = p.tune.fc_relevel;
        const bool relevel = n_resolved && (mode >= 0 ? mode != 0 : (uint64_t)n_resolved * 16 >= b->B);
        if (!relevel) return solve_resume(b);
        b->solved = false;
    }
    hipStream_t s = b->stream;
    if (int rc = upload_fc_tables(b, 0)) return rc;  // the level kernels read the resolved results too
    b->n_launches = 0;
    b->n_brillig_retries = 0;
    b->host_bb_msg.clear();
    b->arith_kernel_ms = 0;
    b->dyn_kernel_ms = 0;
    b->slow_path_ms = 0;
    for (int k = 0; k < (int)N_CLS; k++) b->cls_kernel_ms[k] = 0;
    LaunchTimers tm;
    auto next_event = [&]() -> hipEvent_t {
        if (tm.ev_used == b->ev_pool.size()) {
            hipEvent_t e;
            hipEventCreate(&e);
            b->ev_pool.push_back(e);
        }
        return b->ev_pool[tm.ev_used++];
    };
    if (!b->force_slow)
        if (int rc = ensure_level_events(b)) return rc;
    HIPCHK(hipEventRecord(b->ev_start, s));
    *b->h_flag_count = 0;  // (the kernels of this solve count the instances they flag: ops_common.hpp flag_instance; every earlier solve has been waited for)
    if (b->force_slow) {
        launch_fill_u32(s, b->d_event, 0u, b->B);
        b->events_fresh = false;
    } else {
        if (int rc = enqueue_level_schedule(b, b->profiling ? &tm : nullptr)) return rc;
    }
    HIPCHK(hipGetLastError());
    // the exact job of the PREVIOUS solve (asynchronous mode) is collected here, while the device works on this solve's levels
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    // instances that left the generic path (or hit a failing opcode): exact in-order re-solve from their event on. Usually there is
    // none: only their count comes back -- kept by the kernels that flag, in a host-mapped word the host reads after the synchronisation it
    // needs anyway (no counting kernel and no copy behind the last kernel of a solve)
    uint32_t n_flagged = b->B;
    if (!b->force_slow && b->B) {
        if (next_inputs) {
            if (!b->ev_counted) HIPCHK(hipEventCreate(&b->ev_counted));
            HIPCHK(hipEventRecord(b->ev_counted, s));
            // gate: the device's count of flagged instances, in front of the event words; the import leaves the event words ready for the next solve
            launch_import(s, b->d_W, b->Bp, b->B, (const uint8_t *)next_inputs, b->reuse() ? b->d_init_rows : b->d_init_ids, (uint32_t)p.initial_ids.size(),
                          b->d_event - 4, b->d_byte_plane_of_input, b->d_byte_plane, b->d_event);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventSynchronize(b->ev_counted));
        } else HIPCHK(hipStreamSynchronize(s));
        n_flagged = p.truncated_at != 0xFFFFFFFFu ? b->B : *(volatile uint32_t *)b->h_flag_count;
    }
    const bool imported_next = next_inputs && !b->force_slow && b->B && n_flagged == 0;
    if (n_flagged || !b->events_clean) {
        if (n_flagged) {
            HIPCHK(hipMemcpyAsync(b->h_event.data(), b->d_event, (size_t)b->B * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        } else std::fill(b->h_event.begin(), b->h_event.end(), 0xFFFFFFFFu);
        b->slow_ids.clear();
        std::fill(b->slow_index.begin(), b->slow_index.end(), -1);
        for (uint32_t j = 0; j < b->B; j++)
            if (b->h_event[j] != 0xFFFFFFFFu) {
                b->slow_index[j] = (int32_t)b->slow_ids.size();
                b->slow_ids.push_back(j);
            }
        b->events_clean = n_flagged == 0;
    }
    uint32_t n_slow = (uint32_t)b->slow_ids.size();
    hipEvent_t slow0 = nullptr, slow1 = nullptr;
    // asynchronous exact path (node.cpp): a bounded number of flagged instances is re-solved in the side table on stream_x while the
    // caller goes on to the next tile; a batch full of them (a failing circuit, a truncated plan) keeps the synchronous path
    bool go_async = b->async_exact && n_slow && !b->force_slow && (uint64_t)n_slow * 8 <= std::max<uint64_t>(b->B, 512);
    b->side_job = go_async;
    if (n_slow) {
        if (int rc = ensure_slow_capacity(b, n_slow)) return rc;
        HIPCHK(hipMemcpyAsync(b->d_slow_ids, b->slow_ids.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        b->slow_start.resize(n_slow);
        uint32_t min_start = 0xFFFFFFFFu;
        for (uint32_t t = 0; t < n_slow; t++) {
            // slot reuse: the level table no longer holds what ran before the event: the lane starts over from its initial witnesses
            b->slow_start[t] = b->reuse() ? 0u : b->h_event[b->slow_ids[t]];
            min_start = std::min(min_start, b->slow_start[t]);
        }
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        slow0 = next_event();
        slow1 = next_event();
        hipEventRecord(slow0, s);
        if (b->side()) {
            const int grown = ensure_side_table(b, n_slow, go_async);
            if (grown < 0) {
                // no room for the side table of this many lanes. A handle that does not recycle rows still has every column in the level table:
                // the job runs there, in place, before the caller's next import (the synchronous path); slot reuse has no such fallback
                if (b->reuse()) return grown;
                (void)hipGetLastError();
                go_async = false;
                b->side_job = false;
            }
        }
        if (b->side()) {
            // (on the batch's stream: the level table is read before the next tile's import overwrites it)
            if (b->reuse()) launch_gather_initial(s, b->d_Wx, b->x_cap, b->d_W, b->Bp, b->d_init_ids, b->d_init_rows, (uint32_t)p.initial_ids.size(), b->d_slow_ids, n_slow);
            else {  // every flagged instance's column as far as its event (what its assigned set will hold), plain values: the job resumes there like the in-place path
                launch_gather_columns(s, b->d_Wx, b->x_cap, b->d_W, b->Bp, p.n_witnesses, b->d_slow_ids, n_slow, b->d_unscale_index, b->d_unscale_consts, b->d_producer, b->d_slow_start);
                launch_gather_columns(s, b->d_Memx, b->x_cap, b->d_Mem, b->Bp, p.mem_cells, b->d_slow_ids, n_slow, nullptr, nullptr, nullptr, nullptr);
            }
        } else
        launch_unscale_slow(s, b->d_W, b->Bp, b->d_slow_ids, n_slow, b->unscale, b->d_producer, b->d_slow_start);  // the exact kernels work on plain values
        if (go_async) {  // everything below runs on the side stream, behind the gather
            HIPCHK(hipEventRecord(b->ev_x_ready, s));
            HIPCHK(hipStreamWaitEvent(b->stream_x, b->ev_x_ready, 0));
            b->pending = true;
        }
        hipStream_t xs = b->xstream();
        launch_init_assigned(xs, b->d_assigned, n_slow, b->n_words, p.n_witnesses, b->d_producer, b->d_slow_start);
        b->fc_lane.assign(n_slow, acvm_batch::FcLaneState());
        if (int rc = upload_fc_tables(b, n_slow)) return rc;
        launch_exact_init(xs, exact_lanes(b, n_slow));
        if (int rc = run_exact_segments(b, n_slow, min_start)) return rc;
        if (!go_async) {
            HIPCHK(hipStreamSynchronize(s));
            if (int rc = retry_device_limits(b, n_slow, true, 0xFFFFFFFFu)) return rc;
        }
        hipEventRecord(slow1, s);
    }
    float ms = 0;
    if (imported_next && !n_slow) {  // the solve ended at its event count (waited for above); the next tile's import is still in flight
        HIPCHK(hipEventElapsedTime(&ms, b->ev_start, b->ev_counted));
        b->next_imported = true;
        b->next_inputs = next_inputs;
        b->events_fresh = true;  // (the gated import ran: it left the event words ready)
    } else {
        HIPCHK(hipEventRecord(b->ev_end, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipEventElapsedTime(&ms, b->ev_start, b->ev_end));
    }
    b->solve_device_ms = ms;
    auto sum_pairs = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
        double total = 0;
        for (auto &pr : v) {
            float t = 0;
            hipEventElapsedTime(&t, pr.first, pr.second);
            total += t;
        }
        return total;
    };
    b->arith_kernel_ms = sum_pairs(tm.reg_pairs);
    b->dyn_kernel_ms = sum_pairs(tm.dyn_pairs);
    for (int k = 0; k < (int)N_CLS; k++) b->cls_kernel_ms[k] = sum_pairs(tm.cls_pairs[k]);
    if (n_slow) {
        float t = 0;
        hipEventElapsedTime(&t, slow0, slow1);
        b->slow_path_ms = t;
    }
    b->solved = true;
    if (!n_slow) b->slow_res.clear();
    if (b->pending) return (int)n_slow;  // their outcome is not known yet
    return count_not_solved(b);
}
// the solve, and -- under a caller-supplied BlackBoxFunctionSolver -- the rounds that answer the Brillig programs' internal black-box calls
// (batch_exact.cpp resolve_internal_calls): every round answers all instances waiting at such a call and re-solves (a few lanes on the exact
// kernels, a sizeable part of the batch through the level schedule again: batch_solve_impl's choice for resolved foreign calls)
static int solve_with_internal_calls(acvm_batch *b, const void *next_inputs) {
    int rc = batch_solve_impl(b, next_inputs);
    while (rc >= 0) {
        const int answered = resolve_internal_calls(b);
        if (answered < 0) return answered;
        if (!answered) break;
        rc = batch_solve_impl(b, nullptr);
    }
    return rc;
}
int acvm_batch_solve(acvm_batch_t *b) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    return solve_with_internal_calls(b, nullptr);
} ABI_CATCH
int acvm_batch_solve_then_import(acvm_batch_t *b, const void *d_next_values_be32) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    // (resumed foreign calls, stepping and a caller-supplied solver keep the plain solve: nothing is imported behind them)
    const bool plain = !d_next_values_be32 || b->solved || b->stepping || b->has_solver || b->force_slow;
    return solve_with_internal_calls(b, plain ? nullptr : d_next_values_be32);
} ABI_CATCH


