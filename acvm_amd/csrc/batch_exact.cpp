// batch_exact.cpp -- the exact in-order path of the batch handle: instances that left the generic path are re-solved from their event on by
// the exact kernels (one lane per instance), in place or -- for the node driver -- in a side table beside the next tile; Brillig VM limits
// and their retries; stepping (ACVM::solve_opcode, acvm/src/pwg/mod.rs:243-303); caller-supplied BlackBoxFunctionSolver callbacks; the
// foreign-call round trip (mod.rs:203-228).
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

int ensure_slow_capacity(acvm_batch *b, uint32_t n) {
    if (n <= b->slow_cap) return 0;
    for (void *p : {(void *)b->d_slow_ids, (void *)b->d_assigned, (void *)b->d_slow_res, (void *)b->d_slow_start})
        if (p) hipFree(p);
    b->d_slow_ids = nullptr; b->d_assigned = nullptr; b->d_slow_res = nullptr; b->d_slow_start = nullptr;
    HIPCHK(hipMalloc((void **)&b->d_slow_ids, (size_t)n * 4));
    HIPCHK(hipMalloc((void **)&b->d_slow_start, (size_t)n * 4));
    HIPCHK(hipMalloc((void **)&b->d_assigned, (size_t)n * (b->n_words ? b->n_words : 1) * 4));
    HIPCHK(hipMalloc((void **)&b->d_slow_res, (size_t)n * sizeof(SlowResult)));
    b->slow_cap = n;
    return 0;
}

ExactLanes exact_lanes(acvm_batch *b, uint32_t n_slow) {
    FcLanes fc{b->d_fc_pend_desc, b->fc_pend_desc_words, b->d_fc_pend_vals, b->fc_pend_vals_cap};
    return ExactLanes{b->xids(), n_slow, b->d_assigned, b->d_slow_start, b->d_slow_res, fc, b->br_retry_active ? b->d_br_lane : nullptr};
}

// Foreign-call round trip, device side: the buffers a pending call's inputs are written to (per exact lane, FcLanes::pend_*) and
// the store of the results the host resolved (per opcode slot and instance, FcStoreSlot): dirty slots are rebuilt and uploaded.
int upload_fc_tables(acvm_batch *b, uint32_t n_slow) {
    const Plan &p = b->plan();
    if (!p.has_foreign_calls) return 0;
    b->fc_lane.resize(n_slow);
    const uint32_t pend_words = 1 + p.fc_max_inputs, pend_vals = (uint32_t)std::max<uint64_t>(1, p.fc_pending_vals);
    if (n_slow > b->fc_lanes_cap) {
        for (void *q : {(void *)b->d_fc_pend_desc, (void *)b->d_fc_pend_vals})
            if (q) hipFree(q);
        b->d_fc_pend_desc = nullptr;
        b->d_fc_pend_vals = nullptr;
        b->fc_lanes_cap = n_slow;
        b->fc_pend_desc_words = pend_words;
        b->fc_pend_vals_cap = pend_vals;
        HIPCHK(hipMalloc((void **)&b->d_fc_pend_desc, (size_t)pend_words * n_slow * 4));
        HIPCHK(hipMalloc((void **)&b->d_fc_pend_vals, (size_t)pend_vals * 2 * n_slow * sizeof(uint4)));
    }
    bool any = false;
    for (size_t si = 0; si < b->fc_slots.size(); si++) {
        auto &sl = b->fc_slots[si];
        if (!sl.dirty) continue;
        sl.dirty = false;
        any = true;
        uint32_t desc_words = 1, vals = 1;
        for (auto &kv : sl.inst) {
            uint32_t dw = 1, nv = 0;
            for (auto &res : kv.second) {
                dw += 1 + 2 * (uint32_t)res.size();
                for (auto &v : res) nv += (uint32_t)v.vals.size();
            }
            desc_words = std::max(desc_words, dw);
            vals = std::max(vals, nv);
        }
        if (desc_words > sl.desc_words || vals > sl.vals_cap) {
            for (void *q : {(void *)sl.d_desc, (void *)sl.d_vals})
                if (q) hipFree(q);
            sl.d_desc = nullptr;
            sl.d_vals = nullptr;
            sl.desc_words = desc_words + 8;
            sl.vals_cap = vals + 8;
            HIPCHK(hipMalloc((void **)&sl.d_desc, (size_t)sl.desc_words * b->Bp * 4));
            HIPCHK(hipMalloc((void **)&sl.d_vals, (size_t)sl.vals_cap * 2 * b->Bp * sizeof(uint4)));
        }
        // word w of instance j at desc[w * Bp + j]; value i: halves at (2 i) * Bp + j and (2 i + 1) * Bp + j, 4 words each
        std::vector<uint32_t> desc((size_t)sl.desc_words * b->Bp, 0), vbuf((size_t)sl.vals_cap * 2 * b->Bp * 4, 0);
        for (auto &kv : sl.inst) {
            const uint64_t j = kv.first;
            uint32_t w = 0, vi = 0;
            desc[(size_t)(w++) * b->Bp + j] = (uint32_t)kv.second.size();
            for (auto &res : kv.second) {
                if (!res.empty() && res[0].fail) {  // a failing internal call: a marker instead of a value count, nothing follows
                    desc[(size_t)(w++) * b->Bp + j] = 0xFFFFFFF0u + std::min<uint32_t>(res[0].fail, 3u);
                    continue;
                }
                desc[(size_t)(w++) * b->Bp + j] = (uint32_t)res.size();
                for (auto &v : res) {
                    desc[(size_t)(w++) * b->Bp + j] = v.is_array ? 1u : 0u;
                    desc[(size_t)(w++) * b->Bp + j] = (uint32_t)v.vals.size();
                    for (auto &xh : v.vals) {
                        const FrH x = frh::to_device_form(xh);
                        memcpy(&vbuf[((size_t)(2 * vi) * b->Bp + j) * 4], &x.l[0], 16);
                        memcpy(&vbuf[((size_t)(2 * vi + 1) * b->Bp + j) * 4], &x.l[2], 16);
                        vi++;
                    }
                }
            }
        }
        HIPCHK(hipMemcpy(sl.d_desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sl.d_vals, vbuf.data(), vbuf.size() * 4, hipMemcpyHostToDevice));
    }
    if (any && b->d_fc_store) {  // the kernels reach the tables through this array: the pointers may have moved
        std::vector<FcStoreSlot> tab(b->fc_slots.size());
        for (size_t si = 0; si < tab.size(); si++) tab[si] = FcStoreSlot{b->fc_slots[si].d_desc, b->fc_slots[si].d_vals};
        HIPCHK(hipMemcpy(b->d_fc_store, tab.data(), tab.size() * sizeof(FcStoreSlot), hipMemcpyHostToDevice));
    }
    return 0;
}
// One Pedersen / FixedBaseScalarMul / SchnorrVerify opcode through the caller's BlackBoxFunctionSolver callbacks
// (blackbox_solver/src/lib.rs:27-45) for the instances of the level schedule (exact == false, all B instances) or for the
// exact lanes. Inputs leave the device as canonical big-endian bytes, outputs come back the same way. All instances are
// gathered ONCE (one kernel, one copy), the callbacks run in one loop -- or in ONE call when the vtable has the *_batch
// member -- and the results are scattered once; a pass is capped at 2^18 instances only to bound the staging buffers.
int run_host_blackbox(acvm_batch *b, uint32_t opcode, bool exact, uint32_t n_slow) {
    const Plan &p = b->plan();
    hipStream_t s = b->stream;
    const uint32_t *rec = &p.prog[p.prog_offset[opcode]];
    std::vector<uint32_t> sel, outs;
    uint32_t func = 0;
    switch (rec[0]) {
    case PK_FIXED_BASE: func = BB_FIXED_BASE_SCALAR_MUL; sel = {rec[2], rec[3]}; outs = {rec[4], rec[5], rec[6], rec[7]}; break;
    case PK_PEDERSEN: func = BB_PEDERSEN; sel.assign(rec + 8, rec + 8 + rec[3]); outs = {rec[4], rec[5], rec[6], rec[7]}; break;
    case PK_SCHNORR: func = BB_SCHNORR_VERIFY; sel = {rec[2], rec[3]}; sel.insert(sel.end(), rec + 8, rec + 8 + rec[4] + rec[5]); outs = {rec[6], rec[7]}; break;
    default: return set_err(ACVM_E_INVALID, "not a black box function of the solver trait");
    }
    const uint32_t n_sel = (uint32_t)sel.size(), n_out = (uint32_t)outs.size() / 2;
    const uint32_t n_total = exact ? n_slow : b->B;
    if (!n_total) return 0;
    const uint32_t chunk = std::min<uint32_t>(n_total, 1u << 18);
    const size_t in_row = (size_t)std::max<uint32_t>(n_sel, 1) * 32, out_row = (size_t)n_out * 32;
    // arena: sel | outs | active | in | rc | vals
    const size_t o_sel = 0, o_outs = o_sel + align256((size_t)std::max<uint32_t>(n_sel, 1) * 4), o_active = o_outs + align256(outs.size() * 4),
                 o_in = o_active + align256(n_total), o_rc = o_in + align256(chunk * in_row), o_vals = o_rc + align256(chunk);
    if (int rc = stage_reserve(b, o_vals + align256(chunk * out_row))) return rc;
    uint32_t *d_sel = (uint32_t *)(b->d_stage + o_sel), *d_outs = (uint32_t *)(b->d_stage + o_outs);
    uint8_t *d_active = b->d_stage + o_active, *d_in = b->d_stage + o_in, *d_rc = b->d_stage + o_rc, *d_vals = b->d_stage + o_vals;
    std::vector<uint8_t> active(n_total, 1);
    if (n_sel) HIPCHK(hipMemcpyAsync(d_sel, sel.data(), (size_t)n_sel * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_outs, outs.data(), outs.size() * 4, hipMemcpyHostToDevice, s));
    const ExactLanes L = exact_lanes(b, n_slow);
    if (exact) {
        launch_hostbb_precheck(s, L, opcode, d_sel, n_sel, d_active);
        HIPCHK(hipMemcpyAsync(active.data(), d_active, n_total, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    std::vector<uint8_t> in(chunk * in_row), rc(chunk), vals(chunk * out_row);
    static constexpr size_t ERR_STRIDE = 200;
    const acvm_bb_solver_t &sv = b->solver;
    for (uint32_t first = 0; first < n_total; first += chunk) {
        const uint32_t m = std::min(chunk, n_total - first);
        // (the exact lanes live in the side table under witness-slot reuse -- rows = witness indices, lane t = the t-th flagged instance --
        // the level schedule's instances in the level table, whose rows slot_of maps)
        if (exact) launch_hostbb_gather(s, b->xW(), b->xBp(), b->xids(), first, m, d_sel, n_sel, d_in);
        else launch_hostbb_gather(s, b->d_W, b->Bp, nullptr, first, m, d_sel, n_sel, d_in, b->dp.slot_of);
        if (n_sel) HIPCHK(hipMemcpyAsync(in.data(), d_in, (size_t)m * n_sel * 32, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::fill(vals.begin(), vals.end(), 0);
        // the instances that really call the solver, packed
        std::vector<uint32_t> who;
        for (uint32_t i = 0; i < m; i++) {
            rc[i] = 255;
            if (active[first + i]) who.push_back(i);
        }
        const bool batched = (rec[0] == PK_FIXED_BASE && sv.fixed_base_scalar_mul_batch) || (rec[0] == PK_PEDERSEN && sv.pedersen_batch) ||
                             (rec[0] == PK_SCHNORR && sv.schnorr_verify_batch);
        auto instance_of = [&](uint32_t i) { return exact ? b->slow_ids[first + i] : first + i; };
        if (batched && !who.empty()) {
            const size_t n = who.size();
            std::vector<uint8_t> brc(n, 0), bout(n * 64, 0);
            std::vector<char> berr(n * ERR_STRIDE, 0);
            int r = 0;
            if (rec[0] == PK_FIXED_BASE) {
                std::vector<uint8_t> lh(n * 64);
                for (size_t q = 0; q < n; q++) memcpy(&lh[q * 64], &in[(size_t)who[q] * in_row], 64);
                r = sv.fixed_base_scalar_mul_batch(sv.ctx, n, lh.data(), bout.data(), brc.data(), berr.data(), ERR_STRIDE);
            } else if (rec[0] == PK_PEDERSEN) {
                const size_t k = rec[3];
                std::vector<uint8_t> pin(n * std::max<size_t>(k, 1) * 32);
                for (size_t q = 0; q < n; q++) memcpy(&pin[q * k * 32], &in[(size_t)who[q] * in_row], k * 32);
                r = sv.pedersen_batch(sv.ctx, n, pin.data(), k, rec[2], bout.data(), brc.data(), berr.data(), ERR_STRIDE);
            } else {  // to_u8_vec (signature/mod.rs:5-18): the last big-endian byte of each witness
                const uint32_t n_sig = rec[4], n_msg = rec[5];
                std::vector<uint8_t> pk(n * 64), sig(n * std::max<uint32_t>(n_sig, 1)), msg(n * std::max<uint32_t>(n_msg, 1)), ok(n, 0);
                for (size_t q = 0; q < n; q++) {
                    const uint8_t *a = &in[(size_t)who[q] * in_row];
                    memcpy(&pk[q * 64], a, 64);
                    for (uint32_t k = 0; k < n_sig; k++) sig[q * n_sig + k] = a[(size_t)(2 + k) * 32 + 31];
                    for (uint32_t k = 0; k < n_msg; k++) msg[q * n_msg + k] = a[(size_t)(2 + n_sig + k) * 32 + 31];
                }
                r = sv.schnorr_verify_batch(sv.ctx, n, pk.data(), sig.data(), n_sig, msg.data(), n_msg, ok.data(), brc.data(), berr.data(), ERR_STRIDE);
                for (size_t q = 0; q < n; q++) bout[q * 64 + 31] = ok[q] ? 1 : 0;
            }
            for (size_t q = 0; q < n; q++) {
                const uint32_t i = who[q];
                const int ri = r != 0 ? 3 : brc[q];  // a failing batch call fails every instance of it like a panic
                rc[i] = (uint8_t)(ri > 2 ? 3 : ri);
                memcpy(&vals[(size_t)i * out_row], &bout[q * 64], out_row);
                if (rc[i] != 0) {
                    berr[q * ERR_STRIDE + ERR_STRIDE - 1] = 0;
                    b->host_bb_msg[instance_of(i)] = r != 0 ? "batched BlackBoxFunctionSolver call failed" : &berr[q * ERR_STRIDE];
                }
            }
        } else {
            char err[ERR_STRIDE];
            for (uint32_t i : who) {
                const uint8_t *a = &in[(size_t)i * in_row];
                uint8_t *o = &vals[(size_t)i * out_row];
                err[0] = 0;
                int r = 0;
                if (rec[0] == PK_FIXED_BASE) r = sv.fixed_base_scalar_mul(sv.ctx, a, a + 32, o, o + 32, err, sizeof err);
                else if (rec[0] == PK_PEDERSEN) r = sv.pedersen(sv.ctx, a, rec[3], rec[2], o, o + 32, err, sizeof err);
                else {  // to_u8_vec (signature/mod.rs:5-18): the last big-endian byte of each witness
                    const uint32_t n_sig = rec[4], n_msg = rec[5];
                    std::vector<uint8_t> sig(n_sig + 1), msg(n_msg + 1);
                    for (uint32_t k = 0; k < n_sig; k++) sig[k] = a[(size_t)(2 + k) * 32 + 31];
                    for (uint32_t k = 0; k < n_msg; k++) msg[k] = a[(size_t)(2 + n_sig + k) * 32 + 31];
                    uint8_t ok = 0;
                    r = sv.schnorr_verify(sv.ctx, a, a + 32, sig.data(), n_sig, msg.data(), n_msg, &ok, err, sizeof err);
                    o[31] = ok ? 1 : 0;
                }
                rc[i] = (uint8_t)(r < 0 || r > 2 ? 3 : r);
                if (r != 0) b->host_bb_msg[instance_of(i)] = err;
            }
        }
        HIPCHK(hipMemcpyAsync(d_rc, rc.data(), m, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_vals, vals.data(), (size_t)m * out_row, hipMemcpyHostToDevice, s));
        if (exact) launch_hostbb_apply_exact(s, b->xW(), b->xBp(), L, first, m, opcode, func, d_outs, n_out, d_active, d_rc, d_vals);
        else launch_hostbb_apply_level(s, b->d_W, b->Bp, first, m, opcode, func, d_outs, n_out, d_rc, d_vals, b->d_event, b->dp.slot_of);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    return 0;
}

// run the exact in-order kernels over the current lanes from opcode min_start on and fetch the outcomes
// (stepping: the lanes executed every earlier opcode themselves, nothing is replayed; only opcodes [min_start, end_opcode) run)
int run_exact_segments(acvm_batch *b, uint32_t n_slow, uint32_t min_start, bool replay, uint32_t end_opcode) {
    const Plan &p = b->plan();
    hipStream_t s = b->xstream();  // the batch's stream, or the side stream of an asynchronous job
    const ExactLanes L = exact_lanes(b, n_slow);
    const DeviceProgram xdp = b->xdp();
    // Memory side effects of the opcodes before the earliest event are replayed by the kernel, so the run starts at the first opcode
    // when the circuit has memory blocks, else at the earliest event. One launch covers every class (kernels_brillig.hip
    // exact_run_kernel); only the opcodes of a caller-supplied BlackBoxFunctionSolver split it (host callbacks in between).
    const bool has_mem = replay && p.mem_cells != 0;
    const ExactScratch sc{b->xscratch(CLS_HASH), b->xscratch(CLS_GRUMPKIN), b->br_retry_active ? b->d_br_scratch : b->xscratch(CLS_BRILLIG)};
    const uint32_t end = std::min(end_opcode, p.n_opcodes);
    uint32_t at = has_mem ? 0u : std::min(min_start, end);
    while (at < end) {
        uint32_t stop = at;
        while (stop < end && p.prog_class[stop] != CLS_HOSTBB) stop++;
        launch_exact_run(s, b->xW(), b->xBp(), xdp, L, at, stop, has_mem, b->d_prog_class, sc);
        if (stop < end) {
            if (stop >= min_start)  // (no lane stands before an opcode in front of the earliest start)
                if (int rc = run_host_blackbox(b, stop, true, n_slow)) return rc;
            stop++;
        }
        at = stop;
    }
    launch_exact_finish(s, L, b->stepping ? p.n_opcodes : 0u);
    HIPCHK(hipGetLastError());
    b->slow_res.resize(n_slow);
    // (a copy into pageable host memory blocks the caller until the stream has drained: an asynchronous job fetches its lanes' results
    // when it is collected, batch_finish_pending)
    if (!b->pending) HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
    b->pend_host_valid = false;
    return 0;
}

// The Brillig VM of the reference has no limits: memory grows on write (brillig_vm/src/memory.rs:27-39), a program may run any number
// of steps and nest calls to any depth (lib.rs:154-307). The kernels run with a memory capacity, a step limit and a call-stack depth
// (BrilligLimits); a lane that reaches one ends its pass with DE_PANIC and one of the three device-limit codes. Such lanes are
// RETRIED here: their opcode runs again (a failed VM run has no side effects: outputs are inserted after it finishes) with the
// limit that was hit raised -- memory to twice the cell the write wanted, steps and depth sixteen-fold -- in a scratch that holds
// only the retried lanes, until nothing hits a limit or a stated maximum of the library is reached (tuning.hpp: 2^26 steps, 2^16
// frames, 2^22 cells by default). Past a maximum THAT INSTANCE ends with status Failure / ACVM_ERR_DEVICE_LIMIT -- an outcome the
// reference does not have and the header says so: "this library could not finish the instance, run it with the reference" -- and every
// other instance of the batch keeps its result (round 3 failed the whole solve call and lost them: the reference's caller loop,
// acvm_js/src/execute.rs:60-119, loses one instance at most). Called with the lanes' results on the host (stream synchronised).
bool is_device_limit(const SlowResult &r) {
    return r.status == ACVM_STATUS_FAILURE && r.err == ACVM_ERR_PANIC && (r.msg == 17u || r.msg == 18u || r.msg == 28u);
}
// the lane's final word: Failure / ACVM_ERR_DEVICE_LIMIT at its Brillig opcode, aux0 = the limit that was reached, aux1 = its value
static void give_up_lane(SlowResult &r, uint32_t kind, uint64_t limit, uint64_t wanted) {
    const uint32_t opcode = r.opcode_index;
    memset(&r, 0, sizeof r);
    r.status = ACVM_STATUS_FAILURE;
    r.err = ACVM_ERR_DEVICE_LIMIT;
    r.opcode_index = opcode;
    r.aux0 = kind;
    r.aux1 = (uint32_t)std::min<uint64_t>(limit, 0xFFFFFFFFu);
    r.msg = 29u;  // DM_DEVICE_LIMIT (ops_common.hpp): format_message words it
    r.x0 = (uint32_t)std::min<uint64_t>(wanted, 0xFFFFFFFFu);
}
int retry_device_limits(acvm_batch *b, uint32_t n_slow, bool replay, uint32_t end_opcode) {
    const Plan &p = b->plan();
    const Tuning &tn = p.tune;
    hipStream_t s = b->xstream();
    const BrilligLimits base = b->dp.brillig;
    const uint64_t max_steps = 1ull << (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_steps_max_log2, 0), 31);
    const uint64_t max_depth = (uint64_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_call_depth_max, 1), 1 << 24);
    const uint64_t max_cells = 1ull << (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_mem_max_log2, 0), 30);
    BrilligLimits lim = base;
    int rc = 0;
    for (;;) {
        // the memory a retried lane runs with so far (0 = the planner's estimate of its record)
        uint64_t record_cells = 0;
        for (uint32_t t = 0; t < n_slow; t++) {
            const SlowResult &r = b->slow_res[t];
            if (is_device_limit(r) && r.opcode_index < p.n_opcodes && p.prog[p.prog_offset[r.opcode_index]] == PK_BRILLIG)
                record_cells = std::max<uint64_t>(record_cells, p.prog[p.prog_offset[r.opcode_index] + 8]);
        }
        const uint64_t cur_cells = lim.mem_cap ? lim.mem_cap : record_cells;
        // lanes past a stated maximum are final; the rest is retried with the limits they reached raised
        std::vector<uint32_t> lanes;
        bool hit_steps = false, hit_depth = false, hit_mem = false;
        uint64_t want_cells = 0;
        for (uint32_t t = 0; t < n_slow; t++) {
            SlowResult &r = b->slow_res[t];
            if (!is_device_limit(r)) continue;
            if (r.msg == 18u && lim.steps >= max_steps) { give_up_lane(r, ACVM_LIMIT_BRILLIG_STEPS, max_steps, 0); continue; }
            if (r.msg == 28u && lim.call_depth >= max_depth) { give_up_lane(r, ACVM_LIMIT_BRILLIG_CALL_DEPTH, max_depth, 0); continue; }
            // (the lane's own capacity so far: the raised one of an earlier pass, else the planner's estimate of ITS record -- not the largest
            // estimate among the lanes of this pass, which gave up a lane whose own need still fitted)
            const uint64_t own_cells = lim.mem_cap ? lim.mem_cap : (r.opcode_index < p.n_opcodes && p.prog[p.prog_offset[r.opcode_index]] == PK_BRILLIG ? p.prog[p.prog_offset[r.opcode_index] + 8] : 0u);
            if (r.msg == 17u && ((uint64_t)r.x0 + 1 > max_cells || own_cells >= max_cells)) { give_up_lane(r, ACVM_LIMIT_BRILLIG_MEMORY, max_cells, r.x0); continue; }
            lanes.push_back(t);
            hit_steps |= r.msg == 18u;
            hit_depth |= r.msg == 28u;
            if (r.msg == 17u) {
                hit_mem = true;
                want_cells = std::max<uint64_t>(want_cells, (uint64_t)r.x0 + 1);
            }
        }
        if (lanes.empty()) break;
        if (hit_steps) lim.steps = (uint32_t)std::min<uint64_t>((uint64_t)lim.steps * 16, max_steps);
        if (hit_depth) lim.call_depth = (uint32_t)std::min<uint64_t>((uint64_t)lim.call_depth * 16, max_depth);
        uint64_t cells = std::max<uint64_t>(cur_cells, 64);
        if (hit_mem) cells = std::min<uint64_t>(std::max<uint64_t>(2 * want_cells, 4 * cells), max_cells);
        lim.mem_cap = (uint32_t)cells;
        lim.stride = ((uint64_t)lanes.size() + 63) / 64 * 64;
        const uint64_t words = ((uint64_t)b->br_max_regs + cells) * 8 + lim.call_depth + cells / 4 + 16;
        const size_t bytes = (size_t)words * lim.stride * 4;
        if (bytes > b->br_scratch_bytes) {
            if (b->d_br_scratch) hipFree(b->d_br_scratch);
            b->d_br_scratch = nullptr;
            b->br_scratch_bytes = 0;
            if (hipMalloc((void **)&b->d_br_scratch, bytes) != hipSuccess) {
                (void)hipGetLastError();  // the device cannot hold the VM scratch of these lanes: they are final too
                for (uint32_t t : lanes) give_up_lane(b->slow_res[t], ACVM_LIMIT_DEVICE_MEMORY, bytes >> 20, 0);
                break;
            }
            b->br_scratch_bytes = bytes;
        }
        if (n_slow > b->br_lane_cap) {
            if (b->d_br_lane) hipFree(b->d_br_lane);
            b->d_br_lane = nullptr;
            HIPCHK(hipMalloc((void **)&b->d_br_lane, (size_t)n_slow * 4));
            b->br_lane_cap = n_slow;
        }
        std::vector<uint32_t> col(n_slow, 0xFFFFFFFFu);
        uint32_t min_start = 0xFFFFFFFFu;
        for (size_t i = 0; i < lanes.size(); i++) {
            const uint32_t t = lanes[i];
            col[t] = (uint32_t)i;
            const uint32_t at = b->slow_res[t].opcode_index;
            b->slow_start[t] = at;  // the opcode runs again
            min_start = std::min(min_start, at);
            memset(&b->slow_res[t], 0, sizeof(SlowResult));
            b->slow_res[t].status = ACVM_STATUS_IN_PROGRESS;
        }
        HIPCHK(hipMemcpyAsync(b->d_br_lane, col.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_slow_res, b->slow_res.data(), (size_t)n_slow * sizeof(SlowResult), hipMemcpyHostToDevice, s));
        // (the lanes that were given up stay Failure on the host and on the device: the exact kernels skip a lane that is not InProgress;
        // run_exact_segments fetches the device's records back, which is why the host copy of those lanes is restored below)
        std::vector<std::pair<uint32_t, SlowResult>> final_lanes;
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_res[t].err == ACVM_ERR_DEVICE_LIMIT) final_lanes.push_back({t, b->slow_res[t]});
        b->dp.brillig = lim;
        b->br_retry_active = true;
        rc = run_exact_segments(b, n_slow, min_start, replay, end_opcode);
        if (!rc && b->pending) {  // (an asynchronous job leaves its results on the device: fetch them for the next look at the limits)
            if (hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s) != hipSuccess) rc = set_err(ACVM_E_DEVICE, "hipMemcpyAsync failed in a Brillig retry pass");
        }
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_err(ACVM_E_DEVICE, "hipStreamSynchronize failed in a Brillig retry pass");
        for (auto &fl : final_lanes) b->slow_res[fl.first] = fl.second;
        b->dp.brillig = base;
        b->br_retry_active = false;
        b->n_brillig_retries++;
        if (rc) break;
    }
    // The lanes that were given up are final on the DEVICE too: a caller that keeps stepping (solve_stepping) fetches the device's records
    // after every step, and a lane whose device record still said "panic at a device limit" came back as ACVM_ERR_PANIC and was retried
    // up to the maximum again at every later step. (Every exit of the loop above passes here: no lane left to retry, a failed allocation.)
    if (!rc) {
        bool any_final = false;
        for (uint32_t t = 0; t < n_slow; t++) any_final |= b->slow_res[t].err == ACVM_ERR_DEVICE_LIMIT;
        if (any_final) HIPCHK(hipMemcpyAsync(b->d_slow_res, b->slow_res.data(), (size_t)n_slow * sizeof(SlowResult), hipMemcpyHostToDevice, s));
    }
    return rc;
}

int count_not_solved(acvm_batch *b) {
    int n = 0;
    for (auto &r : b->slow_res)
        if (r.status != ACVM_STATUS_SOLVED) n++;
    return n;
}

// continue the instances whose pending foreign call was resolved (ACVM::solve after resolve_pending_foreign_call)
int solve_resume(acvm_batch *b) {
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    uint32_t min_start = 0xFFFFFFFFu;
    std::vector<uint32_t> resumed;
    for (uint32_t t = 0; t < n_slow; t++)
        if (b->slow_res[t].status == ACVM_STATUS_REQUIRES_FOREIGN_CALL && b->fc_lane[t].resolved_new) {
            resumed.push_back(t);
            b->fc_lane[t].resolved_new = false;
            b->slow_start[t] = b->slow_res[t].opcode_index;
            min_start = std::min(min_start, b->slow_start[t]);
            b->slow_res[t].status = ACVM_STATUS_IN_PROGRESS;
        }
    if (resumed.empty()) return count_not_solved(b);
    if (int rc = upload_fc_tables(b, n_slow)) return rc;
    hipStream_t s = b->stream;
    HIPCHK(hipEventRecord(b->ev_start, s));
    HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(b->d_slow_res, b->slow_res.data(), (size_t)n_slow * sizeof(SlowResult), hipMemcpyHostToDevice, s));
    if (int rc = run_exact_segments(b, n_slow, min_start)) return rc;
    HIPCHK(hipStreamSynchronize(s));
    if (int rc = retry_device_limits(b, n_slow, true, 0xFFFFFFFFu)) return rc;
    HIPCHK(hipEventRecord(b->ev_end, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev_start, b->ev_end));
    b->solve_device_ms = ms;
    b->slow_path_ms = ms;
    return count_not_solved(b);
}

// acvm_batch_solve_opcode (one == true: ACVM::solve_opcode, pwg/mod.rs:243-303) and acvm_batch_solve after some steps
// (one == false: the loop of ACVM::solve :236-241 over what is left). Every instance is an exact lane whose instruction
// pointer is slow_start[t]; see include/acvm_amd.h for the batch semantics.
int solve_stepping(acvm_batch *b, bool one) {
    const Plan &p = b->plan();
    hipStream_t s = b->stream;
    const uint32_t n_slow = b->B;
    if (!b->stepping) {
        b->slow_ids.resize(n_slow);
        b->events_clean = false;
        for (uint32_t j = 0; j < n_slow; j++) { b->slow_ids[j] = j; b->slow_index[j] = (int32_t)j; }
        b->slow_start.assign(n_slow, 0);
        std::fill(b->h_event.begin(), b->h_event.end(), 0u);
        b->host_bb_msg.clear();
        b->stepping = true;
        b->solved = true;
        SlowResult fresh;
        memset(&fresh, 0, sizeof fresh);
        fresh.status = ACVM_STATUS_IN_PROGRESS;
        b->slow_res.assign(n_slow, fresh);
        if (!n_slow) return 0;
        if (int rc = ensure_slow_capacity(b, n_slow)) return rc;
        launch_fill_u32(s, b->d_event, 0u, b->B);  // no column is scaled: the exact kernels write plain values
        HIPCHK(hipMemcpyAsync(b->d_slow_ids, b->slow_ids.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        launch_init_assigned(s, b->d_assigned, n_slow, b->n_words, p.n_witnesses, b->d_producer, b->d_slow_start);
        b->fc_lane.assign(n_slow, acvm_batch::FcLaneState());
        if (int rc = upload_fc_tables(b, n_slow)) return rc;
        launch_exact_init(s, exact_lanes(b, n_slow));
    } else {
        bool any = false;
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_res[t].status == ACVM_STATUS_REQUIRES_FOREIGN_CALL && b->fc_lane[t].resolved_new) {
                b->fc_lane[t].resolved_new = false;
                b->slow_start[t] = b->slow_res[t].opcode_index;  // the opcode re-runs its VM (mod.rs:220-227)
                b->slow_res[t].status = ACVM_STATUS_IN_PROGRESS;
                any = true;
            }
        if (any) {
            if (int rc = upload_fc_tables(b, n_slow)) return rc;
            HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(b->d_slow_res, b->slow_res.data(), (size_t)n_slow * sizeof(SlowResult), hipMemcpyHostToDevice, s));
        }
    }
    if (!n_slow) return 0;
    uint32_t ip = 0xFFFFFFFFu;
    for (uint32_t t = 0; t < n_slow; t++)
        if (b->slow_res[t].status == ACVM_STATUS_IN_PROGRESS) ip = std::min(ip, b->slow_start[t]);
    if (ip == 0xFFFFFFFFu) return count_not_solved(b);
    const uint32_t end = one ? std::min(ip + 1, p.n_opcodes) : p.n_opcodes;
    if (one) {  // advance the instruction pointers on the host between the opcode and the Solved test
        if (ip < p.n_opcodes)
            if (int rc = run_exact_segments(b, n_slow, ip, false, end)) return rc;
        b->slow_res.resize(n_slow);
        HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (int rc = retry_device_limits(b, n_slow, false, end)) return rc;  // (a retried lane stands on `ip` again and re-runs only that opcode)
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_res[t].status == ACVM_STATUS_IN_PROGRESS && b->slow_start[t] <= ip) b->slow_start[t] = end;
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        launch_exact_finish(s, exact_lanes(b, n_slow), p.n_opcodes);
        HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
    } else {
        if (int rc = run_exact_segments(b, n_slow, ip, false)) return rc;  // ends with the Solved test of the lanes still running
        HIPCHK(hipStreamSynchronize(s));
        if (int rc = retry_device_limits(b, n_slow, false, 0xFFFFFFFFu)) return rc;
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_start[t] < p.n_opcodes) b->slow_start[t] = p.n_opcodes;
        // (run_exact_segments finishes only lanes whose pointer is at the end: publish the pointers first)
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        launch_exact_finish(s, exact_lanes(b, n_slow), p.n_opcodes);
        HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    b->pend_host_valid = false;
    for (uint32_t t = 0; t < n_slow; t++) {
        SlowResult &r = b->slow_res[t];
        if (r.status == ACVM_STATUS_IN_PROGRESS) r.opcode_index = b->slow_start[t];  // ACVM::instruction_pointer
        else if (r.status == ACVM_STATUS_REQUIRES_FOREIGN_CALL || r.status == ACVM_STATUS_FAILURE) b->slow_start[t] = r.opcode_index;
    }
    return count_not_solved(b);
}

int acvm_batch_solve_opcode(acvm_batch_t *b) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    if (!b->inputs_set && !b->plan().initial_ids.empty()) return set_err(ACVM_E_STATE, "initial witness not set");
    if (b->solved && !b->stepping) return set_err(ACVM_E_STATE, "acvm_batch_solve_opcode after acvm_batch_solve: reset the batch first");
    if (b->reuse()) return set_err(ACVM_E_UNSUPPORTED, "stepping needs the full witness table: not with ACVM_BATCH_REUSE_SLOTS");
    HIPCHK(hipSetDevice(b->device));
    int rc = solve_stepping(b, true);
    // a Brillig opcode that stopped at an internal black-box call (the caller's solver, resolve_internal_calls) finishes inside THIS step: its
    // lanes are answered and the step runs once more -- the smallest instruction pointer among the InProgress lanes is then theirs
    while (rc >= 0) {
        const int answered = resolve_internal_calls(b);
        if (answered < 0) return answered;
        if (!answered) break;
        rc = solve_stepping(b, true);
    }
    return rc;
} ABI_CATCH

// per-launch HIP-event pairs of one solve (profiling on)

int ensure_side_table(acvm_batch *b, uint32_t n_lanes, bool own_scratch) {
    const Plan &p = b->plan();
    const uint64_t lanes = ((uint64_t)std::max<uint32_t>(n_lanes, 1) + 63) / 64 * 64;
    if (lanes > b->x_cap) {
        // a table of all witnesses per flagged instance: refuse when that is more than the level table itself
        const size_t need = (size_t)p.n_witnesses * 2 * lanes * sizeof(uint4), level_table = (size_t)(b->reuse() ? p.n_slots : p.n_witnesses) * 2 * b->Bp * sizeof(uint4);
        if (need > level_table && need > (8ull << 30))  // (a small batch pads to 64 lanes either way: below 8 GiB the table is simply allocated)
            return set_err(ACVM_E_UNSUPPORTED, "slot reuse: " + std::to_string(n_lanes) + " instances left the generic path; their own table would exceed "
                                               "the level table -- solve this tile without ACVM_BATCH_REUSE_SLOTS");
        for (void *q : {(void *)b->d_Wx, (void *)b->d_Memx, (void *)b->d_ids_x})
            if (q) hipFree(q);
        b->d_Wx = b->d_Memx = nullptr;
        b->d_ids_x = nullptr;
        b->x_cap = 0;
        HIPCHK(hipMalloc((void **)&b->d_Wx, std::max<size_t>(need, 16)));
        HIPCHK(hipMalloc((void **)&b->d_Memx, std::max<size_t>(16, (size_t)p.mem_cells * 2 * lanes * sizeof(uint4))));
        std::vector<uint32_t> ident(lanes);
        for (uint32_t t = 0; t < lanes; t++) ident[t] = t;
        if (int rc = upload(&b->d_ids_x, ident)) return rc;
        b->x_cap = lanes;
    }
    if (own_scratch && b->x_cap > b->x_scratch_lanes) {
        b->x_scratch_lanes = 0;
        for (int k = 0; k < (int)N_CLS; k++) {
            if (b->d_x_scratch[k]) hipFree(b->d_x_scratch[k]);
            b->d_x_scratch[k] = nullptr;
            if (b->layout.cls_exact_words[k]) HIPCHK(hipMalloc((void **)&b->d_x_scratch[k], (size_t)b->layout.cls_exact_words[k] * b->x_cap * 4));
        }
        b->x_scratch_lanes = b->x_cap;
    }
    return 0;
}

// ACVM::solve for the batch. next_inputs (acvm_batch_solve_then_import): the device buffer of the NEXT tile's initial witnesses, whose import is
// enqueued right behind this solve's event count, gated ON THE DEVICE by that count: it runs only if no instance left the generic path (the
// exact path still needs this tile's rows otherwise). The host then waits for the count alone -- not for the import -- so the next tile's
// level kernels are enqueued while the import runs, and the device does not idle across the tile boundary (0.38 ms of a 23.6 ms tile of the
// metric's workload in round 3: event count, read-back, the caller's loop, import, its synchronisation).

// ---- asynchronous exact path (batch.hpp)
// exact lanes whose side table a handle allocates up front: what a tile of a few diverging inputs needs, bounded by 1 GiB
uint32_t async_exact_first_lanes(const Plan &p, uint32_t capacity) {
    const uint64_t by_bytes = (1ull << 30) / std::max<uint64_t>(64, (uint64_t)(p.n_witnesses + p.mem_cells) * 32);
    const uint64_t lanes = std::min<uint64_t>(std::min<uint64_t>(1024, std::max<uint64_t>(capacity / 8, 64)), std::max<uint64_t>(by_bytes, 64));
    return (uint32_t)(lanes / 64 * 64);
}
int batch_enable_async_exact(acvm_batch *b, const uint32_t *keep, uint32_t n_keep, bool digests) {
    const Plan &p = b->plan();
    if (b->has_solver || p.has_foreign_calls || p.truncated_at != 0xFFFFFFFFu || !p.tune.exact_async) return 0;
    if (!b->stream_x) HIPCHK(hipStreamCreateWithFlags(&b->stream_x, hipStreamNonBlocking));
    if (!b->ev_x_ready) HIPCHK(hipEventCreateWithFlags(&b->ev_x_ready, hipEventDisableTiming));
    b->async_exact = true;
    b->async_keep.assign(keep, keep + n_keep);
    for (uint32_t &w : b->async_keep)
        if (w >= p.n_witnesses) w = 0xFFFFFFFFu;  // (no row: exports as unassigned)
    b->async_digest = digests;
    // the side table of the first lanes now: a device without room for it says so at creation, not in the middle of a run (a job with more
    // lanes grows it, and falls back to the in-place path when it cannot)
    if (int rc = ensure_side_table(b, async_exact_first_lanes(p, b->capacity), true)) return rc;
    return 1;
}
size_t batch_device_bytes(const Plan &p, const PlanOpts &opts, uint64_t instances, bool async_exact) {
    const uint64_t Bp = (std::max<uint64_t>(instances, 1) + 63) / 64 * 64;
    const uint64_t rows = (opts.reuse_slots ? p.n_slots : p.n_witnesses) + (uint64_t)p.mem_cells + p.n_inverse_slots + p.n_digest_segments;
    size_t bytes = (size_t)rows * 2 * Bp * sizeof(uint4) + (size_t)Bp * 8;
    const uint64_t scratch_cap_words = std::max<uint64_t>(1, (1ull << 30) / (Bp * 4));
    for (int k = 0; k < (int)N_CLS; k++) {  // class scratch: the fattest level of the class (batch_init chunks a level at scratch_cap_words), or its fattest record
        uint64_t need = 0;
        for (size_t L = 0; L + 1 < p.cls_level_start[k].size(); L++) {
            uint64_t used = 0;
            for (uint32_t r = p.cls_level_start[k][L]; r < p.cls_level_start[k][L + 1]; r++) used += p.cls_scratch[k][r];
            need = std::max(need, std::min(used, std::max<uint64_t>(scratch_cap_words, 1)));
        }
        for (uint32_t oi = 0; oi < p.n_opcodes; oi++)
            if (p.prog_class[oi] == (uint32_t)k) need = std::max<uint64_t>(need, p.prog_scratch[oi]);
        bytes += (size_t)need * Bp * 4;
    }
    if (async_exact) bytes += (size_t)async_exact_first_lanes(p, (uint32_t)std::min<uint64_t>(instances, 0xFFFFFFFFu)) * (p.n_witnesses + (uint64_t)p.mem_cells) * 32;
    return bytes;
}

void batch_take_outcome(acvm_batch *b, ExactOutcome *out) {
    *out = std::move(b->last_outcome);
    b->last_outcome.clear();
}
const std::vector<uint32_t> *batch_exact_instances(const acvm_batch *b) { return &b->slow_ids; }
bool batch_exact_pending(const acvm_batch *b) { return b->pending; }
uint32_t batch_exact_unsolved(const acvm_batch *b, uint32_t n) {
    if (b->pending) return 0;  // (their outcome is not known yet)
    uint32_t bad = 0;
    for (size_t t = 0; t < b->slow_ids.size() && t < b->slow_res.size(); t++) bad += b->slow_ids[t] < n && b->slow_res[t].status != ACVM_STATUS_SOLVED;
    return bad;
}
bool batch_generic_assigned(const acvm_batch *b, uint32_t w) { return w < b->plan().n_witnesses && b->plan().producer[w] != 0xFFFFFFFFu; }
int batch_enqueue_kept(acvm_batch *b, uint32_t n, const uint32_t *d_keep, uint32_t n_keep, uint8_t *d_out, uint8_t *h_out, hipStream_t copy_stream,
                       hipEvent_t exported, hipEvent_t arrived) {
    HIPCHK(hipSetDevice(b->device));
    launch_export(b->stream, b->d_W, b->Bp, 0, n, d_keep, n_keep, d_out, b->unscale, b->d_slot_of);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(exported, b->stream));
    HIPCHK(hipStreamWaitEvent(copy_stream, exported, 0));
    HIPCHK(hipMemcpyAsync(h_out, d_out, (size_t)n * n_keep * 32, hipMemcpyDeviceToHost, copy_stream));
    HIPCHK(hipEventRecord(arrived, copy_stream));
    return 0;
}

// digests of lanes [first, first + n) of a witness table into host memory out32 ([n][32]), staged through the arena on stream s:
// arena = (slow_index) | partial sums | digests. The per-instance lane of `assigned` comes from a device array (d_slow_index) or from the
// batch's host vector (use_host_index: uploaded here); neither is needed when u.event is null (every lane read as an instance of the
// level kernels).

int side_table_outcome(acvm_batch *b, ExactOutcome *out) {
    const Plan &p = b->plan();
    hipStream_t s = b->xstream();
    const uint32_t n_slow = (uint32_t)b->slow_ids.size(), n_keep = (uint32_t)b->async_keep.size();
    out->instance = b->slow_ids;
    out->results.resize(n_slow);
    for (uint32_t t = 0; t < n_slow; t++) {
        acvm_result_t &r = out->results[t];
        memset(&r, 0, sizeof r);
        const SlowResult &sr = b->slow_res[t];
        r.status = sr.status; r.err = sr.err; r.opcode_index = sr.opcode_index; r.aux0 = sr.aux0; r.aux1 = sr.aux1;
        r.n_call_stack = sr.n_call_stack > 16 ? 16 : sr.n_call_stack;
        for (uint32_t k = 0; k < r.n_call_stack; k++) r.call_stack[k] = sr.call_stack[k];
    }
    const size_t sel_bytes = align256((size_t)std::max<uint32_t>(n_keep, 1) * 4), val_bytes = align256((size_t)n_slow * std::max<uint32_t>(n_keep, 1) * 32);
    if (int rc = stage_reserve(b, sel_bytes + val_bytes)) return rc;
    Unscale plain = b->unscale;
    plain.event = b->d_slow_start;  // (opcode indices, never 0xFFFFFFFF = "solved by the level kernels": nothing in the side table is scaled)
    if (n_keep) {
        uint32_t *d_sel = (uint32_t *)b->d_stage;
        uint8_t *d_val = b->d_stage + sel_bytes;
        HIPCHK(hipMemcpyAsync(d_sel, b->async_keep.data(), (size_t)n_keep * 4, hipMemcpyHostToDevice, s));
        launch_export(s, b->d_Wx, b->x_cap, 0, n_slow, d_sel, n_keep, d_val, plain);
        out->kept_values.resize((size_t)n_slow * n_keep * 32);
        HIPCHK(hipMemcpyAsync(out->kept_values.data(), d_val, out->kept_values.size(), hipMemcpyDeviceToHost, s));
        std::vector<uint32_t> bitmap((size_t)n_slow * b->n_words);
        HIPCHK(hipMemcpyAsync(bitmap.data(), b->d_assigned, bitmap.size() * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        out->kept_assigned.resize((size_t)n_slow * n_keep);
        for (uint32_t t = 0; t < n_slow; t++)
            for (uint32_t k = 0; k < n_keep; k++) {
                const uint32_t w = b->async_keep[k];
                const bool a = w < p.n_witnesses && ((bitmap[(size_t)(w >> 5) * n_slow + t] >> (w & 31)) & 1u);
                out->kept_assigned[(size_t)t * n_keep + k] = a;
                if (!a) memset(&out->kept_values[((size_t)t * n_keep + k) * 32], 0, 32);
            }
    }
    if (b->async_digest) {
        out->digests.resize((size_t)n_slow * 32);
        if (int rc = digest_range(b, s, b->d_Wx, b->x_cap, 0, n_slow, plain, (const int32_t *)b->d_ids_x, false, n_slow, out->digests.data())) return rc;
    }
    return 0;
}

int batch_set_live_count(acvm_batch *b, uint32_t n) {
    if (!b || !n || n > b->capacity) return set_err(ACVM_E_INVALID, "live count out of range");
    if (n != b->B) {
        b->B = n;
        b->inputs_set = false;
        b->solved = false;
        b->stepping = false;
    }
    return 0;
}
int batch_finish_pending(acvm_batch *b, ExactOutcome *out) {
    if (out) out->clear();
    if (!b->pending) return 0;
    HIPCHK(hipSetDevice(b->device));
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    b->slow_res.resize(n_slow);
    HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, b->stream_x));
    HIPCHK(hipStreamSynchronize(b->stream_x));
    int rc = retry_device_limits(b, n_slow, true, 0xFFFFFFFFu);
    if (!rc && out) {
        rc = side_table_outcome(b, out);
        for (uint32_t t = 0; t < n_slow && !rc; t++)  // message texts
            if (b->slow_res[t].status == ACVM_STATUS_FAILURE && b->slow_res[t].msg) format_message(b, b->slow_ids[t], b->slow_res[t], out->results[t]);
    }
    b->pending = false;
    return rc;
}


// ---- ACVM::get_pending_foreign_call / resolve_pending_foreign_call (pwg/mod.rs:203-228) per instance
static int fetch_pending(acvm_batch *b) {
    if (b->pend_host_valid) return 0;
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    b->h_pend_desc.assign((size_t)b->fc_pend_desc_words * n_slow, 0);
    b->h_pend_vals.assign((size_t)b->fc_pend_vals_cap * 2 * n_slow * 4, 0);
    if (n_slow && b->d_fc_pend_desc) {
        HIPCHK(hipMemcpy(b->h_pend_desc.data(), b->d_fc_pend_desc, b->h_pend_desc.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b->h_pend_vals.data(), b->d_fc_pend_vals, b->h_pend_vals.size() * 4, hipMemcpyDeviceToHost));
    }
    b->pend_host_valid = true;
    return 0;
}
static int waiting_lane(acvm_batch *b, uint32_t instance) {
    if (!b->solved || instance >= b->B) return -1;
    int32_t t = b->slow_index[instance];
    if (t < 0 || b->slow_res[t].status != ACVM_STATUS_REQUIRES_FOREIGN_CALL) return -1;
    return t;
}

int acvm_batch_pending_foreign_call(acvm_batch_t *b, uint32_t instance, acvm_foreign_call_info_t *info) try {
    if (!b || !info) return set_err(ACVM_E_INVALID, "null argument");
    memset(info, 0, sizeof *info);
    int t = waiting_lane(b, instance);
    if (t < 0) return 0;
    HIPCHK(hipSetDevice(b->device));
    if (int rc = fetch_pending(b)) return rc;
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    const SlowResult &sr = b->slow_res[t];
    info->opcode_index = sr.opcode_index;
    info->brillig_index = sr.x0;
    info->n_inputs = b->h_pend_desc[t];
    for (uint32_t i = 0; i < info->n_inputs; i++) info->n_values += b->h_pend_desc[(size_t)(1 + i) * n_slow + t];
    auto it = b->plan().fc_function.find(((uint64_t)sr.opcode_index << 32) | sr.x0);
    snprintf(info->function, sizeof info->function, "%s", it == b->plan().fc_function.end() ? "" : it->second.c_str());
    return 1;
} ABI_CATCH

int acvm_batch_pending_foreign_call_inputs(acvm_batch_t *b, uint32_t instance, uint32_t *lens, uint8_t *values_be32) try {
    if (!b || !lens || !values_be32) return set_err(ACVM_E_INVALID, "null argument");
    int t = waiting_lane(b, instance);
    if (t < 0) return set_err(ACVM_E_STATE, "instance is not waiting for a foreign call");
    HIPCHK(hipSetDevice(b->device));
    if (int rc = fetch_pending(b)) return rc;
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    const uint32_t n_in = b->h_pend_desc[t];
    uint32_t vi = 0;
    for (uint32_t i = 0; i < n_in; i++) {
        lens[i] = b->h_pend_desc[(size_t)(1 + i) * n_slow + t];
        for (uint32_t c = 0; c < lens[i]; c++, vi++) {
            FrH m;
            memcpy(&m.l[0], &b->h_pend_vals[((size_t)(2 * vi) * n_slow + t) * 4], 16);
            memcpy(&m.l[2], &b->h_pend_vals[((size_t)(2 * vi + 1) * n_slow + t) * 4], 16);
            uint64_t can[4];
            frh::to_canonical(frh::from_device_form(m), can);
            for (int k = 0; k < 32; k++) values_be32[(size_t)vi * 32 + 31 - k] = (uint8_t)(can[k / 8] >> (8 * (k % 8)));
        }
    }
    return 0;
} ABI_CATCH

// ---- the caller's BlackBoxFunctionSolver inside Brillig programs (brillig_vm/src/lib.rs:61,81,298; black_box.rs:139-163). plan.cpp turned
// BlackBoxOp::{SchnorrVerify, Pedersen, FixedBaseScalarMul} into internal foreign calls; the lanes that stopped at one are answered here:
// all waiting instances are read in ONE pass over the pending-call buffers, grouped by call shape, handed to the solver (one *_batch call
// per group when the vtable has the member, the single-instance callback otherwise) and their results appended to the result store like a
// resolved oracle call. A callback that does not return Ok leaves a failure marker: the re-run VM fails (or panics) at the op with the
// solver's text. Returns the number of lanes answered (the caller re-solves when it is not 0), or a negative error.
int resolve_internal_calls(acvm_batch *b) {
    if (!b->has_solver || !b->plan().has_foreign_calls || !b->solved) return 0;
    const Plan &p = b->plan();
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    struct Waiting { uint32_t t, kind; size_t slot; };
    std::vector<Waiting> lanes;
    for (uint32_t t = 0; t < n_slow && t < b->fc_lane.size(); t++) {
        const SlowResult &sr = b->slow_res[t];
        if (sr.status != ACVM_STATUS_REQUIRES_FOREIGN_CALL || b->fc_lane[t].resolved_new) continue;
        auto it = p.fc_function.find(((uint64_t)sr.opcode_index << 32) | sr.x0);
        if (it == p.fc_function.end() || it->second.compare(0, strlen(PLAN_FC_INTERNAL_PREFIX), PLAN_FC_INTERNAL_PREFIX) != 0) continue;
        const uint32_t kind = it->second == PLAN_FC_INTERNAL_SCHNORR ? 6u : it->second == PLAN_FC_INTERNAL_PEDERSEN ? 7u : 8u;
        const size_t si = (size_t)(std::find(p.fc_slot_opcode.begin(), p.fc_slot_opcode.end(), sr.opcode_index) - p.fc_slot_opcode.begin());
        if (si >= b->fc_slots.size()) return set_err(ACVM_E_STATE, "internal black-box call without a result slot");
        lanes.push_back({t, kind, si});
    }
    if (lanes.empty()) return 0;
    HIPCHK(hipSetDevice(b->device));
    if (int rc = fetch_pending(b)) return rc;
    const acvm_bb_solver_t &sv = b->solver;
    static constexpr size_t ERR_STRIDE = 200;
    // canonical big-endian bytes of value number vi of lane t
    auto value_be = [&](uint32_t t, uint32_t vi, uint8_t *out) {
        FrH m;
        memcpy(&m.l[0], &b->h_pend_vals[((size_t)(2 * vi) * n_slow + t) * 4], 16);
        memcpy(&m.l[2], &b->h_pend_vals[((size_t)(2 * vi + 1) * n_slow + t) * 4], 16);
        uint64_t can[4];
        frh::to_canonical(frh::from_device_form(m), can);
        for (int k = 0; k < 32; k++) out[31 - k] = (uint8_t)(can[k / 8] >> (8 * (k % 8)));
    };
    auto input_len = [&](uint32_t t, uint32_t i) { return b->h_pend_desc[(size_t)(1 + i) * n_slow + t]; };
    struct Outcome { int rc = 0; uint8_t out[64] = {0}; std::string err; };
    std::vector<Outcome> outcome(lanes.size());
    // groups of one call shape: kind + the sizes the batched members take as one number (+ Pedersen's domain separator)
    std::map<std::vector<uint64_t>, std::vector<size_t>> groups;
    for (size_t q = 0; q < lanes.size(); q++) {
        const uint32_t t = lanes[q].t, kind = lanes[q].kind;
        std::vector<uint64_t> key = {kind};
        if (kind == 6) { key.push_back(input_len(t, 2)); key.push_back(input_len(t, 3)); }  // message, signature
        else if (kind == 7) {
            const uint32_t n = input_len(t, 0);
            uint8_t dom[32];
            value_be(t, n, dom);
            // registers.get(domain_separator).to_u128().try_into::<u32>() (brillig_vm/src/black_box.rs:152-158): to_u128 keeps the LOW 128 bits
            // (acir_field generic_ark.rs:227-230), so only bits 32..127 can make the conversion fail -- a value of 2^128 + k passes with separator k
            // (ops_brillig.hpp tests the same three words)
            bool fits = true;
            for (int k = 16; k < 28; k++) fits &= dom[k] == 0;
            if (!fits) { outcome[q].rc = 1; outcome[q].err = "Invalid signature length"; continue; }
            key.push_back(n);
            key.push_back((uint64_t)dom[28] << 24 | (uint64_t)dom[29] << 16 | (uint64_t)dom[30] << 8 | dom[31]);
        }
        groups[key].push_back(q);
    }
    for (auto &g : groups) {
        const uint32_t kind = (uint32_t)g.first[0];
        const size_t n = g.second.size();
        std::vector<uint8_t> brc(n, 0), bout(n * 64, 0);
        std::vector<char> berr(n * ERR_STRIDE, 0);
        int call_rc = 0;
        bool batched = false;
        if (kind == 8) {  // FixedBaseScalarMul: inputs [low, high]
            std::vector<uint8_t> lh(n * 64);
            for (size_t i = 0; i < n; i++) { value_be(lanes[g.second[i]].t, 0, &lh[i * 64]); value_be(lanes[g.second[i]].t, 1, &lh[i * 64 + 32]); }
            if ((batched = sv.fixed_base_scalar_mul_batch != nullptr)) call_rc = sv.fixed_base_scalar_mul_batch(sv.ctx, n, lh.data(), bout.data(), brc.data(), berr.data(), ERR_STRIDE);
            else
                for (size_t i = 0; i < n; i++) brc[i] = (uint8_t)std::min(std::max(sv.fixed_base_scalar_mul(sv.ctx, &lh[i * 64], &lh[i * 64 + 32], &bout[i * 64], &bout[i * 64 + 32], &berr[i * ERR_STRIDE], ERR_STRIDE), 0), 255);
        } else if (kind == 7) {  // Pedersen: inputs [vector of k field elements, domain separator]
            const size_t k = (size_t)g.first[1];
            const uint32_t dom = (uint32_t)g.first[2];
            std::vector<uint8_t> pin(n * std::max<size_t>(k, 1) * 32);
            for (size_t i = 0; i < n; i++)
                for (size_t c = 0; c < k; c++) value_be(lanes[g.second[i]].t, (uint32_t)c, &pin[(i * k + c) * 32]);
            if ((batched = sv.pedersen_batch != nullptr)) call_rc = sv.pedersen_batch(sv.ctx, n, pin.data(), k, dom, bout.data(), brc.data(), berr.data(), ERR_STRIDE);
            else
                for (size_t i = 0; i < n; i++) brc[i] = (uint8_t)std::min(std::max(sv.pedersen(sv.ctx, &pin[i * k * 32], k, dom, &bout[i * 64], &bout[i * 64 + 32], &berr[i * ERR_STRIDE], ERR_STRIDE), 0), 255);
        } else {  // SchnorrVerify: inputs [pk x, pk y, message bytes, signature bytes]; to_u8_vec keeps the last byte of every value (black_box.rs:27-36)
            const uint32_t n_msg = (uint32_t)g.first[1], n_sig = (uint32_t)g.first[2];
            std::vector<uint8_t> pk(n * 64), sig(n * std::max<uint32_t>(n_sig, 1)), msg(n * std::max<uint32_t>(n_msg, 1)), ok(n, 0);
            uint8_t tmp[32];
            for (size_t i = 0; i < n; i++) {
                const uint32_t t = lanes[g.second[i]].t;
                value_be(t, 0, &pk[i * 64]);
                value_be(t, 1, &pk[i * 64 + 32]);
                for (uint32_t c = 0; c < n_msg; c++) { value_be(t, 2 + c, tmp); msg[i * n_msg + c] = tmp[31]; }
                for (uint32_t c = 0; c < n_sig; c++) { value_be(t, 2 + n_msg + c, tmp); sig[i * n_sig + c] = tmp[31]; }
            }
            if ((batched = sv.schnorr_verify_batch != nullptr)) call_rc = sv.schnorr_verify_batch(sv.ctx, n, pk.data(), sig.data(), n_sig, msg.data(), n_msg, ok.data(), brc.data(), berr.data(), ERR_STRIDE);
            else
                for (size_t i = 0; i < n; i++) brc[i] = (uint8_t)std::min(std::max(sv.schnorr_verify(sv.ctx, &pk[i * 64], &pk[i * 64 + 32], &sig[i * n_sig], n_sig, &msg[i * n_msg], n_msg, &ok[i], &berr[i * ERR_STRIDE], ERR_STRIDE), 0), 255);
            for (size_t i = 0; i < n; i++) bout[i * 64 + 31] = ok[i] ? 1 : 0;
        }
        for (size_t i = 0; i < n; i++) {
            Outcome &o = outcome[g.second[i]];
            o.rc = batched && call_rc != 0 ? 3 : brc[i];
            memcpy(o.out, &bout[i * 64], 64);
            berr[i * ERR_STRIDE + ERR_STRIDE - 1] = 0;
            o.err = batched && call_rc != 0 ? "batched BlackBoxFunctionSolver call failed" : &berr[i * ERR_STRIDE];
        }
    }
    static const char *names[3] = {"schnorr_verify", "pedersen", "fixed_base_scalar_mul"};
    for (size_t q = 0; q < lanes.size(); q++) {
        const Waiting &w = lanes[q];
        const Outcome &o = outcome[q];
        const uint32_t instance = b->slow_ids[w.t];
        std::vector<acvm_batch::FcValue> res(1);
        if (o.rc != 0) {
            res[0].is_array = false;
            res[0].fail = o.rc == 1 ? 1u : o.rc == 2 ? 2u : 3u;
            // the Display strings of BlackBoxResolutionError (blackbox_solver/src/lib.rs:15-21), a panicking solver's own text
            const char *fn = names[w.kind - 6];
            b->fc_fail_msg[instance] = o.rc == 1 ? std::string("failed to solve blackbox function: ") + fn + ", reason: " + o.err
                                       : o.rc == 2 ? std::string("unsupported blackbox function: ") + fn : o.err;
        } else if (w.kind == 6) {
            res[0].is_array = false;
            res[0].vals = {frh::from_u64(o.out[31] ? 1 : 0)};
        } else {
            res[0].is_array = true;
            res[0].vals = {frh::from_be_bytes32_reduce(o.out, 32), frh::from_be_bytes32_reduce(o.out + 32, 32)};
        }
        b->fc_slots[w.slot].inst[instance].push_back(std::move(res));
        b->fc_slots[w.slot].dirty = true;
        b->fc_lane[w.t].resolved_new = true;
    }
    return (int)lanes.size();
}

int acvm_batch_resolve_foreign_call(acvm_batch_t *b, uint32_t instance, uint32_t n_values, const uint8_t *is_array, const uint32_t *lens,
                                    const uint8_t *values_be32) try {
    if (!b || (n_values && (!is_array || !lens || !values_be32))) return set_err(ACVM_E_INVALID, "null argument");
    int t = waiting_lane(b, instance);
    if (t < 0) return set_err(ACVM_E_STATE, "ACVM is not expecting a foreign call response as no call was made");  // mod.rs:215-217 panics
    auto &ls = b->fc_lane[t];
    const uint32_t opcode = b->slow_res[t].opcode_index;
    if (ls.resolved_new) return set_err(ACVM_E_STATE, "this instance's pending foreign call was already resolved; call acvm_batch_solve");
    const auto &slots = b->plan().fc_slot_opcode;
    const size_t si = (size_t)(std::find(slots.begin(), slots.end(), opcode) - slots.begin());
    if (si >= b->fc_slots.size()) return set_err(ACVM_E_STATE, "the instance does not wait at a Brillig opcode with a foreign call");
    std::vector<acvm_batch::FcValue> res(n_values);
    size_t off = 0;
    for (uint32_t i = 0; i < n_values; i++) {
        res[i].is_array = is_array[i] != 0;
        uint32_t n = res[i].is_array ? lens[i] : 1;
        for (uint32_t c = 0; c < n; c++, off++) res[i].vals.push_back(frh::from_be_bytes32_reduce(values_be32 + off * 32, 32));
    }
    // results accumulate per Brillig opcode and instance (brillig.foreign_call_results.push, mod.rs:223)
    b->fc_slots[si].inst[instance].push_back(std::move(res));
    b->fc_slots[si].dirty = true;
    ls.resolved_new = true;
    return 0;
} ABI_CATCH


