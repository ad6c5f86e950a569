// batch.cpp -- the batch handle behind the C ABI (include/acvm_amd.h): creation (plan + device tables), the initial witness, the small entry
// points of devices / tuning / circuits, statistics. One handle = one circuit plan, one device-resident witness table W[slot][half][instance],
// one stream set. Mirrors the call shape of acvm::pwg::ACVM (acvm/src/pwg/mod.rs:145-304) for B instances at once. The solve lives in
// batch_schedule.cpp (level schedule) and batch_exact.cpp (exact in-order path, foreign calls, stepping), what leaves the device in
// batch_export.cpp, the measurement probes in probes.cpp.
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

static thread_local std::string g_last_error;
int set_err(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int stage_reserve(acvm_batch *b, size_t bytes) {
    if (bytes <= b->stage_cap) return 0;
    if (b->d_stage) { hipFree(b->d_stage); b->d_stage = nullptr; b->stage_cap = 0; }
    const size_t cap = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
    HIPCHK(hipMalloc((void **)&b->d_stage, cap));
    b->stage_cap = cap;
    return 0;
}
void clear_fc_store(acvm_batch *b) {
    for (auto &sl : b->fc_slots)
        if (!sl.inst.empty()) { sl.inst.clear(); sl.dirty = true; }
    b->fc_fail_msg.clear();
}

const char *acvm_last_error(void) { return g_last_error.c_str(); }
int acvm_abi_version(void) { return ACVM_AMD_ABI_VERSION; }

int acvm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int acvm_set_device(int device) {
    HIPCHK(hipSetDevice(device));
    return 0;
}
int acvm_device_synchronize(void) {
    HIPCHK(hipDeviceSynchronize());
    return 0;
}
int acvm_device_arch(char *out, size_t out_len) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    snprintf(out, out_len, "%s", prop.gcnArchName);
    return 0;
}

void *acvm_device_malloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { set_err(ACVM_E_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
int acvm_device_free(void *p) {
    if (p) HIPCHK(hipFree(p));
    return 0;
}
int acvm_device_upload(void *dst_device, const void *src_host, size_t bytes) {
    if (bytes && (!dst_device || !src_host)) return set_err(ACVM_E_INVALID, "null argument");
    if (bytes) HIPCHK(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return 0;
}

long long acvm_device_release_tables(int device) {
    if (device < 0 || device >= acvm_device_count()) return set_err(ACVM_E_INVALID, "device index out of range");
    size_t freed = 0;
    const int rc = device_tables_free(device, &freed);
    if (rc == 1) return set_err(ACVM_E_STATE, "batch handles of device " + std::to_string(device) + " still use its lookup tables: free them first");
    if (rc < 0) return set_err(ACVM_E_DEVICE, "could not select device " + std::to_string(device));
    return (long long)freed;
}

int acvm_tuning_set(const char *key, long long value) {
    if (!tuning_set(key, (int64_t)value)) return set_err(ACVM_E_INVALID, std::string("unknown tuning key (or a value it does not take): ") + (key ? key : "(null)"));
    return 0;
}
int acvm_tuning_get(const char *key, long long *value) {
    int64_t v = 0;
    if (!value || !tuning_get(key, &v)) return set_err(ACVM_E_INVALID, std::string("unknown tuning key ") + (key ? key : "(null)"));
    *value = (long long)v;
    return 0;
}
const char *acvm_tuning_key(unsigned index) { return tuning_key(index); }


acvm_circuit_t *acvm_circuit_from_bytes(const uint8_t *bytes, size_t len) try {
    if (!bytes) { set_err(ACVM_E_INVALID, "null circuit bytes"); return nullptr; }
    if (!frh::self_check()) { set_err(ACVM_E_INVALID, "field constants self-check failed"); return nullptr; }
    std::string err;
    auto c = circuit_from_bytes(bytes, len, err);
    if (!c) { set_err(ACVM_E_MALFORMED, err); return nullptr; }
    auto *h = new acvm_circuit;
    h->c = std::move(c);
    return h;
} ABI_CATCH_PTR
void acvm_circuit_free(acvm_circuit_t *c) { delete c; }
uint32_t acvm_circuit_num_opcodes(const acvm_circuit_t *c) { return c ? (uint32_t)c->c->opcodes.size() : 0; }
uint32_t acvm_circuit_num_witnesses(const acvm_circuit_t *c) { return c ? c->c->max_witness + 1 : 0; }
int acvm_circuit_opcode_kinds(const acvm_circuit_t *c, uint32_t first, uint32_t n, uint32_t *kinds) {
    if (!c || (n && !kinds)) return set_err(ACVM_E_INVALID, "null argument");
    const std::vector<Opcode> &ops = c->c->opcodes;
    if ((uint64_t)first + n > ops.size()) return set_err(ACVM_E_INVALID, "opcode range out of bounds");
    for (uint32_t i = 0; i < n; i++) {
        const Opcode &o = ops[first + i];
        kinds[2 * i] = o.kind;
        kinds[2 * i + 1] = o.kind == OP_BLACKBOX ? o.bb->func : o.kind == OP_DIRECTIVE ? o.dir->kind : o.kind == OP_BRILLIG ? (uint32_t)o.brillig->bytecode.size()
                           : o.kind == OP_MEMORY_OP || o.kind == OP_MEMORY_INIT ? o.block_id : 0u;
    }
    return 0;
}

void plan_stats(const Plan &p, acvm_stats_t *out) {
    memset(out, 0, sizeof *out);
    out->n_opcodes = p.n_opcodes;
    out->n_witnesses = p.n_witnesses;
    out->n_levels = p.n_levels;
    out->n_fast_gates = p.n_fast_gates;
    out->n_dyn_gates = p.n_dyn_gates;
    out->max_level_width = p.max_level_width;
    out->algorithmic_bytes_per_instance = p.algorithmic_bytes;
    out->arith_algorithmic_bytes_per_instance = p.arith_algorithmic_bytes;
    out->dyn_algorithmic_bytes_per_instance = p.dyn_algorithmic_bytes;
    out->plan_ms = p.plan_ms;
    out->n_other_records = p.n_other_records;
    out->truncated_at = p.truncated_at;
    out->n_gate_pairs = p.n_gate_pairs;
    out->n_inverse_slots = p.n_inverse_slots;
    out->n_scaled_witnesses = (uint32_t)p.scaled_ids.size();
    out->n_table_rows = p.slot_of.empty() ? p.n_witnesses : p.n_slots;
    out->n_digest_segments = p.n_digest_segments;
    out->n_brillig_inlined = p.n_brillig_inlined;
    out->n_hash_chained = p.n_hash_chained;
    out->n_gate_out_asis = p.n_gate_out_mode[0];
    out->n_gate_out_weak = p.n_gate_out_mode[1];
    out->n_gate_out_canon = p.n_gate_out_mode[2];
    out->max_gate_bound = p.max_gate_bound;
    out->n_byte_planes = p.n_byte_planes;
    out->n_byte_plane_reads = p.n_byte_plane_reads;
    for (uint32_t L = 0; L + 1 < p.level_start.size(); L++) out->n_arith_launches += (p.level_start[L + 1] - p.level_start[L] + 65534) / 65535;
    for (int k = 0; k < 4; k++) out->class_algorithmic_bytes_per_instance[k] = p.cls_algorithmic_bytes[k];
    out->class_algorithmic_bytes_per_instance[CLS_GRUMPKIN] += p.cls_algorithmic_bytes[CLS_PEDERSEN] + p.cls_algorithmic_bytes[CLS_ECDSA] +
                                                               p.cls_algorithmic_bytes[CLS_HOSTBB];
}

// Host-only: levelise the circuit against a set of initial witness ids without touching a device (plan statistics, and
// whether the circuit holds an opcode no kernel implements). Returns 0, or ACVM_E_UNSUPPORTED with the reason as the error text.
int acvm_circuit_plan_stats(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, acvm_stats_t *out) {
    return acvm_circuit_plan_stats_ex(c, initial_ids, n_initial, 0, nullptr, 0, out);
}
int acvm_circuit_plan_stats_ex(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids,
                               uint32_t n_keep, acvm_stats_t *out) try {
    if (!c || !out || (n_initial && !initial_ids) || (n_keep && !keep_ids)) return set_err(ACVM_E_INVALID, "null argument");
    PlanOpts opts;
    opts.fold_digest = (flags & (ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) != 0;
    opts.reuse_slots = (flags & ACVM_BATCH_REUSE_SLOTS) != 0;
    opts.keep.assign(keep_ids, keep_ids + n_keep);
    const std::shared_ptr<const Plan> sp = plan_for(c, initial_ids, n_initial, opts);
    const Plan &p = *sp;
    plan_stats(p, out);
    if (!p.unsupported.empty()) return set_err(ACVM_E_UNSUPPORTED, p.unsupported);
    return 0;
} ABI_CATCH

// The circuit's plan cache (batch.hpp PlanKey): a hit hands out the shared plan; a miss plans outside the lock (a second thread asking for the
// same plan meanwhile plans too -- the node driver asks once, before its lanes start).
static size_t plan_bytes(const Plan &p) {
    size_t words = p.gate_stream.size() + p.gate_offset.size() + p.prog.size() + p.prog_offset.size() + p.prog_scratch.size() + p.bytecode.size() + p.slot_of.size() + p.kbound.size() +
                   p.unscale_index.size() + p.scaled_ids.size() + p.producer.size() + p.byte_plane_of.size() + p.dyn_offset.size();
    for (int k = 0; k < (int)N_CLS; k++) words += 2 * p.cls_offset[k].size();
    return words * 4 + (p.constants.size() + p.unscale.size()) * sizeof(FrH);
}
// (under c->plan_mutex) sp becomes the newest of the strongly held plans; older ones go while there are more than 8 or more than ~1 GiB of them
static void remember_plan(const acvm_circuit *c, const std::shared_ptr<const Plan> &sp) {
    auto &v = c->recent_plans;
    v.erase(std::remove(v.begin(), v.end(), sp), v.end());
    v.push_back(sp);
    size_t total = 0;
    for (auto &q : v) total += plan_bytes(*q);
    while (v.size() > 1 && (v.size() > 8 || total > (1ull << 30))) { total -= plan_bytes(*v.front()); v.erase(v.begin()); }
}
std::shared_ptr<const Plan> plan_for(const acvm_circuit *c, const uint32_t *initial_ids, uint32_t n_initial, const PlanOpts &opts) {
    PlanKey key;
    key.ids.assign(initial_ids, initial_ids + n_initial);
    if (opts.reuse_slots) key.keep = opts.keep;  // (the kept witnesses only matter to the row assignment of slot reuse: plain handles with different lists share a plan)
    key.host_blackbox = opts.host_blackbox;
    key.fold_digest = opts.fold_digest;
    key.reuse_slots = opts.reuse_slots;
    for (unsigned i = 0; const char *name = tuning_key(i); i++) {
        int64_t v = 0;
        tuning_get(name, &v);
        key.tuning.push_back(v);
    }
    {
        std::lock_guard<std::mutex> g(c->plan_mutex);
        for (auto it = c->plan_cache.begin(); it != c->plan_cache.end();) {
            std::shared_ptr<const Plan> sp = it->second.lock();
            if (!sp) { it = c->plan_cache.erase(it); continue; }
            if (it->first == key) { c->n_plans_shared++; remember_plan(c, sp); return sp; }
            ++it;
        }
    }
    std::shared_ptr<const Plan> sp = std::make_shared<const Plan>(build_plan(*c->c, initial_ids, n_initial, opts));
    std::lock_guard<std::mutex> g(c->plan_mutex);
    c->n_plans_built++;
    c->plan_cache.push_back({std::move(key), sp});
    remember_plan(c, sp);
    return sp;
}
uint64_t acvm_circuit_plans_built(const acvm_circuit_t *c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> g(c->plan_mutex);
    return c->n_plans_built;
}

// Host-only: the hazard checker (schedule_check.cpp) over the level schedule a handle of n_instances instances and these options would
// enqueue. Returns 0 (proved), 1 (findings: counts[4] of them, the first 32 in `report`) or a negative error. counts: launches, waits,
// accesses, records, findings. drop_wait: see include/acvm_amd.h.
int acvm_circuit_check_schedule(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids, uint32_t n_keep,
                                uint32_t n_instances, uint32_t drop_wait, uint64_t *counts, char *report, size_t report_len) try {
    if (!c || (n_initial && !initial_ids) || (n_keep && !keep_ids)) return set_err(ACVM_E_INVALID, "null argument");
    PlanOpts opts;
    opts.fold_digest = (flags & (ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) != 0;
    opts.reuse_slots = (flags & ACVM_BATCH_REUSE_SLOTS) != 0;
    opts.host_blackbox = (flags & 0x100u) != 0;
    opts.keep.assign(keep_ids, keep_ids + n_keep);
    const std::shared_ptr<const Plan> sp = plan_for(c, initial_ids, n_initial, opts);
    if (!sp->unsupported.empty()) return set_err(ACVM_E_UNSUPPORTED, sp->unsupported);
    const LaunchLayout lay = layout_launches(*sp, ((uint64_t)n_instances + 63) / 64 * 64);
    const LevelSchedule sched = level_schedule(*sp, lay);
    const ScheduleReport rep = check_level_schedule(*sp, lay, sched, drop_wait);
    if (counts) { counts[0] = rep.n_launches; counts[1] = rep.n_waits; counts[2] = rep.n_accesses; counts[3] = rep.n_records; counts[4] = rep.n_findings; }
    if (report && report_len) snprintf(report, report_len, "%s", rep.text.c_str());
    return rep.ok ? 0 : 1;
} ABI_CATCH

// Host-only: 64-bit FNV-1a fingerprints of everything the planner hands to the device and to the scheduler, component by component
// (tests/test_plan_host.py, tools/plan_fingerprint.py: a refactoring of the planner must not move a word). out[0..N): see the order below;
// the per-class record lists are hashed level by level in sorted order (the order inside a level is a launch-placement choice).
int acvm_debug_plan_fingerprint(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids,
                                uint32_t n_keep, uint64_t *out, uint32_t cap) try {
    if (!c || !out || (n_initial && !initial_ids) || (n_keep && !keep_ids)) return set_err(ACVM_E_INVALID, "null argument");
    PlanOpts opts;
    opts.fold_digest = (flags & (ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) != 0;
    opts.reuse_slots = (flags & ACVM_BATCH_REUSE_SLOTS) != 0;
    opts.host_blackbox = (flags & 0x100u) != 0;
    opts.keep.assign(keep_ids, keep_ids + n_keep);
    const Plan p = build_plan(*c->c, initial_ids, n_initial, opts);
    if (!p.unsupported.empty()) return set_err(ACVM_E_UNSUPPORTED, p.unsupported);
    std::vector<uint64_t> fp;
    auto fnv = [](const void *data, size_t bytes, uint64_t h = 0xcbf29ce484222325ull) {
        const uint8_t *b = (const uint8_t *)data;
        for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
        return h;
    };
    auto vec = [&](const std::vector<uint32_t> &v) { fp.push_back(fnv(v.data(), v.size() * 4) ^ (uint64_t)v.size() << 40); };
    vec(p.gate_stream); vec(p.gate_offset); vec(p.level_start); vec(p.dyn_offset); vec(p.dyn_level_start); vec(p.level_needs_inverse);
    for (int q = 0; q < N_HEAVY_LANES; q++) { vec(p.level_needs_heavy[q]); vec(p.inv_needs_heavy[q]); vec(p.lane_needs_main[q]); for (int q2 = 0; q2 < N_HEAVY_LANES; q2++) vec(p.lane_needs_lane[q][q2]); }
    vec(p.prog); vec(p.prog_offset); vec(p.prog_scratch); vec(p.slot_of); vec(p.kbound); vec(p.scaled_ids); vec(p.unscale_index); vec(p.bytecode); vec(p.producer); vec(p.byte_plane_of);
    fp.push_back(fnv(p.constants.data(), p.constants.size() * sizeof(FrH)));
    fp.push_back(fnv(p.unscale.data(), p.unscale.size() * sizeof(FrH)));
    fp.push_back(fnv(p.prog_class.data(), p.prog_class.size()));
    for (int k = 0; k < (int)N_CLS; k++) {
        vec(p.cls_level_start[k]);
        std::vector<uint32_t> sorted;
        for (size_t L = 0; L + 1 < p.cls_level_start[k].size(); L++) {
            std::vector<std::pair<uint32_t, uint32_t>> lv;
            for (uint32_t r = p.cls_level_start[k][L]; r < p.cls_level_start[k][L + 1]; r++) lv.push_back({p.cls_offset[k][r], p.cls_scratch[k][r]});
            std::sort(lv.begin(), lv.end());
            for (auto &x : lv) { sorted.push_back(x.first); sorted.push_back(x.second); }
        }
        vec(sorted);
    }
    const uint32_t scal[] = {p.n_witnesses, p.n_opcodes, p.n_levels, p.n_slots, p.mem_cells, p.n_inverse_slots, p.n_digest_segments, p.truncated_at, p.n_byte_planes};
    fp.push_back(fnv(scal, sizeof scal));
    for (size_t i = 0; i < fp.size() && i < cap; i++) out[i] = fp[i];
    return (int)fp.size();
} ABI_CATCH

// The two fixed field elements of the witness-map digest (include/acvm_amd.h acvm_batch_digest): Blake2s-256 of the ASCII strings
// "acvm_amd witness map digest: g" / "... h", read as big-endian integers and reduced modulo p.

static int batch_init(acvm_batch *b) {
    HIPCHK(hipGetDevice(&b->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, b->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(ACVM_E_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_dyn, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_heavy, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_heavy2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_heavy3, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_digest, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&b->ev_start));
    HIPCHK(hipEventCreate(&b->ev_end));
    const Plan &p = b->plan();
    size_t w_bytes = (size_t)(b->reuse() ? p.n_slots : p.n_witnesses) * 2 * b->Bp * sizeof(uint4);
    HIPCHK(hipMalloc((void **)&b->d_W, w_bytes ? w_bytes : 16));
    if (int rc = upload(&b->d_gate_stream, p.gate_stream)) return rc;
    if (int rc = upload(&b->d_gate_offset, p.gate_offset)) return rc;
    std::vector<uint32_t> consts(p.constants.size() * 8);
    for (size_t i = 0; i < p.constants.size(); i++) {
        const FrH d = frh::to_device_form(p.constants[i]);
        memcpy(&consts[8 * i], d.l, 32);
    }
    if (int rc = upload(&b->d_consts, consts)) return rc;
    if (!p.scaled_ids.empty()) {
        std::vector<uint32_t> uc(p.unscale.size() * 8);
        for (size_t i = 0; i < p.unscale.size(); i++) {
            const FrH d = frh::to_device_form(p.unscale[i]);
            memcpy(&uc[8 * i], d.l, 32);
        }
        if (int rc = upload(&b->d_unscale_consts, uc)) return rc;
        for (size_t i = 0; i < p.unscale.size(); i++) {
            uint64_t c[4];
            frh::to_canonical(p.unscale[i], c);
            memcpy(&uc[8 * i], c, 32);
        }
        if (int rc = upload(&b->d_unscale_plain, uc)) return rc;
        if (int rc = upload(&b->d_unscale_index, p.unscale_index)) return rc;
        if (int rc = upload(&b->d_scaled_ids, p.scaled_ids)) return rc;
    }
    if (int rc = upload(&b->d_prog, p.prog)) return rc;
    if (int rc = upload(&b->d_prog_offset, p.prog_offset)) return rc;
    if (int rc = upload(&b->d_bytecode, p.bytecode)) return rc;
    if (int rc = upload(&b->d_init_ids, p.initial_ids)) return rc;
    if (p.n_byte_planes) {  // plan.hpp "Byte planes"
        std::vector<uint32_t> of_input(p.initial_ids.size());
        for (size_t i = 0; i < p.initial_ids.size(); i++) of_input[i] = p.byte_plane_of[p.initial_ids[i]];
        if (int rc = upload(&b->d_byte_plane_of, p.byte_plane_of)) return rc;
        if (int rc = upload(&b->d_byte_plane_of_input, of_input)) return rc;
        HIPCHK(hipMalloc((void **)&b->d_byte_plane, (size_t)p.n_byte_planes * b->Bp * 4));
        HIPCHK(hipMemsetAsync(b->d_byte_plane, 0, (size_t)p.n_byte_planes * b->Bp * 4, b->stream));
    }
    {   // per-instance memory blocks (MemoryInit / MemoryOp), laid out like W
        size_t bytes = (size_t)p.mem_cells * 2 * b->Bp * sizeof(uint4);
        HIPCHK(hipMalloc((void **)&b->d_Mem, bytes ? bytes : 16));
    }
    // non-arithmetic record classes: where their records are cut into launches (schedule.cpp), then the level schedule of one solve
    b->layout = layout_launches(p, b->Bp);
    b->schedule = level_schedule(p, b->layout);
    for (int k = 0; k < (int)N_CLS; k++) {
        if (int rc = upload(&b->d_cls_offset[k], p.cls_offset[k])) return rc;
        if (int rc = upload(&b->d_cls_scratch_off[k], b->layout.scratch_off[k])) return rc;
        if (b->layout.scratch_words[k]) HIPCHK(hipMalloc((void **)&b->d_cls_scratch[k], (size_t)b->layout.scratch_words[k] * b->Bp * 4));
    }
    b->dp.prog = b->d_prog;
    b->dp.prog_offset = b->d_prog_offset;
    b->dp.consts = b->d_consts;
    b->dp.bytecode = b->d_bytecode;
    b->dp.Mem = b->d_Mem;
    b->dp.grumpkin = GrumpkinTables{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    b->dp.ped_seed = nullptr;
    b->dp.ecdsa_g = nullptr;
    b->dp.fc_store = nullptr;
    b->dp.slot_of = nullptr;
    b->dp.byte_plane_of = b->d_byte_plane_of;
    b->dp.byte_plane = b->d_byte_plane;
    {   // limits of the Brillig VM for the level kernels and the first pass of the exact kernels (tuning.hpp)
        const Tuning &tn = p.tune;
        b->dp.brillig.steps = 1u << (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_steps_log2, 0), 31);
        b->dp.brillig.call_depth = (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_call_depth, 1), 1 << 20);
        b->dp.brillig.mem_cap = 0;
        b->dp.brillig.stride = 0;
        for (uint32_t oi = 0; oi < p.n_opcodes; oi++)
            if (p.prog[p.prog_offset[oi]] == PK_BRILLIG) b->br_max_regs = std::max(b->br_max_regs, p.prog[p.prog_offset[oi] + 7]);
    }
    if (p.n_digest_segments) {
        HIPCHK(hipMalloc((void **)&b->d_leaves, (size_t)p.n_digest_segments * 2 * b->Bp * sizeof(uint4)));
        if (int rc = ensure_digest_tables(b)) return rc;
    }
    if (!p.slot_of.empty()) {
        if (int rc = upload(&b->d_slot_of, p.slot_of)) return rc;
        b->dp.slot_of = b->d_slot_of;
        std::vector<uint32_t> rows(p.initial_ids.size());
        for (size_t i = 0; i < rows.size(); i++) rows[i] = p.slot_of[p.initial_ids[i]];
        if (int rc = upload(&b->d_init_rows, rows)) return rc;
    }
    if (!p.fc_slot_opcode.empty()) {
        b->fc_slots.resize(p.fc_slot_opcode.size());
        std::vector<FcStoreSlot> tab(b->fc_slots.size(), FcStoreSlot{nullptr, nullptr});
        if (int rc = upload(&b->d_fc_store, tab)) return rc;
        b->dp.fc_store = b->d_fc_store;
    }
    if (p.needs_ecdsa || p.needs_grumpkin) {  // the handle holds the device's table set until it is destroyed
        device_tables_retain(b->device);
        b->holds_tables = true;
    }
    if (p.needs_ecdsa) {
        b->dp.ecdsa_g = ecdsa_generator_tables();
        if (!b->dp.ecdsa_g) return set_err(ACVM_E_DEVICE, "could not build the ECDSA generator tables on the device");
    }
    if (p.needs_grumpkin) {
        // the level schedule's Pedersen kernel reads the 503 MB pair table (one mixed addition per 18 bits of input)
        const bool pairs = !p.cls_offset[CLS_PEDERSEN].empty();
        const bool windows = pairs && p.tune.pedersen_window_bits != 0;  // 23.6 GB of 24-bit windows (GRUMPKIN_PEDW_BITS) instead of the 503 MB of pairs
        bool windows_built = false, have = false;
        GrumpkinTables tabs;
        const GrumpkinTables *t = &tabs;
        if (windows) {
            have = windows_built = grumpkin_window_table(&tabs);
            if (!have) (void)hipGetLastError();  // (no room for the window table beside what the process holds: the pair table serves the same kernel)
        }
        if (!have) have = pairs ? grumpkin_pair_table(&tabs) : grumpkin_tables(&tabs);
        if (!have) return set_err(ACVM_E_DEVICE, "could not build the Grumpkin tables on the device");
        b->dp.grumpkin = *t;
        if (!windows_built) b->dp.grumpkin.pedw = nullptr;  // (another batch of the process may have built it: this one was planned for the pair table)
        if (!p.pedersen_seeds.empty()) {  // the instance-independent head of every Pedersen chain (kernels_grumpkin.hip)
            std::vector<uint32_t> keys;
            for (auto &k : p.pedersen_seeds) { keys.push_back(k.first); keys.push_back(k.second); }
            uint32_t *d_keys = nullptr;
            if (int rc = upload(&d_keys, keys)) return rc;
            HIPCHK(hipMalloc((void **)&b->d_ped_seed, p.pedersen_seeds.size() * 64));
            launch_pedersen_seeds(b->stream, *t, d_keys, (uint32_t)p.pedersen_seeds.size(), b->d_ped_seed);
            HIPCHK(hipStreamSynchronize(b->stream));
            hipFree(d_keys);
        }
        b->dp.ped_seed = b->d_ped_seed;
    }
    b->n_words = (p.n_witnesses + 31) / 32;
    if (int rc = upload(&b->d_producer, p.producer)) return rc;
    if (int rc = upload(&b->d_prog_class, p.prog_class)) return rc;
    if (int rc = upload(&b->d_dyn_offset, p.dyn_offset)) return rc;
    {
        size_t bytes = (size_t)p.n_inverse_slots * 2 * b->Bp * sizeof(uint4);
        HIPCHK(hipMalloc((void **)&b->d_inv, bytes ? bytes : 16));
    }
    {   // event words, with the count of flagged instances and the device address of the host-visible counter in front (ops_common.hpp flag_instance)
        HIPCHK(hipMalloc((void **)&b->d_event_base, ((size_t)b->B + 4 + 4) * 4));
        b->d_event = b->d_event_base + 4;
        HIPCHK(hipHostMalloc((void **)&b->h_flag_count, 64, hipHostMallocMapped));
        void *d_count = nullptr;
        HIPCHK(hipHostGetDevicePointer(&d_count, b->h_flag_count, 0));
        uint32_t hdr[4] = {0, 0, 0, 0};
        memcpy(&hdr[2], &d_count, sizeof d_count);
        HIPCHK(hipMemcpy(b->d_event_base, hdr, sizeof hdr, hipMemcpyHostToDevice));
    }
    b->unscale = Unscale{b->d_unscale_index, b->d_unscale_consts, b->d_unscale_plain, b->d_scaled_ids, (uint32_t)p.scaled_ids.size(), b->d_event};
    b->h_event.assign(b->B, 0xFFFFFFFFu);
    b->slow_index.assign(b->B, -1);
    return 0;
}

acvm_batch_t *acvm_batch_new(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances,
                             const uint32_t *initial_ids, uint32_t n_initial) {
    return acvm_batch_new_ex(c, solver, n_instances, initial_ids, n_initial, 0, nullptr, 0);
}
acvm_batch_t *acvm_batch_new_ex(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances, const uint32_t *initial_ids,
                                uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids, uint32_t n_keep) try {
    if (!c || (n_initial && !initial_ids) || (n_keep && !keep_ids)) { set_err(ACVM_E_INVALID, "null argument"); return nullptr; }
    if (flags & ~(uint32_t)(ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) { set_err(ACVM_E_INVALID, "unknown batch flag"); return nullptr; }
    if (solver && (!solver->schnorr_verify || !solver->pedersen || !solver->fixed_base_scalar_mul)) {
        set_err(ACVM_E_INVALID, "acvm_bb_solver_t with a null function pointer");
        return nullptr;
    }
    {
        std::vector<uint32_t> ids(initial_ids, initial_ids + n_initial);
        std::sort(ids.begin(), ids.end());
        if (std::adjacent_find(ids.begin(), ids.end()) != ids.end()) { set_err(ACVM_E_INVALID, "duplicate initial witness id"); return nullptr; }
    }
    auto b = std::make_unique<acvm_batch>();
    if (solver) { b->has_solver = true; b->solver = *solver; }
    b->opts.host_blackbox = solver != nullptr;
    b->opts.fold_digest = (flags & (ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) != 0;  // a recycled row must be hashed before it is reused
    b->opts.reuse_slots = (flags & ACVM_BATCH_REUSE_SLOTS) != 0;
    b->opts.keep.assign(keep_ids, keep_ids + n_keep);
    b->plan_ref = plan_for(c, initial_ids, n_initial, b->opts);
    if (!b->plan().unsupported.empty()) {
        set_err(ACVM_E_UNSUPPORTED, b->plan().unsupported);
        return nullptr;
    }
    b->B = b->capacity = n_instances;
    b->Bp = ((uint64_t)n_instances + 63) / 64 * 64;
    if (batch_init(b.get()) != 0) return nullptr;
    return b.release();
} ABI_CATCH_PTR
void acvm_batch_free(acvm_batch_t *b) { delete b; }

int batch_import_async(acvm_batch *b, const void *d_values_be32, hipEvent_t imported) {
    HIPCHK(hipSetDevice(b->device));
    // (acvm_batch_solve_then_import put exactly this import behind the previous solve, and it ran: the rows are there, in stream order)
    const bool already = b->next_imported && b->next_inputs == d_values_be32;
    b->next_imported = false;
    b->next_inputs = nullptr;
    if (!already)
        b->events_fresh = launch_import(b->stream, b->d_W, b->Bp, b->B, (const uint8_t *)d_values_be32, b->reuse() ? b->d_init_rows : b->d_init_ids,
                                        (uint32_t)b->plan().initial_ids.size(), nullptr, b->d_byte_plane_of_input, b->d_byte_plane, b->d_event);
    HIPCHK(hipGetLastError());
    if (imported) HIPCHK(hipEventRecord(imported, b->stream));
    b->inputs_set = true;
    b->solved = false;
    b->stepping = false;
    clear_fc_store(b);
    return 0;
}
int acvm_batch_set_initial_witness_device(acvm_batch_t *b, const void *d_values_be32) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    const bool already = b->next_imported && b->next_inputs == d_values_be32;
    if (int rc = batch_import_async(b, d_values_be32, nullptr)) return rc;
    // the caller may reuse its buffer as soon as the call returns (an import that ran behind the previous solve left the buffer alone since)
    if (!already) HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
} ABI_CATCH

int acvm_batch_set_initial_witness(acvm_batch_t *b, const uint8_t *values_be32) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    size_t bytes = (size_t)b->B * b->plan().initial_ids.size() * 32;
    if (bytes && !values_be32) return set_err(ACVM_E_INVALID, "null values");
    HIPCHK(hipSetDevice(b->device));
    if (int rc = stage_reserve(b, bytes)) return rc;
    if (bytes) HIPCHK(hipMemcpyAsync(b->d_stage, values_be32, bytes, hipMemcpyHostToDevice, b->stream));
    return acvm_batch_set_initial_witness_device(b, b->d_stage);
} ABI_CATCH

int acvm_batch_set_force_slow_path(acvm_batch_t *b, int on) {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    if (on && b->reuse()) return set_err(ACVM_E_UNSUPPORTED, "the exact path for every instance needs the full witness table: not with ACVM_BATCH_REUSE_SLOTS");
    b->force_slow = on != 0;
    return 0;
}
int acvm_batch_set_profiling(acvm_batch_t *b, int on) {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    b->profiling = on != 0;
    return 0;
}
int acvm_batch_set_instances(acvm_batch_t *b, uint32_t n) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    return batch_set_live_count(b, n);
} ABI_CATCH
int acvm_batch_reset(acvm_batch_t *b) {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    b->solved = false;
    b->stepping = false;
    clear_fc_store(b);
    return 0;
}


int acvm_batch_stats(acvm_batch_t *b, acvm_stats_t *out) {
    if (!b || !out) return set_err(ACVM_E_INVALID, "null argument");
    const Plan &p = b->plan();
    plan_stats(p, out);
    out->n_kernel_launches = b->n_launches;
    out->n_slow_instances = (uint32_t)b->slow_ids.size();
    out->solve_device_ms = b->solve_device_ms;
    out->arith_kernel_ms = b->arith_kernel_ms;
    out->dyn_kernel_ms = b->dyn_kernel_ms;
    for (int k = 0; k < 4; k++) out->class_kernel_ms[k] = b->cls_kernel_ms[k];
    out->class_kernel_ms[CLS_GRUMPKIN] += b->cls_kernel_ms[CLS_PEDERSEN] + b->cls_kernel_ms[CLS_ECDSA] + b->cls_kernel_ms[CLS_HOSTBB];
    out->slow_path_ms = b->slow_path_ms;
    out->n_brillig_retries = b->n_brillig_retries;
    for (const SchedStep &st : b->schedule.steps) {
        if (st.kind == SK_LAUNCH && st.stream < 6) out->n_stream_launches[st.stream]++;
        else if (st.kind == SK_WAIT) out->n_stream_waits++;
    }
    return 0;
}


