// node.cpp -- the node-level driver behind the C ABI (include/acvm_amd.h acvm_node_*): ONE call solves a global batch of witness
// instances of one circuit on every GPU of the node. It replaces the loop a caller of the reference runs once per instance
// (acvm_js/src/execute.rs:60-119: ACVM::new, solve, finalize / error string, witness extraction) and does what SURVEY 8e describes:
//   * the global batch is split contiguously over the devices (device d owns instances [d * per, (d + 1) * per)), no exchange step;
//   * per device one host thread drives one batch handle (one levelised plan, one witness table) through its shard in tiles;
//   * the inputs of tile k + 1 travel host -> pinned staging -> device beside the solve of tile k (a second thread per device fills
//     the two staging buffers and issues hipMemcpyAsync on a copy stream);
//   * instances that leave the generic path are re-solved by the exact kernels in a side table on a stream of their own, beside the
//     level schedule of the NEXT tile (batch.cpp, asynchronous exact path): a few diverging inputs per tile do not stall the tile;
//   * per instance the caller gets the result record, the kept witnesses (normally the circuit's return values) and the 32-byte
//     digest of the whole witness map, in global instance order.
#include "batch.hpp"
#include "circuit.hpp"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <unistd.h>

namespace {

struct DeviceLane {
    int device = 0;
    acvm_batch_t *batch = nullptr;
    bool async = false;
    uint8_t *pinned[2] = {nullptr, nullptr};
    uint8_t *d_in[2] = {nullptr, nullptr};
    hipStream_t copy = nullptr, out = nullptr;  // uploads of the producer thread; downloads of the kept witnesses
    hipEvent_t ev_h2d[2] = {nullptr, nullptr};
    hipEvent_t ev_imported[2] = {nullptr, nullptr};  // the import kernel has read staging slot k: the next upload into it may start
    hipEvent_t ev_exported[2] = {nullptr, nullptr}, ev_arrived[2] = {nullptr, nullptr};  // kept witnesses: export kernel done / copy in pinned memory
    // kept witnesses leave the device beside the NEXT tile's solve: export kernel + D2H into pinned memory are enqueued behind a solve and
    // harvested after the following one
    uint32_t *d_keep = nullptr;
    uint8_t *d_exp[2] = {nullptr, nullptr}, *h_exp[2] = {nullptr, nullptr};
    // statistics of the last solve
    double device_ms = 0, h2d_wait_ms = 0, export_ms = 0, total_ms = 0;
    uint32_t tiles = 0, exact_instances = 0;
    uint64_t not_solved = 0;
    std::string error;
    int rc = 0;
    // host placement: the CPUs local to the device (sysfs), empty = unknown / not pinned
    int numa_node = -1;
    std::vector<uint32_t> cpus;
};

// "0-15,32-47" (sysfs cpulist format) -> CPU numbers; anything malformed ends the list where it stands
std::vector<uint32_t> parse_cpulist(const std::string &text) {
    std::vector<uint32_t> out;
    size_t i = 0;
    auto number = [&](uint32_t &v) {
        if (i >= text.size() || text[i] < '0' || text[i] > '9') return false;
        uint64_t x = 0;
        while (i < text.size() && text[i] >= '0' && text[i] <= '9' && x < (1u << 20)) x = x * 10 + (uint64_t)(text[i++] - '0');
        v = (uint32_t)x;
        return x < (1u << 20);
    };
    while (i < text.size()) {
        uint32_t a, b;
        if (!number(a)) break;
        b = a;
        if (i < text.size() && text[i] == '-') { i++; if (!number(b) || b < a) break; }
        for (uint32_t c = a; c <= b && out.size() < 4096; c++) out.push_back(c);
        if (i < text.size() && text[i] == ',') i++;
        else break;
    }
    return out;
}
// NUMA node and local CPUs of a PCI device from sysfs (<root>/<bus id>/numa_node, local_cpulist); false if the files are not there
bool device_locality(const std::string &pci_root, const std::string &bus_id, int *numa_node, std::vector<uint32_t> *cpus) {
    std::string id = bus_id;
    for (char &ch : id) ch = (char)tolower((unsigned char)ch);
    std::ifstream fn(pci_root + "/" + id + "/numa_node"), fc(pci_root + "/" + id + "/local_cpulist");
    if (!fn || !fc) return false;
    int node = -1;
    fn >> node;
    std::string list;
    std::getline(fc, list);
    *numa_node = node;
    *cpus = parse_cpulist(list);
    return true;
}
// the calling thread onto the given CPUs (and with it every thread it creates from now on); silently not when the set is empty or refused
// (only CPUs the process may use at all: the device's local list is intersected with the thread's current mask -- a container's cpuset, a
// caller's taskset; an empty intersection leaves the thread where it is)
void pin_thread(const std::vector<uint32_t> &cpus) {
    if (cpus.empty()) return;
    cpu_set_t allowed, set;
    CPU_ZERO(&allowed);
    const bool have_allowed = pthread_getaffinity_np(pthread_self(), sizeof allowed, &allowed) == 0;
    CPU_ZERO(&set);
    int n = 0;
    for (uint32_t c : cpus)
        if (c < CPU_SETSIZE && (!have_allowed || CPU_ISSET(c, &allowed))) { CPU_SET(c, &set); n++; }
    if (n) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}
// threads that are joined when the scope ends, whichever way it ends
struct Joiner {
    std::vector<std::thread> t;
    ~Joiner() {
        for (auto &x : t)
            if (x.joinable()) x.join();
    }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct acvm_node {
    std::vector<DeviceLane> lanes;
    std::vector<uint32_t> ids, keep;
    uint32_t tile = 0, flags = 0;
    uint64_t last_n = 0;
    double last_total_ms = 0;
    std::shared_ptr<const Plan> plan;  // the one plan every lane's handle shares
    uint32_t plans_built = 0;          // how often acvm_node_new levelised the circuit (0: the circuit's cache already held the plan; never more than 1)
    double create_ms = 0, plan_ms = 0;
    ~acvm_node() {
        for (DeviceLane &l : lanes) {
            hipSetDevice(l.device);
            if (l.batch) acvm_batch_free(l.batch);
            for (int k = 0; k < 2; k++) {
                if (l.pinned[k]) hipHostFree(l.pinned[k]);
                if (l.d_in[k]) hipFree(l.d_in[k]);
                for (hipEvent_t e : {l.ev_h2d[k], l.ev_imported[k], l.ev_exported[k], l.ev_arrived[k]})
                    if (e) hipEventDestroy(e);
                if (l.d_exp[k]) hipFree(l.d_exp[k]);
                if (l.h_exp[k]) hipHostFree(l.h_exp[k]);
            }
            if (l.d_keep) hipFree(l.d_keep);
            if (l.copy) hipStreamDestroy(l.copy);
            if (l.out) hipStreamDestroy(l.out);
        }
    }
};

namespace {

// instances per handle when the caller leaves the choice to the library: the largest power of two (at most 2^17, the measured optimum of
// the 10k-gate circuit, NOTEBOOK.md section 7) that every listed device has room for. Per device: 90 % of its free memory, less the lookup
// tables the circuit would still build there, divided by the number of handles the device is listed for; a handle needs its tables, class
// scratch and first side table (batch_device_bytes) plus this driver's staging and export buffers.
uint32_t auto_tile(const Plan &p, const PlanOpts &opts, const std::vector<uint32_t> &ids, const std::vector<uint32_t> &keep, const std::vector<int> &devices) {
    const bool pedersen_level = !p.cls_offset[CLS_PEDERSEN].empty();
    const bool window_table = pedersen_level && p.tune.pedersen_window_bits != 0;
    double budget = 1e30;
    for (int d : devices) {
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(d) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { set_err(ACVM_E_DEVICE, "hipMemGetInfo failed"); return 0; }
        const double tables = (double)device_tables_missing_bytes(p.needs_grumpkin, pedersen_level, window_table, p.needs_ecdsa);
        const double copies = (double)std::count(devices.begin(), devices.end(), d);
        budget = std::min(budget, (0.9 * (double)free_b - tables) / copies);
    }
    const double io_per_instance = 2.0 * 32.0 * (double)ids.size() * 2.0 /* pinned + device staging are separate pools; count the device one twice for slack */ +
                                   2.0 * 32.0 * (double)keep.size();
    uint32_t t = 1u << 17;
    while (t > 64 && (double)batch_device_bytes(p, opts, t, true) + io_per_instance * t > budget) t >>= 1;
    if ((double)batch_device_bytes(p, opts, t, true) + io_per_instance * t > budget) { set_err(ACVM_E_DEVICE, "not enough free device memory for a tile of 64 instances of this circuit"); return 0; }
    return t;
}

// rows [first, first + n) of the caller's inputs into a pinned buffer (a partial tile fills and uploads its own rows only: batch_set_live_count)
// (a tile of the 10k-gate circuit is 64 MB: one core copies that in ~10 ms, which would be exposed in front of the first tile; four do it in ~3)
void fill_staging(uint8_t *dst, const uint8_t *values, size_t row, uint64_t first, uint32_t n) {
    if (row == 0) return;
    const uint8_t *src = values + first * row;
    const size_t bytes = (size_t)n * row;
    constexpr size_t PIECE = 8u << 20;
    if (bytes <= 2 * PIECE) memcpy(dst, src, bytes);
    else {
        const size_t part = (bytes / 4 + 63) / 64 * 64;
        Joiner helpers;  // (a thread that cannot be created throws: the ones that run are joined before the exception leaves)
        for (int q = 0; q < 3; q++)
            helpers.t.emplace_back([=] { const size_t at = (size_t)(q + 1) * part; if (at < bytes) memcpy(dst + at, src + at, std::min(part, bytes - at)); });
        memcpy(dst, src, std::min(part, bytes));
    }
}

// an outcome of the exact path of the tile that started at global instance `base` (n_valid instances of it are the caller's)
uint64_t patch_outcome(const ExactOutcome &o, uint64_t base, uint32_t n_valid, uint32_t n_keep, acvm_result_t *results, uint8_t *kept, uint8_t *kept_assigned,
                       uint8_t *digests) {
    uint64_t not_solved = 0;
    for (size_t t = 0; t < o.instance.size(); t++) {
        const uint32_t j = o.instance[t];
        if (j >= n_valid) continue;
        const uint64_t g = base + j;
        not_solved += o.results[t].status != ACVM_STATUS_SOLVED;
        if (results) results[g] = o.results[t];
        if (kept && n_keep && !o.kept_values.empty()) memcpy(kept + g * n_keep * 32, &o.kept_values[t * (size_t)n_keep * 32], (size_t)n_keep * 32);
        if (kept_assigned && n_keep && !o.kept_assigned.empty()) memcpy(kept_assigned + g * n_keep, &o.kept_assigned[t * (size_t)n_keep], n_keep);
        if (digests && !o.digests.empty()) memcpy(digests + g * 32, &o.digests[t * 32], 32);
    }
    return not_solved;
}

void run_lane_body(acvm_node *node, DeviceLane &L, uint64_t first, uint64_t last, const uint8_t *values, acvm_result_t *results, uint8_t *kept, uint8_t *kept_assigned,
                   uint8_t *digests) {
    const double t_begin = now_ms();
    L.rc = 0;
    L.error.clear();
    L.device_ms = L.h2d_wait_ms = L.export_ms = 0;
    L.tiles = L.exact_instances = 0;
    L.not_solved = 0;
    auto fail = [&](int rc, const std::string &what) { if (!L.rc) { L.rc = rc; L.error = what + ": " + acvm_last_error(); } };
    if (hipSetDevice(L.device) != hipSuccess) { fail(ACVM_E_DEVICE, "hipSetDevice"); return; }
    const uint32_t tile = node->tile, n_keep = (uint32_t)node->keep.size();
    const size_t row = node->ids.size() * 32;
    const uint64_t n = last - first;
    const uint32_t n_tiles = (uint32_t)((n + tile - 1) / tile);
    if (!n_tiles) return;
    // ---- producer: host rows -> pinned staging -> device, one tile ahead of the solver. Every failure of it reaches the solver (producer_rc):
    // a solver that waited for a tile that never comes would hang, one that went on after a failed copy would solve stale inputs.
    std::mutex mu;
    std::condition_variable cv;
    uint32_t staged = 0, consumed = 0;  // tiles whose H2D was issued / whose import was enqueued (the staging slot is handed back through ev_imported)
    bool abort = false;
    int producer_rc = 0;
    std::string producer_err;
    std::thread producer([&] {
        auto give_up_rc = [&](int rc, const std::string &what) {
            {
                std::lock_guard<std::mutex> lk(mu);
                producer_rc = rc;
                producer_err = what;
            }
            cv.notify_all();
        };
        auto give_up = [&](const char *what, hipError_t e) { give_up_rc(ACVM_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
        try {  // nothing unwinds out of a thread (std::terminate) nor through the ABI: an allocation failure ends the lane with ACVM_E_NOMEM
        if (hipError_t e = hipSetDevice(L.device); e != hipSuccess) { give_up("hipSetDevice (upload thread)", e); return; }  // HIP's current device is per thread
        for (uint32_t k = 0; k < n_tiles; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return abort || k < consumed + 2; });
                if (abort) return;
            }
            const uint64_t base = first + (uint64_t)k * tile;
            const uint32_t m = (uint32_t)std::min<uint64_t>(tile, last - base);
            const int slot = (int)(k & 1);
            // the pinned half of the slot is free once its previous upload has been read by the copy engine (ev_h2d, waited for by the solver
            // before it enqueued that tile's import), the device half once that import kernel has run (ev_imported)
            if (k >= 2)
                if (hipError_t e = hipStreamWaitEvent(L.copy, L.ev_imported[slot], 0); e != hipSuccess) { give_up("hipStreamWaitEvent", e); return; }
            fill_staging(L.pinned[slot], values, row, base, m);
            if (row)
                if (hipError_t e = hipMemcpyAsync(L.d_in[slot], L.pinned[slot], (size_t)m * row, hipMemcpyHostToDevice, L.copy); e != hipSuccess) { give_up("hipMemcpyAsync (upload)", e); return; }
            if (hipError_t e = hipEventRecord(L.ev_h2d[slot], L.copy); e != hipSuccess) { give_up("hipEventRecord", e); return; }
            {
                std::lock_guard<std::mutex> lk(mu);
                staged = k + 1;
            }
            cv.notify_all();
        }
        } catch (const std::bad_alloc &) {
            give_up_rc(ACVM_E_NOMEM, "out of host memory in the upload thread");
        } catch (const std::exception &e) {
            give_up_rc(ACVM_E_INVALID, std::string("upload thread: ") + e.what());
        }
    });
    auto stop_producer = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            abort = true;
        }
        cv.notify_all();
        producer.join();
    };
    try {  // (an exception of the solver side -- std::bad_alloc of its bookkeeping vectors -- must not skip the join of the producer below)
    // ---- solver
    uint64_t prev_base = 0;
    uint32_t prev_valid = 0;
    bool in_flight = false;  // kept witnesses of a tile on their way to pinned memory
    int fl_slot = 0;
    uint64_t fl_base = 0;
    uint32_t fl_n = 0;
    std::vector<uint32_t> fl_flagged;
    auto harvest = [&]() -> bool {  // the kept witnesses of the tile in flight: wait for their copy (it ran beside the solve that followed), scatter
        if (!in_flight) return true;
        in_flight = false;
        if (hipEventSynchronize(L.ev_arrived[fl_slot]) != hipSuccess) { fail(ACVM_E_DEVICE, "hipEventSynchronize (kept witnesses)"); return false; }
        memcpy(kept + fl_base * n_keep * 32, L.h_exp[fl_slot], (size_t)fl_n * n_keep * 32);
        std::vector<uint8_t> flagged(fl_n, 0);
        for (uint32_t j : fl_flagged)
            if (j < fl_n) flagged[j] = 1;
        for (uint32_t k2 = 0; k2 < n_keep; k2++) {  // the level kernels' assigned set is the planner's
            const bool produced = batch_generic_assigned(L.batch, node->keep[k2]);
            for (uint32_t j = 0; j < fl_n; j++) {
                if (flagged[j]) continue;
                if (kept_assigned) kept_assigned[(fl_base + j) * n_keep + k2] = produced;
                if (!produced) memset(kept + ((fl_base + j) * n_keep + k2) * 32, 0, 32);
            }
        }
        return true;
    };
    for (uint32_t k = 0; k < n_tiles && !L.rc; k++) {
        const uint64_t base = first + (uint64_t)k * tile;
        const uint32_t m = (uint32_t)std::min<uint64_t>(tile, last - base);
        const int slot = (int)(k & 1);
        const double t0 = now_ms();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return staged > k || producer_rc != 0; });
            if (producer_rc) { L.rc = producer_rc; L.error = "upload of tile " + std::to_string(k) + ": " + producer_err; break; }
        }
        if (hipEventSynchronize(L.ev_h2d[slot]) != hipSuccess) { fail(ACVM_E_DEVICE, "hipEventSynchronize"); break; }
        L.h2d_wait_ms += now_ms() - t0;
        if (int rc = batch_set_live_count(L.batch, m)) { fail(rc, "live count"); break; }  // a partial last tile: the lanes behind it are dead
        // the import is enqueued, not waited for: it runs behind the export kernel of the previous tile and in front of this tile's levels
        if (int rc = batch_import_async(L.batch, L.d_in[slot], L.ev_imported[slot])) { fail(rc, "set_initial_witness"); break; }
        {   // the producer may prepare the tile after next (its upload waits for ev_imported on the copy stream)
            std::lock_guard<std::mutex> lk(mu);
            consumed = k + 1;
        }
        cv.notify_all();
        const int rc = acvm_batch_solve(L.batch);  // asynchronous handles: collects the exact job of tile k - 1 on the way
        if (rc < 0) { fail(rc, "solve"); break; }
        acvm_stats_t st;
        acvm_batch_stats(L.batch, &st);
        L.device_ms += st.solve_device_ms;
        L.exact_instances += st.n_slow_instances;
        L.tiles++;
        const double t1 = now_ms();
        // the kept witnesses of the PREVIOUS tile: their copy ran on the download stream beside this solve
        if (!harvest()) break;
        if (k > 0) {
            ExactOutcome o;
            batch_take_outcome(L.batch, &o);
            if (!o.instance.empty()) L.not_solved += patch_outcome(o, prev_base, prev_valid, n_keep, results, kept, kept_assigned, digests);
        }
        const bool pending = batch_exact_pending(L.batch);
        const std::vector<uint32_t> &exact_ids = *batch_exact_instances(L.batch);
        L.not_solved += batch_exact_unsolved(L.batch, m);  // (a synchronous exact path: its lanes are final)
        // instances of a SYNCHRONOUS exact path have their values (and assigned sets) where only the batch knows: such a tile exports in one
        // synchronous step; every other tile enqueues its kept witnesses and goes on
        const bool overlap = kept && n_keep && (exact_ids.empty() || pending);
        if (int rc2 = batch_export_tile(L.batch, m, node->keep.data(), overlap ? 0u : n_keep, results ? results + base : nullptr,
                                        kept && !overlap ? kept + base * n_keep * 32 : nullptr, kept_assigned && !overlap ? kept_assigned + base * n_keep : nullptr,
                                        digests ? digests + base * 32 : nullptr)) {
            fail(rc2, "export");
            break;
        }
        if (overlap) {
            if (int rc3 = batch_enqueue_kept(L.batch, m, L.d_keep, n_keep, L.d_exp[slot], L.h_exp[slot], L.out, L.ev_exported[slot], L.ev_arrived[slot])) { fail(rc3, "export of the kept witnesses"); break; }
            in_flight = true;
            fl_slot = slot;
            fl_base = base;
            fl_n = m;
            fl_flagged.assign(exact_ids.begin(), exact_ids.end());  // (arrive with the exact job's outcome)
        }
        L.export_ms += now_ms() - t1;
        prev_base = base;
        prev_valid = m;
    }
    if (!L.rc) harvest();
    if (!L.rc) {  // the exact job of the last tile
        ExactOutcome o;
        if (int rc = batch_finish_pending(L.batch, &o)) fail(rc, "exact path");
        else L.not_solved += patch_outcome(o, prev_base, prev_valid, n_keep, results, kept, kept_assigned, digests);
    }
    } catch (const std::bad_alloc &) {
        if (!L.rc) { L.rc = ACVM_E_NOMEM; L.error = "out of host memory in the lane's solver thread"; }
    } catch (const std::exception &e) {
        if (!L.rc) { L.rc = ACVM_E_INVALID; L.error = std::string("lane: ") + e.what(); }
    }
    stop_producer();
    hipStreamSynchronize(L.copy);
    hipStreamSynchronize(L.out);
    if (L.rc) {  // leave the handle reusable
        ExactOutcome o;
        batch_finish_pending(L.batch, &o);
        acvm_device_synchronize();  // (an import enqueued for a tile that was never solved)
    }
    L.total_ms = now_ms() - t_begin;
}

// a lane's thread: pinned to the CPUs of its device's NUMA node (its upload thread and the staging helpers inherit the mask), and closed
// against exceptions: std::terminate is not an error code
void run_lane(acvm_node *node, DeviceLane &L, uint64_t first, uint64_t last, const uint8_t *values, acvm_result_t *results, uint8_t *kept, uint8_t *kept_assigned,
              uint8_t *digests) {
    try {
        pin_thread(L.cpus);
        run_lane_body(node, L, first, last, values, results, kept, kept_assigned, digests);
    } catch (const std::bad_alloc &) {
        if (!L.rc) { L.rc = ACVM_E_NOMEM; L.error = "out of host memory"; }
    } catch (const std::exception &e) {
        if (!L.rc) { L.rc = ACVM_E_INVALID; L.error = e.what(); }
    } catch (...) {
        if (!L.rc) { L.rc = ACVM_E_INVALID; L.error = "unknown exception"; }
    }
}

}  // namespace

extern "C" {

acvm_node_t *acvm_node_new(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, const uint32_t *initial_ids, uint32_t n_initial, const uint32_t *keep_ids,
                           uint32_t n_keep, const acvm_node_opts_t *opts) try {
    if (!c || (n_initial && !initial_ids) || (n_keep && !keep_ids)) { set_err(ACVM_E_INVALID, "null argument"); return nullptr; }
    const int visible = acvm_device_count();
    if (visible < 1) { set_err(ACVM_E_DEVICE, "no HIP device visible; the library has no CPU fallback"); return nullptr; }
    auto node = std::make_unique<acvm_node>();
    node->ids.assign(initial_ids, initial_ids + n_initial);
    node->keep.assign(keep_ids, keep_ids + n_keep);
    // a kept witness beyond the circuit is simply never assigned (WitnessMap::get -> None): on the device it travels as 0xFFFFFFFF, "no row"
    for (uint32_t &w : node->keep)
        if (w >= acvm_circuit_num_witnesses(c)) w = 0xFFFFFFFFu;
    node->flags = opts ? opts->batch_flags : 0;
    std::vector<int> devices;
    // n_devices == 0 selects every visible device and `devices` is not read (it holds n_devices entries: none)
    const bool listed = opts && opts->n_devices && opts->devices;
    const uint32_t n_dev = opts && opts->n_devices ? opts->n_devices : (uint32_t)visible;
    if (n_dev > 16) { set_err(ACVM_E_INVALID, "at most 16 handles per node (acvm_node_stats_t)"); return nullptr; }
    for (uint32_t i = 0; i < n_dev; i++) {
        const int d = listed ? opts->devices[i] : (int)i;
        if (d < 0 || d >= visible) { set_err(ACVM_E_INVALID, "device index " + std::to_string(d) + " out of range (" + std::to_string(visible) + " visible)"); return nullptr; }
        devices.push_back(d);
    }
    // The circuit is levelised ONCE per node (the reference's callers build one opcode list per circuit for any number of executions,
    // acvm_js/src/execute.rs:60-119): the plan is immutable and shared -- auto_tile sizes the handles by it and every lane's
    // acvm_batch_new_ex finds it in the circuit's plan cache (batch.cpp plan_for) instead of planning again (until round 5: 1 + N times).
    const double t_create = now_ms();
    const uint64_t built_before = acvm_circuit_plans_built(c);
    {
        PlanOpts po;  // exactly what acvm_batch_new_ex derives from the same arguments
        po.host_blackbox = solver != nullptr;
        po.fold_digest = (node->flags & (ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) != 0;
        po.reuse_slots = (node->flags & ACVM_BATCH_REUSE_SLOTS) != 0;
        po.keep = node->keep;
        node->plan = plan_for(c, node->ids.data(), n_initial, po);
        if (!node->plan->unsupported.empty()) { set_err(ACVM_E_UNSUPPORTED, node->plan->unsupported); return nullptr; }
        node->plan_ms = node->plan->plan_ms;
        node->tile = opts && opts->tile_instances ? opts->tile_instances : auto_tile(*node->plan, po, node->ids, node->keep, devices);
    }
    if (!node->tile) return nullptr;
    node->lanes.resize(devices.size());
    const size_t row = (size_t)n_initial * 32;
    // one handle per device, created side by side (each allocates tens of GB and levelises the circuit)
    std::vector<std::thread> th;
    for (size_t i = 0; i < devices.size(); i++) {
        node->lanes[i].device = devices[i];
        th.emplace_back([&, i] {
            DeviceLane &L = node->lanes[i];
            try {  // (nothing leaves a thread: std::terminate is not an error code)
            auto fail = [&](int rc, const std::string &what) { L.rc = rc; L.error = what + ": " + acvm_last_error(); };
            if (hipSetDevice(L.device) != hipSuccess) { L.rc = ACVM_E_DEVICE; L.error = "hipSetDevice failed"; return; }
            {   // where the device hangs: its NUMA node and the CPUs next to it (pinned staging is allocated and filled from there)
                char bus[32] = {0};
                if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, L.device) == hipSuccess) device_locality("/sys/bus/pci/devices", bus, &L.numa_node, &L.cpus);
                else (void)hipGetLastError();
                pin_thread(L.cpus);  // (this thread allocates the lane's pinned buffers: first touch on the device's node)
            }
            L.batch = acvm_batch_new_ex(c, solver, node->tile, node->ids.data(), n_initial, node->flags, node->keep.data(), n_keep);
            if (!L.batch) { fail(ACVM_E_DEVICE, "acvm_batch_new_ex"); return; }
            const int a = batch_enable_async_exact(L.batch, node->keep.data(), n_keep, true);
            if (a < 0) { fail(a, "async exact path"); return; }
            L.async = a == 1;
            const size_t bytes = std::max<size_t>((size_t)node->tile * row, 16);
            bool ok = hipStreamCreateWithFlags(&L.copy, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&L.out, hipStreamNonBlocking) == hipSuccess;
            for (int k = 0; k < 2 && ok; k++)
                ok = hipHostMalloc((void **)&L.pinned[k], bytes, hipHostMallocDefault) == hipSuccess && hipMalloc((void **)&L.d_in[k], bytes) == hipSuccess &&
                     hipEventCreateWithFlags(&L.ev_h2d[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&L.ev_imported[k], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&L.ev_exported[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&L.ev_arrived[k], hipEventDisableTiming) == hipSuccess;
            if (ok && n_keep) {
                const size_t eb = (size_t)node->tile * n_keep * 32;
                ok = hipMalloc((void **)&L.d_keep, (size_t)n_keep * 4) == hipSuccess &&
                     hipMemcpy(L.d_keep, node->keep.data(), (size_t)n_keep * 4, hipMemcpyHostToDevice) == hipSuccess;
                for (int k = 0; k < 2 && ok; k++)
                    ok = hipMalloc((void **)&L.d_exp[k], eb) == hipSuccess && hipHostMalloc((void **)&L.h_exp[k], eb, hipHostMallocDefault) == hipSuccess;
            }
            if (!ok) { L.rc = ACVM_E_DEVICE; L.error = "staging buffers: allocation failed"; }
            } catch (const std::bad_alloc &) {
                if (!L.rc) { L.rc = ACVM_E_NOMEM; L.error = "out of host memory"; }
            } catch (const std::exception &e) {
                if (!L.rc) { L.rc = ACVM_E_INVALID; L.error = e.what(); }
            } catch (...) {
                if (!L.rc) { L.rc = ACVM_E_INVALID; L.error = "unknown exception"; }
            }
        });
    }
    for (auto &t : th) t.join();
    for (DeviceLane &L : node->lanes)
        if (L.rc) { set_err(L.rc, "device " + std::to_string(L.device) + ": " + L.error); return nullptr; }
    node->plans_built = (uint32_t)(acvm_circuit_plans_built(c) - built_before);
    node->create_ms = now_ms() - t_create;
    return node.release();
} ABI_CATCH_PTR

void acvm_node_free(acvm_node_t *n) { delete n; }

uint32_t acvm_node_tile_instances(const acvm_node_t *n) { return n ? n->tile : 0; }
uint32_t acvm_node_num_devices(const acvm_node_t *n) { return n ? (uint32_t)n->lanes.size() : 0; }

long long acvm_node_solve(acvm_node_t *n, uint64_t n_instances, const uint8_t *values_be32, acvm_result_t *results, uint8_t *kept_be32, uint8_t *kept_assigned,
                          uint8_t *digests32) try {
    if (!n || (n_instances && !n->ids.empty() && !values_be32)) return set_err(ACVM_E_INVALID, "null argument");
    const double t0 = now_ms();
    const size_t D = n->lanes.size();
    // contiguous split, multiples of 64 instances per device except the last
    std::vector<uint64_t> bound(D + 1, 0);
    const uint64_t per = ((n_instances + D - 1) / D + 63) / 64 * 64;
    for (size_t d = 0; d <= D; d++) bound[d] = std::min<uint64_t>(n_instances, per * d);
    std::vector<std::thread> th;
    for (size_t d = 0; d < D; d++)
        th.emplace_back(run_lane, n, std::ref(n->lanes[d]), bound[d], bound[d + 1], values_be32, results, kept_be32, kept_assigned, digests32);
    for (auto &t : th) t.join();
    n->last_n = n_instances;
    n->last_total_ms = now_ms() - t0;
    for (DeviceLane &L : n->lanes)
        if (L.rc) return set_err(L.rc, "device " + std::to_string(L.device) + ": " + L.error);
    long long not_solved = 0;
    for (DeviceLane &L : n->lanes) not_solved += (long long)L.not_solved;
    return not_solved;
} ABI_CATCH

int acvm_node_stats(acvm_node_t *n, acvm_node_stats_t *out) {
    if (!n || !out) return set_err(ACVM_E_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    out->n_devices = (uint32_t)n->lanes.size();
    out->tile_instances = n->tile;
    out->n_instances = n->last_n;
    out->total_ms = n->last_total_ms;
    for (size_t d = 0; d < n->lanes.size() && d < 16; d++) {
        const DeviceLane &L = n->lanes[d];
        out->device[d] = L.device;
        out->async_exact[d] = L.async;
        out->tiles[d] = L.tiles;
        out->exact_instances[d] = L.exact_instances;
        out->lane_ms[d] = L.total_ms;
        out->solve_device_ms[d] = L.device_ms;
        out->h2d_wait_ms[d] = L.h2d_wait_ms;
        out->export_ms[d] = L.export_ms;
        out->numa_node[d] = L.numa_node;
        out->n_cpus_pinned[d] = (uint32_t)L.cpus.size();
        out->first_cpu[d] = L.cpus.empty() ? -1 : (int)L.cpus.front();
    }
    out->plans_built = n->plans_built;
    out->create_ms = n->create_ms;
    out->plan_ms = n->plan_ms;
    {   // resident set of the process (the shared plan of a 10^6-opcode circuit is hundreds of MB; it used to exist once per lane)
        std::ifstream f("/proc/self/statm");
        unsigned long long pages_total = 0, pages_resident = 0;
        if (f >> pages_total >> pages_resident) out->host_rss_bytes = (uint64_t)pages_resident * (uint64_t)sysconf(_SC_PAGESIZE);
    }
    return 0;
}

// host-only probes of the placement logic (tests/test_sharding.py: no GPU, a made-up sysfs tree)
int acvm_debug_cpulist(const char *text, uint32_t *cpus, uint32_t cap) try {
    if (!text || (cap && !cpus)) return set_err(ACVM_E_INVALID, "null argument");
    const std::vector<uint32_t> v = parse_cpulist(text);
    for (size_t i = 0; i < v.size() && i < cap; i++) cpus[i] = v[i];
    return (int)v.size();
} ABI_CATCH
int acvm_debug_device_locality(const char *pci_root, const char *bus_id, int *numa_node, uint32_t *cpus, uint32_t cap) try {
    if (!pci_root || !bus_id || !numa_node || (cap && !cpus)) return set_err(ACVM_E_INVALID, "null argument");
    std::vector<uint32_t> v;
    if (!device_locality(pci_root, bus_id, numa_node, &v)) return set_err(ACVM_E_STATE, "no numa_node / local_cpulist for this device");
    for (size_t i = 0; i < v.size() && i < cap; i++) cpus[i] = v[i];
    return (int)v.size();
} ABI_CATCH

}  // extern "C"
