// shim.cpp -- host layers above the batch ABI (include/acvm_amd.h), written against that ABI only:
//   * the two fakes of the reference's tests as built-in BlackBoxFunctionSolver vtables
//     (StubbedBackend acvm/tests/solver.rs:20-46, DummyBlackBoxSolver brillig_vm/src/lib.rs:392-420);
//   * struct ACVM (acvm/src/pwg/mod.rs:129-304) for ONE instance as a handle over a batch of one, method by method;
//   * acvm_multi_*: instances whose initial WitnessMaps assign different id sets (ACVM::new takes any map, mod.rs:146):
//     instances are grouped by id set, one levelised batch per group, results in the caller's order.
#include "../../include/acvm_amd.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------- built-in fakes
int stub_schnorr(void *, const uint8_t *, const uint8_t *, const uint8_t *, size_t, const uint8_t *, size_t, uint8_t *, char *err, size_t n) {
    if (err && n) snprintf(err, n, "Path not trodden by this test");
    return 3;  // panic!
}
int stub_pedersen(void *, const uint8_t *, size_t, uint32_t, uint8_t *, uint8_t *, char *err, size_t n) {
    if (err && n) snprintf(err, n, "Path not trodden by this test");
    return 3;
}
int stub_fixed(void *, const uint8_t *, const uint8_t *, uint8_t *, uint8_t *, char *err, size_t n) {
    if (err && n) snprintf(err, n, "Path not trodden by this test");
    return 3;
}
void put_small(uint8_t out[32], uint8_t v) {
    memset(out, 0, 32);
    out[31] = v;
}
int dummy_schnorr(void *, const uint8_t *, const uint8_t *, const uint8_t *, size_t, const uint8_t *, size_t, uint8_t *ok, char *, size_t) {
    *ok = 1;
    return 0;
}
int dummy_pedersen(void *, const uint8_t *, size_t, uint32_t, uint8_t *x, uint8_t *y, char *, size_t) {
    put_small(x, 2);
    put_small(y, 3);
    return 0;
}
int dummy_fixed(void *, const uint8_t *, const uint8_t *, uint8_t *x, uint8_t *y, char *, size_t) {
    put_small(x, 4);
    put_small(y, 5);
    return 0;
}
// the constants need no per-instance call: the batched members fill n results at once
int dummy_schnorr_batch(void *, size_t n, const uint8_t *, const uint8_t *, size_t, const uint8_t *, size_t, uint8_t *ok, uint8_t *rc, char *, size_t) {
    memset(ok, 1, n);
    memset(rc, 0, n);
    return 0;
}
int dummy_pedersen_batch(void *, size_t n, const uint8_t *, size_t, uint32_t, uint8_t *xy, uint8_t *rc, char *, size_t) {
    for (size_t i = 0; i < n; i++) { put_small(xy + 64 * i, 2); put_small(xy + 64 * i + 32, 3); }
    memset(rc, 0, n);
    return 0;
}
int dummy_fixed_batch(void *, size_t n, const uint8_t *, uint8_t *xy, uint8_t *rc, char *, size_t) {
    for (size_t i = 0; i < n; i++) { put_small(xy + 64 * i, 4); put_small(xy + 64 * i + 32, 5); }
    memset(rc, 0, n);
    return 0;
}
const acvm_bb_solver_t STUBBED = {nullptr, stub_schnorr, stub_pedersen, stub_fixed, nullptr, nullptr, nullptr};
const acvm_bb_solver_t DUMMY = {nullptr, dummy_schnorr, dummy_pedersen, dummy_fixed, dummy_schnorr_batch, dummy_pedersen_batch, dummy_fixed_batch};

// (index, value) pairs of the assigned witnesses of instance `index` of batch b, ascending
long long map_pairs(acvm_batch_t *b, uint32_t index, uint32_t *ids, uint8_t *values_be32, uint32_t cap) {
    acvm_stats_t st;
    if (int rc = acvm_batch_stats(b, &st)) return rc;
    const uint32_t nw = st.n_witnesses;
    std::vector<uint8_t> assigned(nw ? nw : 1), values((size_t)(nw ? nw : 1) * 32);
    if (int rc = acvm_batch_witness_map(b, index, 1, assigned.data(), values.data())) return rc;
    long long n = 0;
    for (uint32_t w = 0; w < nw; w++)
        if (assigned[w]) {
            if ((uint64_t)n < cap) {
                if (ids) ids[n] = w;
                if (values_be32) memcpy(values_be32 + 32 * (size_t)n, &values[(size_t)w * 32], 32);
            }
            n++;
        }
    return n;
}

}  // namespace

struct acvm_instance {
    acvm_batch_t *b = nullptr;
    std::map<uint32_t, std::vector<uint8_t>> initial;  // the map handed to ACVM::new: witness_map() before the first solve call
    ~acvm_instance() { if (b) acvm_batch_free(b); }
};

struct acvm_multi {
    std::vector<acvm_batch_t *> groups;
    std::vector<uint32_t> group_of, index_in;  // per instance
    std::vector<std::vector<uint32_t>> members;  // per group: instances in batch order
    uint32_t n = 0, nw = 0;
    ~acvm_multi() { for (auto *g : groups) if (g) acvm_batch_free(g); }
};

#define SHIM_CATCH(ret) catch (const std::exception &) { return ret; }

extern "C" {

const acvm_bb_solver_t *acvm_bb_stubbed(void) { return &STUBBED; }
const acvm_bb_solver_t *acvm_bb_dummy(void) { return &DUMMY; }

// ---------------------------------------------------------------------------------------------- struct ACVM, one instance
acvm_t *acvm_new(const acvm_circuit_t *c, const acvm_bb_solver_t *backend, const uint32_t *initial_ids, const uint8_t *values_be32,
                 uint32_t n_initial) try {
    auto a = std::make_unique<acvm_instance>();
    a->b = acvm_batch_new(c, backend, 1, initial_ids, n_initial);
    if (!a->b) return nullptr;
    if (acvm_batch_set_initial_witness(a->b, values_be32) != 0) return nullptr;
    for (uint32_t i = 0; i < n_initial; i++) a->initial[initial_ids[i]].assign(values_be32 + 32 * (size_t)i, values_be32 + 32 * (size_t)i + 32);
    return a.release();
} SHIM_CATCH(nullptr)
void acvm_free(acvm_t *a) { delete a; }

static int status_of(acvm_t *a) {
    acvm_result_t r;
    if (int rc = acvm_batch_results(a->b, &r)) return rc;
    return (int)r.status;
}
int acvm_solve(acvm_t *a) {
    if (!a) return ACVM_E_INVALID;
    const int rc = acvm_batch_solve(a->b);
    return rc < 0 ? rc : status_of(a);
}
int acvm_solve_opcode(acvm_t *a) {
    if (!a) return ACVM_E_INVALID;
    const int rc = acvm_batch_solve_opcode(a->b);
    return rc < 0 ? rc : status_of(a);
}
int acvm_get_status(acvm_t *a, acvm_result_t *out) {
    if (!a || !out) return ACVM_E_INVALID;
    // ACVM::new leaves the status InProgress (Solved for an empty circuit, mod.rs:147) before the first solve call
    const int rc = acvm_batch_results(a->b, out);
    if (rc == ACVM_E_STATE) {
        memset(out, 0, sizeof *out);
        acvm_stats_t st;
        if (int rc2 = acvm_batch_stats(a->b, &st)) return rc2;
        out->status = st.n_opcodes == 0 ? ACVM_STATUS_SOLVED : ACVM_STATUS_IN_PROGRESS;
        return 0;
    }
    return rc;  // any other failure (device, memory) leaves *out unwritten and is the caller's error
}
uint32_t acvm_instruction_pointer(acvm_t *a) {
    acvm_result_t r;
    if (!a || acvm_get_status(a, &r) != 0) return 0;
    if (r.status == ACVM_STATUS_SOLVED) {  // the pointer ran off the end of the opcodes
        acvm_stats_t st;
        return acvm_batch_stats(a->b, &st) == 0 ? st.n_opcodes : 0;
    }
    return r.opcode_index;
}
long long acvm_witness_map(acvm_t *a, uint32_t *ids, uint8_t *values_be32, uint32_t cap) try {
    if (!a) return ACVM_E_INVALID;
    acvm_result_t r;
    if (acvm_batch_results(a->b, &r) == ACVM_E_STATE) {  // nothing ran yet: the initial map (values as given, not reduced)
        long long n = 0;
        for (auto &kv : a->initial) {
            if ((uint64_t)n < cap) {
                if (ids) ids[n] = kv.first;
                if (values_be32) memcpy(values_be32 + 32 * (size_t)n, kv.second.data(), 32);
            }
            n++;
        }
        return n;
    }
    return map_pairs(a->b, 0, ids, values_be32, cap);
} SHIM_CATCH(ACVM_E_NOMEM)
long long acvm_finalize(acvm_t *a, uint32_t *ids, uint8_t *values_be32, uint32_t cap) try {
    if (!a) return ACVM_E_INVALID;
    acvm_result_t r;
    if (int rc = acvm_get_status(a, &r)) return rc;
    if (r.status != ACVM_STATUS_SOLVED) return ACVM_E_STATE;  // "ACVM is not ready to be finalized" (mod.rs:177-179 panics)
    return map_pairs(a->b, 0, ids, values_be32, cap);
} SHIM_CATCH(ACVM_E_NOMEM)
int acvm_get_pending_foreign_call(acvm_t *a, acvm_foreign_call_info_t *info) {
    return a ? acvm_batch_pending_foreign_call(a->b, 0, info) : ACVM_E_INVALID;
}
int acvm_pending_foreign_call_inputs(acvm_t *a, uint32_t *lens, uint8_t *values_be32) {
    return a ? acvm_batch_pending_foreign_call_inputs(a->b, 0, lens, values_be32) : ACVM_E_INVALID;
}
int acvm_resolve_pending_foreign_call(acvm_t *a, uint32_t n_values, const uint8_t *is_array, const uint32_t *lens, const uint8_t *values_be32) {
    return a ? acvm_batch_resolve_foreign_call(a->b, 0, n_values, is_array, lens, values_be32) : ACVM_E_INVALID;
}

// ---------------------------------------------------------------------------------------------- heterogeneous initial maps
acvm_multi_t *acvm_multi_new(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances, const uint64_t *offsets,
                             const uint32_t *ids, const uint8_t *values_be32) try {
    if (!c || (n_instances && !offsets)) return nullptr;
    auto m = std::make_unique<acvm_multi>();
    m->n = n_instances;
    m->nw = acvm_circuit_num_witnesses(c);
    m->group_of.resize(n_instances);
    m->index_in.resize(n_instances);
    // group key: the SORTED id set (a map has no order); each instance keeps its own permutation into that order
    std::map<std::vector<uint32_t>, uint32_t> key_to_group;
    std::vector<std::vector<uint32_t>> keys;
    for (uint32_t i = 0; i < n_instances; i++) {
        if (offsets[i + 1] < offsets[i]) return nullptr;
        std::vector<uint32_t> key(ids + offsets[i], ids + offsets[i + 1]);
        std::sort(key.begin(), key.end());
        key.erase(std::unique(key.begin(), key.end()), key.end());  // a map has every id once: a repeated id keeps its last value (scatter below)
        auto it = key_to_group.find(key);
        if (it == key_to_group.end()) {
            it = key_to_group.emplace(key, (uint32_t)keys.size()).first;
            keys.push_back(key);
            m->members.emplace_back();
        }
        m->group_of[i] = it->second;
        m->index_in[i] = (uint32_t)m->members[it->second].size();
        m->members[it->second].push_back(i);
    }
    m->groups.assign(keys.size(), nullptr);
    for (size_t g = 0; g < keys.size(); g++) {
        const std::vector<uint32_t> &key = keys[g];
        const std::vector<uint32_t> &mem = m->members[g];
        m->groups[g] = acvm_batch_new(c, solver, (uint32_t)mem.size(), key.data(), (uint32_t)key.size());
        if (!m->groups[g]) return nullptr;
        std::vector<uint8_t> vals(mem.size() * std::max<size_t>(key.size(), 1) * 32, 0);
        for (size_t t = 0; t < mem.size(); t++) {
            const uint32_t i = mem[t];
            for (uint64_t k = offsets[i]; k < offsets[i + 1]; k++) {  // a repeated id keeps its last value, like a map insert
                const size_t pos = (size_t)(std::lower_bound(key.begin(), key.end(), ids[k]) - key.begin());
                memcpy(&vals[(t * key.size() + pos) * 32], values_be32 + 32 * k, 32);
            }
        }
        if (acvm_batch_set_initial_witness(m->groups[g], vals.data()) != 0) return nullptr;
        acvm_stats_t st;
        if (acvm_batch_stats(m->groups[g], &st) == 0) m->nw = std::max(m->nw, st.n_witnesses);
    }
    return m.release();
} SHIM_CATCH(nullptr)
void acvm_multi_free(acvm_multi_t *m) { delete m; }
uint32_t acvm_multi_num_groups(const acvm_multi_t *m) { return m ? (uint32_t)m->groups.size() : 0; }
uint32_t acvm_multi_num_witnesses(const acvm_multi_t *m) { return m ? m->nw : 0; }

int acvm_multi_solve(acvm_multi_t *m) {
    if (!m) return ACVM_E_INVALID;
    int not_solved = 0;
    for (auto *g : m->groups) {
        const int rc = acvm_batch_solve(g);
        if (rc < 0) return rc;
        not_solved += rc;
    }
    return not_solved;
}
int acvm_multi_results(acvm_multi_t *m, acvm_result_t *out) try {
    if (!m || (m->n && !out)) return ACVM_E_INVALID;
    for (size_t g = 0; g < m->groups.size(); g++) {
        std::vector<acvm_result_t> res(m->members[g].size());
        if (int rc = acvm_batch_results(m->groups[g], res.data())) return rc;
        for (size_t t = 0; t < res.size(); t++) out[m->members[g][t]] = res[t];
    }
    return 0;
} SHIM_CATCH(ACVM_E_NOMEM)
int acvm_multi_witness_map(acvm_multi_t *m, uint32_t instance, uint8_t *assigned, uint8_t *values_be32) try {
    if (!m || instance >= m->n || !assigned || !values_be32) return ACVM_E_INVALID;
    acvm_batch_t *g = m->groups[m->group_of[instance]];
    acvm_stats_t st;
    if (int rc = acvm_batch_stats(g, &st)) return rc;
    // a group whose ids stay below the circuit's witnesses has a table of the circuit's size; pad to the common size
    std::vector<uint8_t> a(st.n_witnesses ? st.n_witnesses : 1), v((size_t)(st.n_witnesses ? st.n_witnesses : 1) * 32);
    if (int rc = acvm_batch_witness_map(g, m->index_in[instance], 1, a.data(), v.data())) return rc;
    memset(assigned, 0, m->nw);
    memset(values_be32, 0, (size_t)m->nw * 32);
    memcpy(assigned, a.data(), std::min(m->nw, st.n_witnesses));
    memcpy(values_be32, v.data(), (size_t)std::min(m->nw, st.n_witnesses) * 32);
    return 0;
} SHIM_CATCH(ACVM_E_NOMEM)
acvm_batch_t *acvm_multi_locate(acvm_multi_t *m, uint32_t instance, uint32_t *index_in_batch) {
    if (!m || instance >= m->n) return nullptr;
    if (index_in_batch) *index_in_batch = m->index_in[instance];
    return m->groups[m->group_of[instance]];
}

}  // extern "C"
