// plan.cpp -- static replay of the reference's assigned-set bookkeeping + levelisation (see plan.hpp).
#include "plan.hpp"
#include "gate_record.hpp"
#include "tuning.hpp"
#include "scratch_layout.hpp"
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <map>
#include <queue>
#include <stdexcept>
#include <unordered_map>

namespace acvm {
namespace {

// The circuit's constants (pool ids of the device table) and the memo of -1/c. One open-addressing table serves both: a 10^6-opcode circuit
// with distinct coefficients makes ~10^7 lookups, and two node-based maps spent 40 % of the planning time in cache misses.
struct ConstPool {
    std::vector<FrH> &pool;
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    struct Entry {
        FrH v, ninv;
        uint32_t id = NONE, inv_id = NONE;  // pool ids of v and of -1/v once something refers to them
        uint8_t has_inv = 0;                // 1: ninv = -1/v; 2: queued in prefill_neg_inv
    };
    std::vector<Entry> ents;
    std::vector<uint64_t> slots;  // tag << 32 | entry index + 1; 0 = empty
    size_t mask = 0;
    FrH one = frh::one(), minus_one = frh::neg(frh::one());
    explicit ConstPool(std::vector<FrH> &p) : pool(p) { rehash(1u << 12); }
    static uint64_t hash(const FrH &c) {
        uint64_t h = c.l[0] * 0x9E3779B97F4A7C15ULL ^ c.l[1] * 0xC2B2AE3D27D4EB4FULL ^ c.l[2] * 0x165667B19E3779F9ULL ^ c.l[3] * 0x27D4EB2F165667C5ULL;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ULL;
        return h ^ h >> 32;
    }
    void rehash(size_t n) {
        slots.assign(n, 0);
        mask = n - 1;
        for (size_t e = 0; e < ents.size(); e++) {
            const uint64_t h = hash(ents[e].v);
            size_t i = h & mask;
            while (slots[i]) i = (i + 1) & mask;
            slots[i] = (h >> 32) << 32 | (uint64_t)(e + 1);
        }
    }
    uint32_t entry(const FrH &c) {  // index of c's entry, created on first sight
        const uint64_t h = hash(c), tag = h >> 32;
        size_t i = h & mask;
        for (; slots[i]; i = (i + 1) & mask)
            if (slots[i] >> 32 == tag && ents[(uint32_t)slots[i] - 1].v == c) return (uint32_t)slots[i] - 1;
        if (ents.size() >= 0xFFFFFFF0u) throw std::length_error("too many constants");
        Entry e;
        e.v = c;
        e.ninv = frh::zero();
        ents.push_back(e);
        slots[i] = tag << 32 | (uint64_t)ents.size();
        if (ents.size() * 2 > slots.size()) rehash(slots.size() * 2);
        return (uint32_t)ents.size() - 1;
    }
    uint32_t id_of(uint32_t e) {
        if (ents[e].id == NONE) {
            ents[e].id = (uint32_t)pool.size();
            pool.push_back(ents[e].v);
        }
        return ents[e].id;
    }
    uint32_t coef(const FrH &c) {  // multiplicative coefficient
        if (c == one) return COEF_ONE;
        if (c == minus_one) return COEF_MINUS_ONE;
        return intern(c);
    }
    uint32_t coef_or_zero(const FrH &c) { return c.is_zero() ? COEF_ZERO : coef(c); }
    uint32_t constant(const FrH &c) {  // additive constant
        if (c.is_zero()) return COEF_ZERO;
        return intern(c);
    }
    uint32_t intern(const FrH &c) { return id_of(entry(c)); }
    // -1/c for a whole set of coefficients with ONE field inversion (Montgomery's trick): the planner needs -1/c of every
    // coefficient of every Arithmetic opcode (gate folding, and the exact kernels' records), and a host inversion costs
    // as much as 300 products
    void prefill_neg_inv(const std::vector<FrH> &coefs) {
        std::vector<uint32_t> todo;
        std::vector<FrH> prefix;
        for (const FrH &c : coefs) {
            if (c.is_zero() || c == one || c == minus_one) continue;
            const uint32_t e = entry(c);
            if (!ents[e].has_inv) { ents[e].has_inv = 2; todo.push_back(e); }
        }
        if (todo.empty()) return;
        prefix.resize(todo.size());
        FrH acc = frh::one();
        for (size_t i = 0; i < todo.size(); i++) { acc = frh::mul(acc, ents[todo[i]].v); prefix[i] = acc; }
        FrH inv = frh::inverse(acc);
        for (size_t i = todo.size(); i-- > 0;) {
            const FrH inv_i = i ? frh::mul(inv, prefix[i - 1]) : inv;
            inv = frh::mul(inv, ents[todo[i]].v);
            ents[todo[i]].ninv = frh::neg(inv_i);
            ents[todo[i]].has_inv = 1;
        }
    }
    uint32_t entry_with_inv(const FrH &c) {
        const uint32_t e = entry(c);
        if (ents[e].has_inv != 1) {
            ents[e].ninv = frh::neg(frh::inverse(c));
            ents[e].has_inv = 1;
        }
        return e;
    }
    // -1/c, memoised (real circuits repeat a handful of coefficients)
    FrH neg_inv(const FrH &c) {
        if (c == one) return minus_one;
        if (c == minus_one) return one;
        return ents[entry_with_inv(c)].ninv;
    }
    // (coefficient id, id of -1/coefficient) of a linear term of an in-order record, c != 0: one lookup for both
    void lin_ids(const FrH &c, uint32_t &cid, uint32_t &iid) {
        if (c == one) { cid = COEF_ONE; iid = COEF_MINUS_ONE; return; }
        if (c == minus_one) { cid = COEF_MINUS_ONE; iid = COEF_ONE; return; }
        const uint32_t e = entry_with_inv(c);
        cid = id_of(e);
        if (ents[e].inv_id == NONE) {
            const FrH ni = ents[e].ninv;  // (coef() may grow `ents`)
            const uint32_t id = coef(ni);
            ents[e].inv_id = id;
        }
        iid = ents[e].inv_id;
    }
};

struct PendingGate {
    uint32_t level;
    std::vector<uint32_t> words;
    std::vector<uint32_t> reads;     // distinct operand witnesses
    uint32_t n_tails = 0;            // gates fused behind this one (they run in the same wave and read this output from registers)
    size_t last_w0 = 0;              // position in `words` of the header of the last record of the chain
    bool fused = false;              // this gate runs as the tail of another one
    uint32_t run_level = 0;          // the level whose launch executes it (its wave's host's level)
    uint32_t owner = 0, last_gate = 0;  // host only: the gate whose output the wave's `local` registers hold at the end of the program, and the last record
};
// (GATE_TAIL_FLAG, GATE_SETLOCAL_FLAG, GATE_LOCAL, the output modes and the bound units: gate_record.hpp)
// operand words of a gate record (layout in build_plan)
template <class F>
static void for_each_operand_word(std::vector<uint32_t> &w, F fn) {
    const uint32_t np_mac = (w[0] >> 8) & 0xff, nl_mac = (w[0] >> 16) & 0xff;
    const uint32_t np_pos = w[5] & 0xff, np_neg = (w[5] >> 8) & 0xff, nl_pos = (w[5] >> 16) & 0xff, nl_neg = w[5] >> 24;
    size_t t = 6;
    for (uint32_t i = 0; i < np_mac; i++, t += 10) { fn(w[t + 8]); fn(w[t + 9]); }
    for (uint32_t i = 0; i < nl_mac; i++, t += 9) fn(w[t + 8]);
    for (uint32_t i = 0; i < 2 * (np_pos + np_neg) + nl_pos + nl_neg; i++, t++) fn(w[t]);
}
struct PendingRecord {
    uint32_t level, cls, opcode;
    std::vector<uint32_t> reads;  // witnesses it reads (compared outputs included)
    bool synthetic = false;       // a level-schedule record without an opcode of its own (digest leaves, merged RANGE checks): `opcode` is its offset in prog
    uint32_t prog_at = 0xFFFFFFFFu;  // the level schedule runs this copy of the opcode's record instead (a hash record extended by fused RANGE checks)
    bool chained = false;            // runs behind another hash record in that record's workgroup (hash chains): not in the level lists
};
// inversion of a SOLVE_DYN gate's denominator: runs beside level `level`, its result is read by gate `gate` at `use_level`
struct PendingInverse {
    uint32_t level, use_level, partner, opcode, gate;
};
// Denominators are inverted ahead of their gates, in batches: every inv_epoch-th level (tuning.hpp, default 4) the inversion kernel
// takes all denominators that became known since the last batch, so that one field inversion (Montgomery's trick) is shared by all of
// them and its latency is off the path of the levels in between. A gate whose denominator is younger than the last batch
// waits for the next one (at most inv_epoch levels).
//
// Records of the heavy classes (hashes, Grumpkin / Pedersen, ECDSA, Brillig: bound by the integer pipe or by latency) run on their
// own streams beside the levels of the main stream. Two scheduling modes exist for circuits whose main stream stalls on them:
//   * heavy_epoch = K: heavy records are launched only every K-th level, all that became ready since the last batch together
//     (like the inversion batches): fewer, fatter launches;
//   * heavy_latency = D (pedersen_latency for Pedersen alone): the main stream may read the outputs of the heavy batch of level L from
//     level L + D + 1 on, i.e. the planner prices a heavy batch at D levels of main-stream work and puts the consumers behind it
//     instead of letting the whole level wait. Heavy records that read heavy outputs (same stream, in order) only need a later batch.
// Measured on the config-5 mix at 250 k opcodes, tile of 4 096 instances (profiles/r02_config5_schedule.txt): when only a few records
// read heavy outputs (round 1's generator) K = D = 4 takes a tile from 46.5 to 43.5 ms; when every tenth witness is a heavy output
// and gates read them everywhere (the SURVEY 8d generator) the same setting stretches the DAG from 180 to 529 levels and the tile
// from 54 to 69 ms. The default is therefore K = 1, D = 0: every record at its earliest level.

// Expression record: [n_mul, n_lin, qc, (coef, l, r) x n_mul, (coef, -1/coef, w) x n_lin]
void emit_expr(std::vector<uint32_t> &s, ConstPool &pool, const Expr &e) {
    s.push_back((uint32_t)e.mul.size());
    s.push_back((uint32_t)e.lin.size());
    s.push_back(pool.constant(e.qc));
    for (auto &t : e.mul) {
        s.push_back(pool.coef_or_zero(t.c));
        s.push_back(t.l);
        s.push_back(t.r);
    }
    for (auto &t : e.lin) {
        uint32_t cid = COEF_ZERO, iid = COEF_ZERO;
        if (!t.c.is_zero()) pool.lin_ids(t.c, cid, iid);
        s.push_back(cid);
        s.push_back(iid);
        s.push_back(t.w);
    }
}

// generic-instance view of an expression: every witness it mentions must be known (get_value, pwg/mod.rs:321-332)
struct Reads {
    const std::vector<uint8_t> &known;
    const std::vector<uint32_t> &level;
    std::vector<uint32_t> ws;
    uint32_t lvl = 0;
    bool ok = true;
    Reads(const std::vector<uint8_t> &k, const std::vector<uint32_t> &l) : known(k), level(l) {}
    void witness(uint32_t w) {
        if (w >= known.size() || !known[w]) { ok = false; return; }
        lvl = std::max(lvl, level[w]);
        ws.push_back(w);
    }
    void expr(const Expr &e) {
        for (auto &t : e.mul) { witness(t.l); witness(t.r); }
        for (auto &t : e.lin) witness(t.w);
    }
    size_t distinct() {
        std::sort(ws.begin(), ws.end());
        ws.erase(std::unique(ws.begin(), ws.end()), ws.end());
        return ws.size();
    }
};

uint32_t clamp_reg(uint64_t r) { return r > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)r; }
// 2^bits mod p (Brillig's BinaryIntOp::Sub at a bit size beyond 256: brillig_vm/src/arithmetic.rs:34, ops_light.hpp int_op_core)
FrH pow2_mod_p(uint32_t bits) {
    FrH base = frh::from_u64(2), acc = frh::one();
    for (uint32_t e = bits; e; e >>= 1) {
        if (e & 1) acc = frh::mul(acc, base);
        base = frh::mul(base, base);
    }
    return acc;
}

// =========================================================================== straight-line Brillig
// The stdlib's integer fallbacks (stdlib/src/blackbox_fallbacks/uint.rs:212-260) and most compiler-generated helper calls (inversion and
// comparison hints) are Brillig programs of a few instructions without loops. Running them in the VM kernel costs a launch of a 298-VGPR
// interpreter per level (profiles/r02v_config5_1m.json: 51-58 ms per tile for 0.5 % of the opcodes). Such an opcode -- single-value inputs
// and outputs; bytecode made of BinaryFieldOp / BinaryIntOp / Const / Mov / Stop / Trap and jumps that only go FORWARD; at most SL_REGS
// registers -- gets a second record, [PK_BRILLIG_SL, ...] (ops_light.hpp op_brillig_sl), which the LEVEL schedule runs on the main stream
// beside the gates. The exact path keeps the VM record, so every failure shape (panicking integer op, trap, output conflict, missing
// input) is still reported by the VM in the reference's words.
constexpr uint32_t SL_REGS = 4, SL_MAX_INS = 255;
void emit_straight_line(const BrilligCall &b, uint32_t oi, ConstPool &pool, std::vector<uint32_t> &out, std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> &sl_of) {
    const size_t n = b.bytecode.size();
    if (n == 0 || n > SL_MAX_INS || b.inputs.size() > SL_REGS || b.outputs.size() > SL_REGS) return;
    for (auto &in : b.inputs) if (in.is_array) return;
    for (auto &ot : b.outputs) if (ot.is_array) return;
    // registers: inputs and outputs keep their index; any other register moves to a free slot
    const uint64_t fixed = std::max(b.inputs.size(), b.outputs.size());
    std::vector<uint64_t> used;
    auto use = [&](uint64_t r) { if (r >= fixed && std::find(used.begin(), used.end(), r) == used.end()) used.push_back(r); };
    for (size_t k = 0; k < n; k++) {
        const BrilligOp &op = b.bytecode[k];
        switch (op.op) {
        case BR_BINARY_INT_OP: case BR_BINARY_FIELD_OP: use(op.a); use(op.b); use(op.c); break;
        case BR_CONST: use(op.a); break;
        case BR_MOV: use(op.a); use(op.b); break;
        case BR_JUMP_IF: case BR_JUMP_IF_NOT:
            use(op.a);
            [[fallthrough]];
        case BR_JUMP:
            if (op.location <= k) return;  // a loop (or a jump onto itself): the VM runs it
            break;
        case BR_STOP: case BR_TRAP: break;
        default: return;  // calls, memory, foreign calls, black boxes
        }
    }
    if (fixed + used.size() > SL_REGS) return;
    for (uint64_t r : used) if (r >= 65536) return;  // (register index past the VM's maximum: a panic of the reference)
    auto slot = [&](uint64_t r) -> uint32_t { return r < fixed ? (uint32_t)r : (uint32_t)(fixed + (std::find(used.begin(), used.end(), r) - used.begin())); };
    const uint32_t at = (uint32_t)out.size();
    out.insert(out.end(), {PK_BRILLIG_SL, oi, b.has_predicate ? 1u : 0u, (uint32_t)b.inputs.size(), (uint32_t)b.outputs.size(), (uint32_t)n});
    if (b.has_predicate) emit_expr(out, pool, b.predicate);
    for (auto &in : b.inputs) emit_expr(out, pool, in.single);
    const uint32_t flags_at = (uint32_t)out.size() + 1 - at;
    for (auto &ot : b.outputs) { out.push_back(ot.w); out.push_back(0); }
    for (size_t k = 0; k < n; k++) {
        const BrilligOp &op = b.bytecode[k];
        uint32_t kind = 0, sub = 0, dst = 0, ra = 0, rb = 0, x = 0, cidx = 0;
        switch (op.op) {
        case BR_BINARY_FIELD_OP: case BR_BINARY_INT_OP:
            kind = op.op == BR_BINARY_INT_OP ? 1u : 0u; sub = op.sub_op; dst = slot(op.a); ra = slot(op.b); rb = slot(op.c); x = op.bit_size;
            if (kind == 1 && sub == 1 && op.bit_size > 256) cidx = pool.intern(pow2_mod_p(op.bit_size));  // a wide Sub: 2^bits mod p
            break;
        case BR_CONST: kind = 2; dst = slot(op.a); cidx = pool.intern(op.value); break;
        case BR_MOV: kind = 3; dst = slot(op.a); ra = slot(op.b); break;
        case BR_JUMP: kind = 4; x = (uint32_t)std::min<uint64_t>(op.location, 0xFFFFFFFEull); break;
        case BR_JUMP_IF: case BR_JUMP_IF_NOT: kind = op.op == BR_JUMP_IF ? 5u : 6u; ra = slot(op.a); x = (uint32_t)std::min<uint64_t>(op.location, 0xFFFFFFFEull); break;
        case BR_STOP: kind = 7; break;
        default: kind = 8; break;  // Trap
        }
        out.insert(out.end(), {kind | sub << 8 | dst << 16 | ra << 20 | rb << 24, x, cidx});
    }
    sl_of[oi] = {at, flags_at};
}


// =========================================================================== placement of a level's records inside their class's list
// The records of one level are independent, so their order inside a class's list is a launch-placement choice (schedule.cpp
// layout_launches cuts the lists into launches along these groups):
//   * CLS_HASH: byte-message records first -- short messages (<= 256 bytes: 16 KiB of LDS per 64 instances), then long ones (the LDS message
//     of a launch is sized by its longest record: a 1 024-byte message takes 77 KiB of the workgroup's 160 KiB on gfx950; one long message
//     somewhere in the circuit must not cost every 64-byte SHA record its occupancy) -- then the records of the scratch-carrying kernel;
//   * CLS_GRUMPKIN: the longest records first (SchnorrVerify ~1.7 ms of one wave per SIMD, FixedBaseScalarMul 0.3): workgroups are placed in
//     grid order, and at ~240 registers a SIMD holds two of these waves -- a long wave that arrives last waits for a slot behind short ones;
//   * CLS_LIGHT, CLS_BRILLIG: straight-line Brillig records last: they have a kernel of their own (kernels_ops.hip LightSlOp; on the Brillig lane
//     with tuning sl_lane, on the main stream without).
void order_level_records(Plan &p) {
    const std::vector<uint32_t> &pg = p.prog;
    for (size_t L = 0; L < p.n_levels; L++) {
        {
            const int k = CLS_HASH;
            const uint32_t lo = p.cls_level_start[k][L], hi = p.cls_level_start[k][L + 1];
            std::vector<std::pair<uint32_t, uint32_t>> recs;  // (offset, scratch)
            for (int pass = 0; pass < 3 && hi > lo; pass++)
                for (uint32_t r = lo; r < hi; r++)
                    if (hash_launch_group(pg, p.cls_offset[k][r]) == pass) recs.push_back({p.cls_offset[k][r], p.cls_scratch[k][r]});
            for (uint32_t r = lo; r < hi; r++) { p.cls_offset[k][r] = recs[r - lo].first; p.cls_scratch[k][r] = recs[r - lo].second; }
        }
        {
            const int k = CLS_GRUMPKIN;
            const uint32_t lo = p.cls_level_start[k][L], hi = p.cls_level_start[k][L + 1];
            if (hi - lo > 1) {
                std::vector<std::pair<uint32_t, uint32_t>> recs;
                for (uint32_t r = lo; r < hi; r++) recs.push_back({p.cls_offset[k][r], p.cls_scratch[k][r]});
                std::stable_sort(recs.begin(), recs.end(), [&](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) {
                    auto rank = [&](uint32_t off) { const uint32_t kind = pg[off]; return kind == PK_SCHNORR ? 0 : kind == PK_PEDERSEN ? 1 : 2; };
                    return rank(x.first) < rank(y.first);
                });
                for (uint32_t r = lo; r < hi; r++) { p.cls_offset[k][r] = recs[r - lo].first; p.cls_scratch[k][r] = recs[r - lo].second; }
            }
        }
        for (const int k : {(int)CLS_LIGHT, (int)CLS_BRILLIG}) {
            const uint32_t lo = p.cls_level_start[k][L], hi = p.cls_level_start[k][L + 1];
            std::vector<std::pair<uint32_t, uint32_t>> recs;
            for (int pass = 0; pass < 2 && hi > lo; pass++)
                for (uint32_t r = lo; r < hi; r++)
                    if ((pg[p.cls_offset[k][r]] == PK_BRILLIG_SL) == (pass == 1)) recs.push_back({p.cls_offset[k][r], p.cls_scratch[k][r]});
            for (uint32_t r = lo; r < hi; r++) { p.cls_offset[k][r] = recs[r - lo].first; p.cls_scratch[k][r] = recs[r - lo].second; }
        }
    }
}

// memory blocks: cell ranges of the per-instance memory table + the program-order chain level
// level = level of the last write (MemoryInit included), rlevel = latest level of a read since that write: reads of a block
// between two writes have no order among themselves and may share a level; a write comes after every earlier access
struct Block { uint32_t base = 0, cap = 0, len = 0, readable = 0, level = 0, rlevel = 0; bool seen = false; };
inline bool is_heavy(uint32_t cls) { return cls == CLS_HASH || cls == CLS_GRUMPKIN || cls == CLS_BRILLIG || cls == CLS_PEDERSEN || cls == CLS_ECDSA || cls == CLS_DIGEST; }

// The planner: one object per build_plan call, one method per PASS, in the order build_plan (at the end of this file) runs them. What a pass
// leaves behind for the later ones is a member -- the intermediate state is explicit -- and what each pass promises is checked by
// tests/test_plan_host.py pass by pass (acvm_debug_plan_passes) and, end to end, by the hazard checker over the schedule (schedule_check.cpp);
// tools/plan_fingerprint.py asserts that a change here moved no word of any plan.
struct Planner {
    const Circuit &c;
    const uint32_t *initial_ids;
    const uint32_t n_initial;
    const PlanOpts &opts;
    const bool host_blackbox;
    const Tuning tune;  // one snapshot for the whole plan
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    Plan p;
    uint32_t nw = 0;
    // ---- pass 1 (init) -> everybody
    std::vector<uint8_t> known;        // the generic instance's assigned set, in program order
    std::vector<uint32_t> level;       // per witness: the level from which the MAIN stream may read it
    ConstPool pool;
    std::map<uint32_t, Block> blocks;
    // ---- pass 2 (in-order program) -> replay
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> out_slots;  // per opcode: (position of flag word, witness); the flags are patched by the replay
    std::vector<uint32_t> sl_prog;     // straight-line Brillig: the light records of the eligible opcodes, appended to `prog` behind the in-order program
    std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> sl_of;  // opcode -> (offset in sl_prog, position of its first output flag word)
    uint32_t sl_base = 0;
    // ---- pass 3 (pins) + 4 (replay: levels, folded gates, records) -> the schedule passes
    std::vector<PendingGate> gates;
    std::vector<PendingRecord> records;
    std::vector<PendingInverse> inverses;
    std::vector<uint32_t> heavy_level;  // witnesses produced by a record of a heavy class (those run on their own stream, batch.cpp): level of the record, else 0
    std::vector<uint8_t> wlane;         // 1 + heavy lane of the record that produces w, 0 = not a heavy output
    // hlevel[w]: the level from which the HEAVY stream may read w (level[w] is the main stream's; they differ for the outputs of
    // heavy records: the heavy stream is in order, the main stream sees them HEAVY_LATENCY levels later)
    std::vector<uint32_t> hlevel;
    uint32_t K_heavy = 1, D_heavy = 0, D_pedersen = 0, K_pedersen = 1;
    std::vector<std::pair<uint32_t, uint32_t>> heavy_reads;  // (level of a main-stream record, witness of a heavy record it reads)
    uint32_t out_latency = 0;  // levels of slack of the record whose outputs are being assigned
    uint8_t out_lane = 0;
    std::vector<uint8_t> pinned, is_scaled;
    bool scaling_on = false, relax_on = false;
    std::vector<FrH> ws, wsi;  // scale and 1 / scale of the scaled witnesses (indexed through scale_slot)
    std::vector<uint32_t> scale_slot, gate_of;  // gate_of[w]: the gate that writes w
    const FrH f_one = frh::one(), f_minus_one = frh::neg(frh::one());
    // ---- the schedule passes
    std::vector<uint32_t> wdef;  // wdef[w]: the level whose launches write w (a fused gate writes in its host's wave; 0 = initial witness)
    uint32_t max_level = 0;

    Planner(const Circuit &circ, const uint32_t *ids, uint32_t n_ids, const PlanOpts &o)
        : c(circ), initial_ids(ids), n_initial(n_ids), opts(o), host_blackbox(o.host_blackbox), tune(tuning()), pool(p.constants) {}

    void unsupported(uint32_t oi, const std::string &what) {
        if (p.unsupported.empty()) p.unsupported = "opcode " + std::to_string(oi) + ": " + what + " has no kernel";
    }
    Plan finish() {
        p.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return std::move(p);
    }

    // =========================================================================== pass 1: the witness table, the initial set, the constants, the memory blocks
    bool init() {
    // The witness table is dense (slot = witness index); the reference's BTreeMap takes any u32 index. Circuit bytes are
    // untrusted input: an index near 2^32 must neither wrap `max + 1` nor make the planner allocate per-witness vectors of
    // many GB, so indices above PLAN_MAX_WITNESSES - 1 are refused (ACVM_E_UNSUPPORTED at acvm_batch_new / plan_stats).
    uint64_t nw64 = (uint64_t)c.max_witness + 1;
    for (uint32_t i = 0; i < n_initial; i++) nw64 = std::max<uint64_t>(nw64, (uint64_t)initial_ids[i] + 1);
    if (nw64 > PLAN_MAX_WITNESSES) {
        p.unsupported = "witness index " + std::to_string(nw64 - 1) + " exceeds the dense witness table (at most " +
                        std::to_string(PLAN_MAX_WITNESSES) + " witnesses per circuit)";
        return false;
    }
    nw = (uint32_t)nw64;
    p.n_witnesses = nw;
    p.initial_ids.assign(initial_ids, initial_ids + n_initial);
    p.producer.assign(nw, 0xFFFFFFFFu);
    known.assign(nw, 0);
    level.assign(nw, 0);
    for (uint32_t i = 0; i < n_initial; i++) {
        known[initial_ids[i]] = 1;
        p.producer[initial_ids[i]] = 0xFFFFFFFEu;
    }
    {
        std::vector<FrH> coefs;
        for (const Opcode &o : c.opcodes)
            if (o.kind == OP_ARITHMETIC) {
                for (auto &t : o.expr.mul) coefs.push_back(t.c);
                for (auto &t : o.expr.lin) coefs.push_back(t.c);
            }
        pool.prefill_neg_inv(coefs);
    }
    gates.reserve(c.opcodes.size());

    for (auto &o : c.opcodes)
        if (o.kind == OP_MEMORY_INIT) {
            Block &b = blocks[o.block_id];
            b.cap = std::max(b.cap, (uint32_t)o.init.size());
        } else if (o.kind == OP_MEMORY_OP) blocks[o.block_id];
    for (auto &kv : blocks) {
        kv.second.base = p.mem_cells;
        p.mem_cells += kv.second.cap;
    }
        return true;
    }

    // one record per opcode, original expressions; false: an opcode no kernel implements (p.unsupported says which)
    bool emit_in_order_program() {
    // =========================================================================== in-order program (all opcodes)
    p.prog_offset.reserve(c.opcodes.size());
    p.prog_class.assign(c.opcodes.size(), CLS_LIGHT);
    p.prog_scratch.assign(c.opcodes.size(), 0);
    // the `flag` words (was the output already assigned for the generic instance?) are patched in the second pass
    out_slots.assign(c.opcodes.size(), {});  // (position of flag word, witness)
    // straight-line Brillig (below): the light records of the eligible opcodes, appended to `prog` behind the in-order program
    {
        std::map<uint32_t, Block> st = blocks;  // running len / readable per block in program order
        for (auto &kv : st) { kv.second.len = 0; kv.second.readable = 0; }
        for (uint32_t oi = 0; oi < c.opcodes.size(); oi++) {
            const Opcode &o = c.opcodes[oi];
            auto &s = p.prog;
            p.prog_offset.push_back((uint32_t)s.size());
            auto out = [&](uint32_t w) {
                s.push_back(w);
                out_slots[oi].push_back({(uint32_t)s.size(), w});
                s.push_back(0);
            };
            switch (o.kind) {
            case OP_ARITHMETIC:
                s.push_back(PK_ARITH); s.push_back(oi);
                emit_expr(s, pool, o.expr);
                break;
            case OP_BLACKBOX: {
                const BlackBoxCall &b = *o.bb;
                switch (b.func) {
                case BB_RANGE:
                    s.insert(s.end(), {PK_RANGE, oi, b.in[0][0].witness, b.in[0][0].num_bits});
                    break;
                case BB_AND: case BB_XOR:
                    s.insert(s.end(), {PK_LOGIC, oi, b.func == BB_XOR ? 1u : 0u, b.in[0][0].witness, b.in[1][0].witness,
                                       b.in[0][0].num_bits, b.in[1][0].num_bits});
                    out(b.out[0]);
                    break;
                case BB_SHA256: case BB_BLAKE2S: case BB_KECCAK256: case BB_KECCAK256_VAR: case BB_HASH_TO_FIELD_128: {
                    // [PK_HASH, oi, func (| PLAN_HASH_COOP_FLAG), n_in, n_out, var_w (or NONE), (w, num_bits) x n_in, (out, flag) x n_out]
                    p.prog_class[oi] = CLS_HASH;
                    uint64_t bytes = 0;
                    for (auto &in : b.in[0]) bytes += std::min<uint32_t>((in.num_bits + 7) / 8, 32);
                    if (bytes > (1u << 24)) { unsupported(oi, "hash input above 16 MiB"); }
                    p.prog_scratch[oi] = (uint32_t)((bytes + 3) / 4 + 1);
                    // byte messages (every input one byte wide) of the three plain hashes: the level kernel unpacks them with four
                    // waves per 64 instances through LDS (kernels_hash.hip hash_coop_level_kernel); bit 8 of the function word says so
                    bool coop = (b.func == BB_SHA256 || b.func == BB_BLAKE2S || b.func == BB_KECCAK256) && b.out.size() == 32 &&
                                !b.in[0].empty() && b.in[0].size() <= PLAN_HASH_COOP_MAX_BYTES;
                    for (auto &in : b.in[0]) coop = coop && in.num_bits >= 1 && in.num_bits <= 8;
                    s.insert(s.end(), {PK_HASH, oi, b.func | (coop ? PLAN_HASH_COOP_FLAG : 0u), (uint32_t)b.in[0].size(), (uint32_t)b.out.size(),
                                       b.func == BB_KECCAK256_VAR ? b.in[1][0].witness : 0xFFFFFFFFu});
                    for (auto &in : b.in[0]) { s.push_back(in.witness); s.push_back(in.num_bits); }
                    for (uint32_t w : b.out) out(w);
                    if (coop && tune.byte_plane)  // its inputs that are initial witnesses: a byte plane each (plan.hpp)
                        for (auto &in : b.in[0])
                            if (p.producer[in.witness] == 0xFFFFFFFEu) {
                                if (p.byte_plane_of.empty()) p.byte_plane_of.assign(nw, 0xFFFFFFFFu);
                                if (p.byte_plane_of[in.witness] == 0xFFFFFFFFu) p.byte_plane_of[in.witness] = p.n_byte_planes++;
                                p.n_byte_plane_reads++;
                            }
                    break;
                }
                case BB_PEDERSEN:
                    // [PK_PEDERSEN, oi, domain_separator, n_in, out_x, fx, out_y, fy, ws..., seed row]
                    p.prog_class[oi] = host_blackbox ? CLS_HOSTBB : CLS_GRUMPKIN;
                    p.needs_grumpkin |= !host_blackbox;
                    s.insert(s.end(), {PK_PEDERSEN, oi, b.domain_separator, (uint32_t)b.in[0].size()});
                    out(b.out[0]); out(b.out[1]);
                    for (auto &in : b.in[0]) s.push_back(in.witness);
                    // trailing word: row of the seed table (batch.cpp): the first hash_pair of the chain, hash_pair(IV, n), does
                    // not depend on the instance and is computed once per batch for the level kernel
                    s.push_back((uint32_t)p.pedersen_seeds.size());
                    p.pedersen_seeds.push_back({(uint32_t)b.in[0].size(), b.domain_separator});
                    break;
                case BB_FIXED_BASE_SCALAR_MUL:
                    p.prog_class[oi] = host_blackbox ? CLS_HOSTBB : CLS_GRUMPKIN;
                    p.needs_grumpkin |= !host_blackbox;
                    s.insert(s.end(), {PK_FIXED_BASE, oi, b.in[0][0].witness, b.in[1][0].witness});
                    out(b.out[0]); out(b.out[1]);
                    break;
                case BB_SCHNORR_VERIFY: {
                    // [PK_SCHNORR, oi, pkx, pky, n_sig, n_msg, out, flag, sig ws..., msg ws...]
                    p.prog_class[oi] = host_blackbox ? CLS_HOSTBB : CLS_GRUMPKIN;
                    p.needs_grumpkin |= !host_blackbox;
                    // challenge preimage (32 + message bytes) + the per-lane window table of e * pk (15 Jacobian points x 27 words)
                    p.prog_scratch[oi] = (uint32_t)((32 + b.in[3].size() + 3) / 4 + 1) + GRUMPKIN_VARBASE_SCRATCH_WORDS;  // message + the window table of e * pk (ops_grumpkin.hpp)
                    s.insert(s.end(), {PK_SCHNORR, oi, b.in[0][0].witness, b.in[1][0].witness, (uint32_t)b.in[2].size(),
                                       (uint32_t)b.in[3].size()});
                    out(b.out[0]);
                    for (auto &in : b.in[2]) s.push_back(in.witness);
                    for (auto &in : b.in[3]) s.push_back(in.witness);
                    break;
                }
                case BB_RECURSIVE_AGGREGATION: {
                    // [PK_ZERO_OUT, oi, n_in, n_out, in ws..., (out, flag)...]; inputs in get_inputs_vec order
                    // (black_box_function_call.rs:262-290: vk, proof, public_inputs, key_hash, input_aggregation_object?)
                    std::vector<uint32_t> ins;
                    for (int g = 0; g < 4; g++)
                        for (auto &in : b.in[g]) ins.push_back(in.witness);
                    if (b.has_in_agg)
                        for (auto &in : b.in_agg) ins.push_back(in.witness);
                    s.insert(s.end(), {PK_ZERO_OUT, oi, (uint32_t)ins.size(), (uint32_t)b.out.size()});
                    s.insert(s.end(), ins.begin(), ins.end());
                    for (uint32_t w : b.out) out(w);
                    break;
                }
                case BB_ECDSA_SECP256K1: case BB_ECDSA_SECP256R1:
                    // [PK_ECDSA, oi, curve, n_x, n_y, n_sig, n_msg, out, flag, ws: x..., y..., sig..., msg...]
                    p.prog_class[oi] = CLS_ECDSA;
                    p.needs_ecdsa = true;
                    s.insert(s.end(), {PK_ECDSA, oi, b.func == BB_ECDSA_SECP256R1 ? 1u : 0u, (uint32_t)b.in[0].size(), (uint32_t)b.in[1].size(),
                                       (uint32_t)b.in[2].size(), (uint32_t)b.in[3].size()});
                    out(b.out[0]);
                    for (int g = 0; g < 4; g++)
                        for (auto &in : b.in[g]) s.push_back(in.witness);
                    break;
                default:
                    unsupported(oi, "black box function tag " + std::to_string(b.func));
                    s.insert(s.end(), {0xFFFFFFFFu, oi});
                }
                break;
            }
            case OP_DIRECTIVE: {
                const Directive &d = *o.dir;
                if (d.kind == DIR_QUOTIENT) {
                    // [PK_QUOTIENT, oi, q, fq, r, fr, has_pred, E(a), E(b), E(pred)?]
                    s.insert(s.end(), {PK_QUOTIENT, oi});
                    out(d.q); out(d.r);
                    s.push_back(d.has_predicate ? 1u : 0u);
                    emit_expr(s, pool, d.a);
                    emit_expr(s, pool, d.b);
                    if (d.has_predicate) emit_expr(s, pool, d.predicate);
                } else if (d.kind == DIR_TO_LE_RADIX) {
                    // [PK_TO_LE_RADIX, oi, radix, n_out, (out, flag) x n_out, E(a)]
                    s.insert(s.end(), {PK_TO_LE_RADIX, oi, d.radix, (uint32_t)d.bw.size()});
                    for (uint32_t w : d.bw) out(w);
                    emit_expr(s, pool, d.a);
                } else {
                    // [PK_PERM_SORT, oi, n, tuple, n_sort_by, n_bits, sort_by..., (bit witness, flag)..., E x (n * tuple)]
                    // (runs in the scratch-carrying hash class kernel; see ops_sort.hpp for the scratch map)
                    const uint32_t n = (uint32_t)d.sort_inputs.size();
                    bool shape_ok = n < (1u << 16);
                    for (auto &el : d.sort_inputs) shape_ok &= el.size() == d.tuple;
                    if (!shape_ok) unsupported(oi, "Directive::PermutationSort with a malformed tuple shape");
                    p.prog_class[oi] = CLS_HASH;
                    uint32_t lg = 0;
                    while ((1u << lg) < n) lg++;
                    p.prog_scratch[oi] = 8 * n * d.tuple + 4 * n + n * (lg + 1) + 1 + 3 * 72 + (4 * n + 8) * (lg + 2) + 64;
                    s.insert(s.end(), {PK_PERM_SORT, oi, n, d.tuple, (uint32_t)d.sort_by.size(), (uint32_t)d.bw.size()});
                    s.insert(s.end(), d.sort_by.begin(), d.sort_by.end());
                    for (uint32_t w : d.bw) out(w);
                    for (auto &el : d.sort_inputs)
                        for (auto &ex : el) emit_expr(s, pool, ex);
                }
                break;
            }
            case OP_MEMORY_INIT: {
                // [PK_MEM_INIT, oi, cell_base, len, ws...]
                Block &b = st[o.block_id];
                b.len = (uint32_t)o.init.size();
                b.readable = std::max(b.readable, b.len);
                s.insert(s.end(), {PK_MEM_INIT, oi, b.base, b.len});
                s.insert(s.end(), o.init.begin(), o.init.end());
                break;
            }
            case OP_MEMORY_OP: {
                // [PK_MEM_OP, oi, cell_base, block_len, readable_len, has_pred, mode, target_w, target_flag,
                //  E(operation), E(index), E(value), E(pred)?]; mode / target are patched in the second pass
                Block &b = st[o.block_id];
                s.insert(s.end(), {PK_MEM_OP, oi, b.base, b.len, b.readable, o.has_predicate ? 1u : 0u, 2u, 0xFFFFFFFFu, 0u});
                emit_expr(s, pool, o.mem_operation);
                emit_expr(s, pool, o.mem_index);
                emit_expr(s, pool, o.mem_value);
                if (o.has_predicate) emit_expr(s, pool, o.predicate);
                break;
            }
            case OP_BRILLIG: {
                // [PK_BRILLIG, oi, has_pred, n_inputs, n_outputs, bc_offset, n_bytecode, n_regs, mem_cap, fc_desc_off, fc_vals_off,
                //  fc_slot, E(pred)?, inputs: (is_array, n, E x n)..., outputs: (is_array, n, (w, flag) x n)...]
                const BrilligCall &b = *o.brillig;
                p.prog_class[oi] = CLS_BRILLIG;
                // A caller-supplied BlackBoxFunctionSolver is the VM's solver too (brillig_vm/src/lib.rs:61,81,298; black_box.rs:139-163): the three
                // trait functions inside a Brillig program become INTERNAL foreign calls -- the VM stops there like at an oracle call, hands
                // the operands to the host (same layout: registers as one value, heap vectors as their slices), the batch driver answers every
                // waiting instance through the solver in one pass (*_batch members when the vtable has them) and the opcode re-runs its VM with
                // the answers in the result store (batch_exact.cpp resolve_internal_calls). The caller never sees these calls.
                std::vector<BrilligOp> rewritten;
                if (host_blackbox) {
                    rewritten = b.bytecode;
                    auto R = [](uint64_t r) { return RegOrMem{0u, r, 0}; };
                    auto A = [](uint64_t ptr, uint64_t n) { return RegOrMem{1u, ptr, n}; };
                    auto V = [](uint64_t ptr, uint64_t size_reg) { return RegOrMem{2u, ptr, size_reg}; };
                    for (BrilligOp &op : rewritten) {
                        if (op.op != BR_BLACK_BOX || op.bbop < 6) continue;
                        BrilligOp f;
                        f.op = BR_FOREIGN_CALL;
                        if (op.bbop == 6) {  // SchnorrVerify { public_key_x, public_key_y, message, signature, result }
                            f.function = PLAN_FC_INTERNAL_SCHNORR;
                            f.inputs = {R(op.bb[0]), R(op.bb[1]), V(op.bb[2], op.bb[3]), V(op.bb[4], op.bb[5])};
                            f.dests = {R(op.bb[6])};
                        } else if (op.bbop == 7) {  // Pedersen { inputs, domain_separator, output }: two values are written whatever size the array declares
                            f.function = PLAN_FC_INTERNAL_PEDERSEN;
                            f.inputs = {V(op.bb[0], op.bb[1]), R(op.bb[2])};
                            f.dests = {A(op.bb[3], 2)};
                        } else {  // FixedBaseScalarMul { low, high, result }
                            f.function = PLAN_FC_INTERNAL_FIXED_BASE;
                            f.inputs = {R(op.bb[0]), R(op.bb[1])};
                            f.dests = {A(op.bb[2], 2)};
                        }
                        op = f;
                    }
                }
                const std::vector<BrilligOp> &code = host_blackbox ? rewritten : b.bytecode;
                uint32_t bc_off = (uint32_t)p.bytecode.size();
                uint64_t max_reg = std::max(b.inputs.size(), b.outputs.size());
                uint64_t mem_hint = 0, arr_cells = 0;
                auto reg = [&](uint64_t r) { if (r < 65536) max_reg = std::max<uint64_t>(max_reg, r + 1); return clamp_reg(r); };
                for (auto &in : b.inputs) if (in.is_array) arr_cells += in.arr.size();
                bool has_foreign = false, has_grumpkin = false;
                uint32_t max_hash_hint = 0;
                for (auto &op : code) {
                    // 8 words per instruction: [op, a, b, c, sub_op | bit_size << 8, location, const index, extra offset]
                    uint32_t w[8] = {op.op, 0, 0, 0, 0, 0, 0, 0};
                    switch (op.op) {
                    case BR_BINARY_FIELD_OP: case BR_BINARY_INT_OP:
                        w[1] = reg(op.a); w[2] = reg(op.b); w[3] = reg(op.c);
                        // (bit sizes of 512 and more all behave alike -- no operation can reach the modulus 2^bits -- but for the constant below)
                        w[4] = op.sub_op | (op.op == BR_BINARY_INT_OP ? std::min<uint32_t>(op.bit_size, 0xFFFFFFu) << 8 : 0u);
                        if (op.op == BR_BINARY_INT_OP && op.sub_op == 1 && op.bit_size > 256) w[6] = pool.intern(pow2_mod_p(op.bit_size));  // Sub: 2^bits mod p
                        break;
                    case BR_JUMP_IF_NOT: case BR_JUMP_IF:
                        w[1] = reg(op.a); w[5] = (uint32_t)std::min<uint64_t>(op.location, 0xFFFFFFFFull);
                        break;
                    case BR_JUMP: case BR_CALL: w[5] = (uint32_t)std::min<uint64_t>(op.location, 0xFFFFFFFFull); break;
                    case BR_CONST: {
                        w[1] = reg(op.a);
                        w[6] = pool.intern(op.value);
                        uint64_t cv[4];
                        frh::to_canonical(op.value, cv);
                        if (!(cv[1] | cv[2] | cv[3]) && cv[0] < 65536) mem_hint = std::max(mem_hint, cv[0]);
                        break;
                    }
                    case BR_MOV: case BR_LOAD: case BR_STORE: w[1] = reg(op.a); w[2] = reg(op.b); break;
                    case BR_BLACK_BOX: {
                        // operands go to an extra block behind the instruction array (patched below)
                        w[4] = op.bbop;
                        if (op.bbop >= 6) has_grumpkin = true;
                        if (op.bbop == 4 || op.bbop == 5) p.needs_ecdsa = true;
                        max_hash_hint = 1;
                        break;
                    }
                    case BR_FOREIGN_CALL: has_foreign = true; break;  // operands: extra block, patched below
                    default: break;
                    }
                    p.bytecode.insert(p.bytecode.end(), w, w + 8);
                }
                // extra blocks (black box operands) after the fixed-size instruction array
                for (size_t k = 0; k < code.size(); k++) {
                    const BrilligOp &op = code[k];
                    if (op.op != BR_BLACK_BOX) continue;
                    // operand layout mirrors brillig/src/black_box.rs:7-53: HeapVector = (pointer reg, size reg),
                    // HeapArray = (pointer reg, literal size), RegisterIndex = reg. n_reg_words = leading register words.
                    static const int nwords[9] = {4, 4, 4, 3, 9, 9, 7, 5, 4};
                    // bit i set: operand word i is a register (else a literal array size)
                    static const uint32_t reg_mask[9] = {0x7, 0x7, 0x7, 0x7, 0x157, 0x157, 0x7f, 0xf, 0x7};
                    p.bytecode[bc_off + 8 * k + 7] = (uint32_t)p.bytecode.size();
                    for (int i = 0; i < nwords[op.bbop]; i++)
                        p.bytecode.push_back((reg_mask[op.bbop] >> i) & 1 ? reg(op.bb[i]) : (uint32_t)std::min<uint64_t>(op.bb[i], 0xFFFFFFFFull));
                }
                // ForeignCall operands: [n_dests, n_inputs, (kind, reg, size) x n_dests, (kind, reg, size) x n_inputs];
                // kind 0 register, 1 HeapArray (size literal), 2 HeapVector (size register)
                uint64_t fc_pending_vals = 0;
                for (size_t k = 0; k < code.size(); k++) {
                    const BrilligOp &op = code[k];
                    if (op.op != BR_FOREIGN_CALL) continue;
                    p.bytecode[bc_off + 8 * k + 7] = (uint32_t)p.bytecode.size();
                    p.bytecode.push_back((uint32_t)op.dests.size());
                    p.bytecode.push_back((uint32_t)op.inputs.size());
                    uint64_t vals = 0;
                    for (const std::vector<RegOrMem> *v : {&op.dests, &op.inputs})
                        for (const RegOrMem &m : *v) {
                            p.bytecode.push_back(m.kind);
                            p.bytecode.push_back(reg(m.reg));
                            p.bytecode.push_back(m.kind == 2 ? reg(m.size) : (uint32_t)std::min<uint64_t>(m.size, 0xFFFFFFFFull));
                            if (v == &op.inputs) vals += m.kind == 0 ? 1 : (m.kind == 1 ? std::min<uint64_t>(m.size, 1u << 20) : 0);
                            if (m.kind == 1) mem_hint = std::max<uint64_t>(mem_hint, std::min<uint64_t>(m.size, 1u << 16));
                        }
                    fc_pending_vals = std::max(fc_pending_vals, vals);
                    p.fc_function[((uint64_t)oi << 32) | (uint32_t)k] = op.function;
                    p.fc_max_inputs = std::max<uint32_t>(p.fc_max_inputs, (uint32_t)op.inputs.size());
                }
                // results the circuit already carries (Brillig::foreign_call_results): descriptor
                // [n_results, (n_values, (is_array, n) x n_values) x n_results] + one constant index per value
                uint32_t fc_desc_off = 0xFFFFFFFFu, fc_vals_off = 0xFFFFFFFFu;
                if (has_foreign) {
                    std::vector<uint32_t> vals;
                    fc_desc_off = (uint32_t)p.bytecode.size();
                    p.bytecode.push_back((uint32_t)b.fc_results.size());
                    for (auto &res : b.fc_results) {
                        p.bytecode.push_back((uint32_t)res.values.size());
                        for (auto &v : res.values) {
                            p.bytecode.push_back(v.is_array ? 1u : 0u);
                            p.bytecode.push_back(v.is_array ? (uint32_t)v.arr.size() : 1u);
                            if (v.is_array) for (auto &x : v.arr) vals.push_back(pool.intern(x));
                            else vals.push_back(pool.intern(v.single));
                        }
                    }
                    fc_vals_off = (uint32_t)p.bytecode.size();
                    p.bytecode.insert(p.bytecode.end(), vals.begin(), vals.end());
                    p.has_foreign_calls = true;
                }
                if (has_grumpkin) p.needs_grumpkin = true;
                uint64_t mem_cap = arr_cells + mem_hint + 64 + (max_hash_hint ? 64 : 0) + (has_foreign ? 64 : 0);
                mem_cap = std::max<uint64_t>(mem_cap, (uint64_t)std::max<int64_t>(tune.brillig_mem_cells, 0));
                mem_cap = std::min<uint64_t>(mem_cap, 1u << 20);
                uint32_t n_regs = (uint32_t)std::max<uint64_t>(max_reg, 1);
                // scratch: registers + memory cells (8 words each) + call stack + byte staging for hashes (batch.cpp brillig_scratch_words
                // sizes the retry passes of the exact path by the same formula)
                p.prog_scratch[oi] = (uint32_t)((n_regs + mem_cap) * 8 + (uint64_t)std::min<int64_t>(std::max<int64_t>(tune.brillig_call_depth, 1), 1 << 20) +
                                                (max_hash_hint ? mem_cap / 4 + 16 : 0));
                p.fc_pending_vals = std::max<uint64_t>(p.fc_pending_vals, has_foreign ? fc_pending_vals + mem_cap : 0);
                uint32_t fc_slot = 0xFFFFFFFFu;
                if (has_foreign) { fc_slot = (uint32_t)p.fc_slot_opcode.size(); p.fc_slot_opcode.push_back(oi); }
                s.insert(s.end(), {PK_BRILLIG, oi, b.has_predicate ? 1u : 0u, (uint32_t)b.inputs.size(), (uint32_t)b.outputs.size(), bc_off,
                                   (uint32_t)b.bytecode.size(), n_regs, (uint32_t)mem_cap, fc_desc_off, fc_vals_off, fc_slot});
                if (b.has_predicate) emit_expr(s, pool, b.predicate);
                for (auto &in : b.inputs) {
                    s.push_back(in.is_array ? 1u : 0u);
                    s.push_back(in.is_array ? (uint32_t)in.arr.size() : 1u);
                    if (in.is_array) for (auto &e : in.arr) emit_expr(s, pool, e);
                    else emit_expr(s, pool, in.single);
                }
                for (auto &ot : b.outputs) {
                    s.push_back(ot.is_array ? 1u : 0u);
                    s.push_back(ot.is_array ? (uint32_t)ot.arr.size() : 1u);
                    if (ot.is_array) for (uint32_t w : ot.arr) out(w);
                    else out(ot.w);
                }
                if (tune.brillig_inline) emit_straight_line(b, oi, pool, sl_prog, sl_of);
                break;
            }
            default:
                unsupported(oi, "opcode kind " + std::to_string(o.kind));
                s.insert(s.end(), {0xFFFFFFFFu, oi});
            }
        }
    }
    if (!p.unsupported.empty()) return false;
    sl_base = (uint32_t)p.prog.size();  // nothing in `prog` moves from here on: later passes only append
    p.prog.insert(p.prog.end(), sl_prog.begin(), sl_prog.end());
        return true;
    }

    void assign_out(uint32_t oi, uint32_t lvl, bool heavy) {
        // insert_value (pwg/mod.rs:338-357) for the outputs of opcode oi, in record order
        for (auto &slot : out_slots[oi]) {
            uint32_t w = slot.second;
            if (known[w]) p.prog[slot.first] = 1;  // compare, never overwrite
            else {
                known[w] = 1;
                level[w] = hlevel[w] = lvl;
                p.producer[w] = oi;
                if (heavy) { heavy_level[w] = lvl; level[w] = lvl + out_latency; wlane[w] = out_lane; }
            }
        }
    }
    uint32_t out_levels(uint32_t oi, uint32_t lvl, bool heavy) {  // an already-assigned output is read (compared)
        for (auto &slot : out_slots[oi])
            if (known[slot.second]) lvl = std::max(lvl, heavy ? hlevel[slot.second] : level[slot.second]);
        return lvl;
    }

    void pin_witnesses() {
    // =========================================================================== projective witnesses
    // A witness that only Arithmetic opcodes touch may be kept as scale_w * value: the gate that solves it picks scale_w so that
    // its most expensive coefficient becomes 1 (q_M a b + q_1 c + q_c over q_o: the product needs no coefficient multiplication
    // and shares one Montgomery reduction with q_1' c), and every later gate divides its coefficients by the scales of its
    // operands -- all on the host, field arithmetic is exact, so the canonical value scale_w^-1 * stored is bit-identical.
    // Pinned (scale 1): initial witnesses and everything a non-Arithmetic opcode mentions. Export and the exact path unscale.
    pinned.assign(nw, 0);
    is_scaled.assign(nw, 0);
    scaling_on = tune.scale != 0;
    {
        auto pin = [&](uint32_t w) { if (w < nw) pinned[w] = 1; };
        auto pin_expr = [&](const Expr &e) {
            for (auto &t : e.mul) { pin(t.l); pin(t.r); }
            for (auto &t : e.lin) pin(t.w);
        };
        for (uint32_t i = 0; i < n_initial; i++) pin(initial_ids[i]);
        for (uint32_t oi = 0; oi < c.opcodes.size(); oi++) {
            const Opcode &o = c.opcodes[oi];
            if (o.kind == OP_ARITHMETIC) continue;
            for (auto &slot : out_slots[oi]) pin(slot.second);
            switch (o.kind) {
            case OP_BLACKBOX:
                for (int g = 0; g < 4; g++)
                    for (auto &in : o.bb->in[g]) pin(in.witness);
                for (auto &in : o.bb->in_agg) pin(in.witness);
                for (uint32_t w : o.bb->out) pin(w);
                break;
            case OP_DIRECTIVE:
                pin_expr(o.dir->a); pin_expr(o.dir->b); pin_expr(o.dir->predicate);
                pin(o.dir->q); pin(o.dir->r);
                for (uint32_t w : o.dir->bw) pin(w);
                for (auto &el : o.dir->sort_inputs)
                    for (auto &ex : el) pin_expr(ex);
                break;
            case OP_MEMORY_INIT:
                for (uint32_t w : o.init) pin(w);
                break;
            case OP_MEMORY_OP:
                pin_expr(o.mem_operation); pin_expr(o.mem_index); pin_expr(o.mem_value); pin_expr(o.predicate);
                break;
            case OP_BRILLIG:
                pin_expr(o.brillig->predicate);
                for (auto &in : o.brillig->inputs) {
                    pin_expr(in.single);
                    for (auto &e : in.arr) pin_expr(e);
                }
                for (auto &ot : o.brillig->outputs) {
                    pin(ot.w);
                    for (uint32_t w : ot.arr) pin(w);
                }
                break;
            default: break;
            }
        }
    }
    }

    // =========================================================================== pass 4: generic-instance replay + levels
    // every opcode in program order: a non-Arithmetic opcode becomes a record of its class at its level (level_record), an Arithmetic opcode a
    // folded gate (level_gate); stops at the first opcode the generic instance cannot execute (p.truncated_at)
    void replay() {
    // =========================================================================== generic-instance replay + levels
    heavy_level.assign(nw, 0);
    wlane.assign(nw, 0);
    hlevel.assign(nw, 0);
    K_heavy = (uint32_t)std::max<int64_t>(tune.heavy_epoch, 1), D_heavy = (uint32_t)std::max<int64_t>(tune.heavy_latency, 0);
    D_pedersen = (uint32_t)std::max<int64_t>(tune.pedersen_latency, 0), K_pedersen = (uint32_t)std::max<int64_t>(tune.pedersen_epoch, 1);
    scale_slot.assign(nw, 0xFFFFFFFFu);
    // Relaxed rows (gate_eval.hpp): a witness that may carry a scale is also stored as ANY representative below 2^256; kbound[w] = its
    // bound in units of p / 256 (everything else is canonical: GATE_K_CANON). gate_of[w]: the gate that writes w.
    relax_on = scaling_on && tune.relax != 0;
    p.kbound.assign(nw, GATE_K_CANON);
    gate_of.assign(nw, 0xFFFFFFFFu);
    for (uint32_t oi = 0; oi < c.opcodes.size() && p.truncated_at == 0xFFFFFFFFu; oi++) {
        const Opcode &o = c.opcodes[oi];
        if (!(o.kind != OP_ARITHMETIC ? level_record(oi, o) : level_gate(oi, o))) break;
    }

    p.unscale_index.assign(nw, 0xFFFFFFFFu);
    for (uint32_t w = 0; w < nw; w++)
        if (is_scaled[w]) {
            p.unscale_index[w] = (uint32_t)p.scaled_ids.size();
            p.scaled_ids.push_back(w);
            p.unscale.push_back(wsi[scale_slot[w]]);
        }
    }

    // a non-Arithmetic opcode: its reads for the generic instance, its level, its outputs; false: the generic instance cannot execute it
    bool level_record(uint32_t oi, const Opcode &o) {
        // a Brillig opcode with a straight-line record runs in the light class of the level schedule (the exact path keeps its VM record)
        const auto sl_it = o.kind == OP_BRILLIG ? sl_of.find(oi) : sl_of.end();
        const uint32_t rec_cls = o.kind == OP_BLACKBOX && o.bb->func == BB_PEDERSEN && !host_blackbox ? (uint32_t)CLS_PEDERSEN
                                 : sl_it != sl_of.end() ? (tune.sl_lane ? (uint32_t)CLS_BRILLIG : (uint32_t)CLS_LIGHT) : (uint32_t)p.prog_class[oi];
        const bool rec_heavy = is_heavy(rec_cls);
        Reads rd(known, rec_heavy ? hlevel : level);
        uint32_t extra_level = 0;
        int mem_access = 0;  // 1 read, 2 write
        uint64_t bytes_written = out_slots[oi].size();
        switch (o.kind) {
        case OP_BLACKBOX: {
            const BlackBoxCall &b = *o.bb;
            for (int g = 0; g < 4; g++)
                for (auto &in : b.in[g]) rd.witness(in.witness);
            if (b.has_in_agg)
                for (auto &in : b.in_agg) rd.witness(in.witness);
            break;
        }
        case OP_DIRECTIVE: {
            const Directive &d = *o.dir;
            if (d.kind == DIR_PERMUTATION_SORT) {
                for (auto &el : d.sort_inputs)
                    for (auto &ex : el) rd.expr(ex);
                break;
            }
            rd.expr(d.a);
            if (d.kind == DIR_QUOTIENT) { rd.expr(d.b); if (d.has_predicate) rd.expr(d.predicate); }
            break;
        }
        case OP_MEMORY_INIT: {
            Block &b = blocks[o.block_id];
            for (uint32_t w : o.init) rd.witness(w);
            extra_level = std::max(b.level, b.rlevel);
            mem_access = 2;
            bytes_written = o.init.size();
            break;
        }
        case OP_MEMORY_OP: {
            Block &b = blocks[o.block_id];
            rd.expr(o.mem_operation);
            rd.expr(o.mem_index);
            if (o.has_predicate) rd.expr(o.predicate);
            // read or write is decided by the VALUE of `operation` (memory_op.rs:91); static only if it is a constant
            uint32_t off = p.prog_offset[oi];
            const Expr &op = o.mem_operation;
            bool is_const = op.mul.empty() && op.lin.empty();
            if (!is_const) { rd.ok = false; break; }
            if (op.qc.is_zero()) {
                // read: Expression::to_witness (expression/mod.rs:158-172) on the evaluated value expression
                const Expr &v = o.mem_value;
                bool shape = v.mul.empty() && v.lin.size() == 1 && v.lin[0].c == frh::one() && v.qc.is_zero() &&
                             v.lin[0].w < known.size() && !known[v.lin[0].w];
                if (!shape) { rd.ok = false; break; }
                p.prog[off + 6] = 1;
                p.prog[off + 7] = v.lin[0].w;
                out_slots[oi].push_back({off + 8, v.lin[0].w});
                bytes_written = 1;
                extra_level = b.level;
                mem_access = 1;
            } else {
                rd.expr(o.mem_value);
                p.prog[off + 6] = 0;
                bytes_written = 1;
                extra_level = std::max(b.level, b.rlevel);
                mem_access = 2;
            }
            break;
        }
        case OP_BRILLIG: {
            const BrilligCall &b = *o.brillig;
            if (b.has_predicate) rd.expr(b.predicate);
            for (auto &in : b.inputs) {
                if (in.is_array) for (auto &e : in.arr) rd.expr(e);
                else rd.expr(in.single);
            }
            break;
        }
        default: rd.ok = false;
        }
        if (!rd.ok) { p.truncated_at = oi; return false; }
        uint32_t lvl = std::max(rd.lvl, extra_level);
        lvl = out_levels(oi, lvl, rec_heavy) + 1;
        if (rec_heavy) lvl = (lvl + K_heavy - 1) / K_heavy * K_heavy;  // the next heavy batch
        if (rec_cls == CLS_PEDERSEN && K_pedersen > 1) lvl = (lvl + K_pedersen - 1) / K_pedersen * K_pedersen;
        std::vector<uint32_t> rec_reads = rd.ws;  // compared outputs are reads too
        for (auto &slot : out_slots[oi])
            if (known[slot.second]) rec_reads.push_back(slot.second);
        if (!rec_heavy)  // a main-stream record: which heavy records does it wait for
            for (uint32_t w : rec_reads)
                if (heavy_level[w]) heavy_reads.push_back({lvl, w});
        out_latency = rec_cls == CLS_PEDERSEN ? std::max(D_heavy, D_pedersen) : D_heavy;
        out_lane = (uint8_t)(1 + heavy_lane(rec_cls));
        assign_out(oi, lvl, rec_heavy);
        if (mem_access == 1) blocks[o.block_id].rlevel = std::max(blocks[o.block_id].rlevel, lvl);
        else if (mem_access == 2) { blocks[o.block_id].level = lvl; blocks[o.block_id].rlevel = 0; }
        uint64_t bytes = 32ull * (rd.distinct() + bytes_written);
        p.algorithmic_bytes += bytes;
        p.cls_algorithmic_bytes[p.prog_class[oi]] += bytes;
        p.n_other_records++;
        records.push_back({lvl, rec_cls, oi, std::move(rec_reads)});
        if (sl_it != sl_of.end()) {  // same "already assigned" flags as the VM record's outputs, in order
            records.back().prog_at = sl_base + sl_it->second.first;
            for (size_t k = 0; k < out_slots[oi].size(); k++) p.prog[sl_base + sl_it->second.first + sl_it->second.second + 2 * k] = p.prog[out_slots[oi][k].first];
            p.n_brillig_inlined++;
        }
        return true;
    }

    // an Arithmetic opcode: classify its terms like ArithmeticSolver::evaluate, fold -1/coeff (and the scales) into the coefficients, choose the
    // output's scale, emit the gate record and bound its result; false: the generic instance cannot execute it (panic / TooManyUnknowns)
    bool level_gate(uint32_t oi, const Opcode &o) {
        std::vector<uint32_t> &kbound = p.kbound;
        const Expr &e = o.expr;
        // classify terms like ArithmeticSolver::evaluate (arithmetic.rs:212-239) for the generic instance
        struct Prod { FrH c; uint32_t a, b; };
        struct Lin { FrH c; uint32_t a; };
        std::vector<Prod> prods;
        std::vector<Lin> lins;
        uint32_t residual_mul = 0, n_unknown = 0;
        bool unk_is_folded = false;
        FrH unk_coef = frh::zero();
        uint32_t unk_w = 0, unk_partner = 0;
        for (auto &t : e.mul) {
            bool kl = known[t.l], kr = known[t.r];
            if (kl && kr) prods.push_back({t.c, t.l, t.r});
            else if (!kl && !kr) { if (!t.c.is_zero()) residual_mul++; }
            else if (!t.c.is_zero()) {  // OneUnknown(c * known, unknown); generic instance: known != 0
                n_unknown++;
                unk_is_folded = true;
                unk_coef = t.c;
                unk_w = kl ? t.r : t.l;
                unk_partner = kl ? t.l : t.r;
            }
        }
        for (auto &t : e.lin) {
            if (known[t.w]) lins.push_back({t.c, t.w});
            else if (!t.c.is_zero()) {
                n_unknown++;
                unk_is_folded = false;
                unk_coef = t.c;
                unk_w = t.w;
            }
        }
        if (residual_mul > 0 || n_unknown > 1) {  // panic / TooManyUnknowns for the generic instance
            p.truncated_at = oi;
            return false;
        }
        PendingGate g;
        uint32_t lvl = 0;
        std::vector<uint32_t> reads;
        auto rd = [&](uint32_t w) { lvl = std::max(lvl, level[w]); reads.push_back(w); };
        for (auto &t : prods) { rd(t.a); rd(t.b); }
        for (auto &t : lins) rd(t.a);
        uint32_t kind = GATE_ASSERT, inv_level = 0;
        FrH scale = frh::one();
        bool scaled = false;
        if (n_unknown == 1) {
            // out = -(sum)/coeff (arithmetic.rs:120) or -(sum)/(c*partner) (:86): fold -1/coeff into every coefficient
            scale = pool.neg_inv(unk_coef);
            scaled = true;
            if (unk_is_folded) {
                kind = GATE_SOLVE_DYN;
                reads.push_back(unk_partner);
                const uint32_t K = (uint32_t)std::max<int64_t>(tune.inv_epoch, 1);
                inv_level = 1 + ((level[unk_partner] + K - 1) / K) * K;  // first batch level after the denominator is known
                lvl = std::max(lvl, inv_level + (uint32_t)std::max<int64_t>(tune.inv_latency, 0));  // (levels of slack before the gate reads the inverse)
            } else kind = GATE_SOLVE;
        }
        // zero-coefficient products / linear terms contribute exactly 0: drop them from the device program.
        // Effective coefficient of a term on the STORED operands: base * c / (scale_a scale_b), base = -1/coeff of the unknown
        // (times the denominator's scale for SOLVE_DYN: the inverse table holds 1 / stored denominator), then times the
        // multiplier m this gate chooses for its own output (scale_out = m) -- or for the whole sum of an ASSERT gate.
        struct Term { bool prod; uint32_t a, b; FrH c, pe; };
        std::vector<Term> terms;
        FrH base = scaled ? scale : f_one, inv_base = scaled ? frh::neg(unk_coef) : f_one;
        if (kind == GATE_SOLVE_DYN && is_scaled[unk_partner]) {
            base = frh::mul(base, ws[scale_slot[unk_partner]]);
            inv_base = frh::mul(inv_base, wsi[scale_slot[unk_partner]]);
        }
        auto over_scale = [&](FrH x, uint32_t w) { return is_scaled[w] ? frh::mul(x, wsi[scale_slot[w]]) : x; };
        auto times_scale = [&](FrH x, uint32_t w) { return is_scaled[w] ? frh::mul(x, ws[scale_slot[w]]) : x; };
        for (auto &t : prods)
            if (!t.c.is_zero()) terms.push_back({true, t.a, t.b, t.c, over_scale(over_scale(frh::mul(base, t.c), t.a), t.b)});
        for (auto &t : lins)
            if (!t.c.is_zero()) terms.push_back({false, t.a, 0, t.c, over_scale(frh::mul(base, t.c), t.a)});
        // multiply-adds of the device's gate sum for a multiplier m (gate_eval.hpp gate_eval): a product with a general
        // coefficient is a product (153) and then a dot participant, a +1 product and a general linear term are dot
        // participants (two share one reduction: 234, a single one 153), a -1 product is a product
        auto gate_cost = [&](const FrH *m) {
            uint32_t n_dot = 0, cost = 0;
            for (auto &t : terms) {
                const FrH e = m ? frh::mul(t.pe, *m) : t.pe;
                const bool is_one = e == f_one, is_m1 = e == f_minus_one;
                if (t.prod) {
                    if (is_m1) cost += 153;
                    else { n_dot++; if (!is_one) cost += 153; }
                } else if (!is_one && !is_m1) n_dot++;
            }
            return cost + 234 * (n_dot / 2) + 153 * (n_dot & 1);
        };
        FrH m = f_one, m_inv = f_one;
        bool have_m = false;
        if (scaling_on && !terms.empty() && (kind == GATE_ASSERT || !pinned[unk_w])) {
            uint32_t best = gate_cost(nullptr);
            for (size_t t = 0; t < terms.size() && t < 4 && best > 0; t++) {
                if (terms[t].pe == f_one) continue;
                // 1 / pe_t without an inversion: every factor's inverse is at hand
                FrH cand = frh::mul(inv_base, frh::neg(pool.neg_inv(terms[t].c)));
                cand = times_scale(cand, terms[t].a);
                if (terms[t].prod) cand = times_scale(cand, terms[t].b);
                const uint32_t cst = gate_cost(&cand);
                if (cst < best) { best = cst; m = cand; m_inv = terms[t].pe; have_m = true; }
            }
        }
        auto sc = [&](const FrH &pe) { return have_m ? frh::mul(pe, m) : pe; };
        // Record layout (gate_record.hpp; consumed by gate_eval, gate_eval.hpp): terms with a general coefficient go through the
        // multiplied lists; terms with coefficient +1 / -1 are listed without a coefficient and are only added / subtracted
        // on the device (up to 255 of each kind, the rest keep an explicit constant).
        //   [kind | np_mac << 8 | nl_mac << 16, opcode, out, q_c, partner, np_pos | np_neg << 8 | nl_pos << 16 | nl_neg << 24,
        //    np_mac x (coef[8], a, b), nl_mac x (coef[8], w), np_pos x (a, b), np_neg x (a, b), nl_pos x (w), nl_neg x (w)]
        std::vector<uint32_t> pm, lm, pp, pn, lp, ln;
        // general coefficients travel inline (8 words, the device's Montgomery form): the wave reaches them with the record
        // itself instead of one more dependent scalar load through the constant pool
        auto push_coef = [](std::vector<uint32_t> &v, const FrH &c) {
            const FrH d = frh::to_device_form(c);
            for (int i = 0; i < 4; i++) { v.push_back((uint32_t)d.l[i]); v.push_back((uint32_t)(d.l[i] >> 32)); }
        };
        const FrH qc_scaled = sc(frh::mul(base, e.qc));
        for (auto &t : terms) {
            const FrH e_t = sc(t.pe);
            const uint32_t cc = e_t == f_one ? COEF_ONE : e_t == f_minus_one ? COEF_MINUS_ONE : 0u;  // (a general coefficient travels inline: nothing to intern)
            if (t.prod) {
                if (cc == COEF_ONE && pp.size() < 2 * 255) { pp.push_back(t.a); pp.push_back(t.b); }
                else if (cc == COEF_MINUS_ONE && pn.size() < 2 * 255) { pn.push_back(t.a); pn.push_back(t.b); }
                else { push_coef(pm, e_t); pm.push_back(t.a); pm.push_back(t.b); }
            } else {
                if (cc == COEF_ONE && lp.size() < 255) lp.push_back(t.a);
                else if (cc == COEF_MINUS_ONE && ln.size() < 255) ln.push_back(t.a);
                else { push_coef(lm, e_t); lm.push_back(t.a); }
            }
        }
        if (pm.size() / 10 > 255 || lm.size() / 9 > 255) { p.truncated_at = oi; return false; }
        g.level = lvl + 1;
        g.words = {kind | (uint32_t)(pm.size() / 10) << 8 | (uint32_t)(lm.size() / 9) << 16, oi, kind == GATE_ASSERT ? 0u : unk_w,
                   pool.constant(qc_scaled), kind == GATE_SOLVE_DYN ? unk_partner : 0u,
                   (uint32_t)(pp.size() / 2) | (uint32_t)(pn.size() / 2) << 8 | (uint32_t)lp.size() << 16 | (uint32_t)ln.size() << 24};
        for (auto *v : {&pm, &lm, &pp, &pn, &lp, &ln}) g.words.insert(g.words.end(), v->begin(), v->end());
        {
            // The bound of the record's result, in the order gate_eval (gate_eval.hpp) sums it: hk = bound of the lazy side sum in units of
            // p / 256, hw = its limb weight (the reduction inside gate_h_room resets both).
            uint32_t hk = qc_scaled.is_zero() ? 0u : GATE_K_CANON, hw = qc_scaled.is_zero() ? 0u : 16u;
            auto room = [&](uint32_t weight) {
                if (hw + weight > GATE_H_MAX) { hk = GATE_K_WEAK; hw = GATE_H_AFTER_WEAK; }
                hw += weight;
            };
            uint32_t sub_k = 1;
            for (uint32_t w : ln)
                while (sub_k < 3 && (GATE_K_CANON << sub_k) < kbound[w]) sub_k++;
            for (size_t i = 0; i < pn.size(); i += 2) { room(33); hk += 2 * GATE_K_CANON; }
            for (uint32_t w : lp) { room(16); hk += kbound[w]; }
            for (size_t i = 0; i < ln.size(); i++) { room(33); hk += GATE_K_CANON << sub_k; }
            // the Montgomery reductions, each taking the running sum along (fr29_dot_add): result < p + what its products add + the sum so far
            auto mac_k = [&](size_t im) {  // the im-th multiplied term: coefficient (canonical) x (product | witness)
                const size_t n_pm = pm.size() / 10;
                if (im < n_pm) return gate_k_product(GATE_K_CANON + gate_k_product(kbound[pm[10 * im + 8]], kbound[pm[10 * im + 9]]), GATE_K_CANON);
                return gate_k_product(kbound[lm[9 * (im - n_pm) + 8]], GATE_K_CANON);
            };
            auto pp_k = [&](size_t ip) { return gate_k_product(kbound[pp[2 * ip]], kbound[pp[2 * ip + 1]]); };
            const size_t n_pp = pp.size() / 2, n_mac = pm.size() / 10 + lm.size() / 9;
            size_t im = 0, ip = 0;
            uint32_t n_red = 0;
            auto reduction = [&](uint32_t adds) {
                hk += GATE_K_CANON + adds;
                if (++n_red == GATE_REDUCTIONS_PER_WEAK) { hk = GATE_K_WEAK; n_red = 0; }
            };
            for (; ip < n_pp && im < n_mac; ip++, im++) reduction(pp_k(ip) + mac_k(im));
            for (; im < n_mac; im += 2) reduction(mac_k(im) + (n_mac - im == 1 ? 0u : mac_k(im + 1)));
            for (; ip < n_pp; ip += 2) reduction(pp_k(ip) + (n_pp - ip == 1 ? 0u : pp_k(ip + 1)));
            uint32_t acc_k = hk;
            uint32_t flags = sub_k << GATE_SUBK_SHIFT;
            if (kind == GATE_SOLVE_DYN) {
                if (acc_k > 7 * GATE_K_CANON) { flags |= GATE_PRESUM_WEAK; acc_k = GATE_K_WEAK; }
                acc_k = GATE_K_CANON + gate_k_product(acc_k, GATE_K_INVERSE);
            }
            if (kind != GATE_ASSERT) {
                uint32_t mode = GATE_OUT_CANON, kb = GATE_K_CANON;
                if (relax_on && !pinned[unk_w]) {
                    if (acc_k <= GATE_K_ROW_MAX) { mode = GATE_OUT_ASIS; kb = acc_k; }
                    else { mode = GATE_OUT_WEAK; kb = GATE_K_WEAK; }
                }
                flags |= mode << GATE_OUT_SHIFT;
                kbound[unk_w] = kb;
                p.n_gate_out_mode[mode]++;
            }
            g.words[0] |= flags;
            p.max_gate_bound = std::max(p.max_gate_bound, acc_k);
        }
        std::sort(reads.begin(), reads.end());
        reads.erase(std::unique(reads.begin(), reads.end()), reads.end());
        g.reads = reads;
        uint64_t bytes = 32ull * (reads.size() + (kind == GATE_ASSERT ? 0 : 1));
        p.algorithmic_bytes += bytes;
        // the denominator of a SOLVE_DYN gate is read by the inversion kernel, everything else by the gate kernel
        p.arith_algorithmic_bytes += bytes - (kind == GATE_SOLVE_DYN ? 32 : 0);
        if (kind == GATE_SOLVE_DYN) p.dyn_algorithmic_bytes += 32;
        if (kind != GATE_ASSERT) {
            known[unk_w] = 1;
            level[unk_w] = hlevel[unk_w] = g.level;
            p.producer[unk_w] = oi;
            if (have_m || (relax_on && !pinned[unk_w])) {  // stored = m * value (m = 1: a relaxed row, whose readers outside the gate kernels take the same road)
                is_scaled[unk_w] = 1;
                scale_slot[unk_w] = (uint32_t)ws.size();
                ws.push_back(m);
                wsi.push_back(m_inv);
            }
            gate_of[unk_w] = (uint32_t)gates.size();
        }
        if (kind == GATE_SOLVE_DYN && gate_of[unk_partner] != 0xFFFFFFFFu) {
            // the inversion kernel tests the denominator for zero on the stored row (arithmetic.rs:217-221): its gate stores the canonical value
            uint32_t &pw0 = gates[gate_of[unk_partner]].words[0];
            const uint32_t was = (pw0 >> GATE_OUT_SHIFT) & 3u;
            if (was != GATE_OUT_CANON) {
                p.n_gate_out_mode[was]--;
                p.n_gate_out_mode[GATE_OUT_CANON]++;
                pw0 = (pw0 & ~(3u << GATE_OUT_SHIFT)) | GATE_OUT_CANON << GATE_OUT_SHIFT;
                kbound[unk_partner] = GATE_K_CANON;  // (the gates in between were sized for the looser bound: still an upper bound)
            }
        }
        if (kind == GATE_SOLVE_DYN) {
            p.n_dyn_gates++;
            inverses.push_back({inv_level, g.level, unk_partner, oi, (uint32_t)gates.size()});
        } else p.n_fast_gates++;
        gates.push_back(std::move(g));
        return true;
    }

    void fuse_gate_pairs() {
    // =========================================================================== gate pairs / wave programs
    // A gate whose only operand from the previous level is the output of a SOLVE gate, and whose other operands are older than
    // that gate's whole wave, runs BEHIND it in the same wave: the intermediate witness comes from registers (`local`) instead
    // of HBM (it is still written: it is a witness). A wave program is the host and up to max_tails records. The records read
    // `local`: the host's output at first; a record may take `local` over (GATE_SETLOCAL_FLAG) when its first consumer is
    // appended directly behind it, so chains host -> tail -> tail's consumer run in one wave as well (the records that read the
    // previous owner are all in front of it by then).
    for (uint32_t gi = 0; gi < gates.size(); gi++) {
        gates[gi].run_level = gates[gi].level;
        gates[gi].owner = gates[gi].last_gate = gi;
    }
    if (tune.pairs) {
        const uint32_t max_tails = (uint32_t)std::min<int64_t>(std::max<int64_t>(tune.max_tails, 0), 64);
        const bool chains = tune.chains != 0;
        std::vector<uint32_t> producer_gate(nw, 0xFFFFFFFFu), root_of(gates.size(), 0xFFFFFFFFu);
        for (uint32_t gi = 0; gi < gates.size(); gi++)
            if ((gates[gi].words[0] & 0xff) == GATE_SOLVE) producer_gate[gates[gi].words[2]] = gi;
        for (uint32_t ci = 0; ci < gates.size(); ci++) {
            PendingGate &cg = gates[ci];
            const uint32_t kind = cg.words[0] & 0xff;
            if (kind == GATE_SOLVE_DYN || cg.level < 2) continue;
            uint32_t crit = 0xFFFFFFFFu, n_crit = 0;
            for (uint32_t w : cg.reads)
                if (level[w] + 1 == cg.level) { crit = w; n_crit++; }
            if (n_crit != 1) continue;
            const uint32_t pi = producer_gate[crit];
            if (pi == 0xFFFFFFFFu || pi == ci) continue;
            const uint32_t hi = gates[pi].fused ? root_of[pi] : pi;
            if (hi != pi && !chains) continue;
            PendingGate &hg = gates[hi];
            if (hg.n_tails >= max_tails) continue;
            bool older = true;  // everything else must be known before the wave's level runs
            for (uint32_t w : cg.reads)
                if (w != crit && level[w] + 1 > hg.level) older = false;
            if (!older) continue;
            if (pi != hg.owner) {
                if (pi != hg.last_gate) continue;  // a record in between still reads the current `local`
                hg.words[hg.last_w0] |= GATE_SETLOCAL_FLAG;
                hg.owner = pi;
            }
            cg.fused = true;
            cg.run_level = hg.level;
            root_of[ci] = hi;
            for_each_operand_word(cg.words, [&](uint32_t &slot) { if (slot == crit) slot = GATE_LOCAL; });
            hg.words[hg.last_w0] |= GATE_TAIL_FLAG;
            hg.last_w0 = hg.words.size();
            hg.words.insert(hg.words.end(), cg.words.begin(), cg.words.end());
            hg.last_gate = ci;
            hg.n_tails++;
            p.n_gate_pairs++;
        }
    }
    }

    void assign_inverse_slots() {
    // =========================================================================== inverse slots: a slot is reused once its gate ran
    std::stable_sort(inverses.begin(), inverses.end(), [](const PendingInverse &a, const PendingInverse &b) { return a.level < b.level; });
    {
        std::priority_queue<std::pair<uint32_t, uint32_t>, std::vector<std::pair<uint32_t, uint32_t>>, std::greater<>> busy;  // (use_level, slot)
        std::vector<uint32_t> free_slots;
        for (auto &iv : inverses) {
            while (!busy.empty() && busy.top().first < iv.level) { free_slots.push_back(busy.top().second); busy.pop(); }
            uint32_t slot;
            if (!free_slots.empty()) { slot = free_slots.back(); free_slots.pop_back(); }
            else slot = p.n_inverse_slots++;
            busy.push({iv.use_level, slot});
            gates[iv.gate].words[4] = slot;
            iv.gate = slot;  // from here on: the slot
        }
    }
    }

    void fold_digest_leaves() {
    // =========================================================================== digest leaves folded into the solve
    // wdef[w]: the level whose launches write w (a fused gate writes in its host's wave; 0 = initial witness)
    wdef.assign(nw, 0);
    for (auto &g : gates)
        if ((g.words[0] & 0xff) != GATE_ASSERT) wdef[g.words[2]] = g.run_level;
    for (auto &r : records)
        for (auto &slot : out_slots[r.opcode])
            if (p.producer[slot.second] == r.opcode) wdef[slot.second] = r.level;
    if (opts.fold_digest && p.truncated_at == 0xFFFFFFFFu) {
        // record: [PK_DIGEST_LEAF, row of the partial-sum table, n, (witness, row of `unscale` or NONE) x n]: the terms value_w * g^(w+1) of the digest
        // (include/acvm_amd.h acvm_batch_digest: a polynomial fingerprint, one product per witness, order-free) of the witnesses that are
        // complete, for the generic instance. Launched in epochs (every digest_epoch-th level all witnesses completed since the last one, in
        // records of at most 256): nothing waits for anything but its own witness, and a row can be recycled as soon as the epoch of its
        // witness has run.
        const uint32_t K_dig = (uint32_t)std::max<int64_t>(tune.digest_epoch, 1);
        const uint32_t PER_RECORD = 256;
        std::map<uint32_t, std::vector<uint32_t>> by_level;  // epoch level -> witnesses
        for (uint32_t w = 0; w < nw; w++) {
            if (p.producer[w] == 0xFFFFFFFFu) continue;
            const uint32_t lvl = std::max(wdef[w] + 1, 1u);
            by_level[(lvl + K_dig - 1) / K_dig * K_dig].push_back(w);
        }
        for (auto &kv : by_level)
            for (size_t at = 0; at < kv.second.size(); at += PER_RECORD) {
                PendingRecord r;
                r.cls = CLS_DIGEST;
                r.synthetic = true;
                r.opcode = (uint32_t)p.prog.size();  // offset of the record: digest records have no opcode
                r.level = kv.first;
                const size_t n = std::min<size_t>(PER_RECORD, kv.second.size() - at);
                p.prog.insert(p.prog.end(), {PK_DIGEST_LEAF, p.n_digest_segments, (uint32_t)n});
                for (size_t k = 0; k < n; k++) {
                    const uint32_t w = kv.second[at + k];
                    p.prog.insert(p.prog.end(), {w, p.unscale_index[w]});
                    r.reads.push_back(w);
                }
                records.push_back(std::move(r));
                p.n_digest_segments++;
            }
    }
    }

    void fuse_range_checks() {
    // =========================================================================== byte RANGE checks fused into the hash that reads the byte
    // RANGE(w, <= 8 bits) on an input of a byte-message hash (the usual shape: every message byte is range-checked) reads the row the hash
    // kernel reads anyway and needs the low limb it forms anyway: the level schedule runs a copy of the hash record extended by
    // (RANGE opcode, bits) per input (function word | PLAN_HASH_RANGE_FLAG) and drops the RANGE record; a failing check flags the instance
    // with the RANGE opcode's index, exactly as its own record would. A check may move to a later level this way (nothing reads a RANGE).
    if (tune.range_fuse) {
        std::unordered_map<uint32_t, std::vector<size_t>> range_of;  // witness -> RANGE records (bits <= 8) not yet fused
        for (size_t i = 0; i < records.size(); i++) {
            const PendingRecord &r = records[i];
            if (r.synthetic || r.cls != CLS_LIGHT) continue;
            const uint32_t *rec = &p.prog[p.prog_offset[r.opcode]];
            if (rec[0] == PK_RANGE && rec[3] <= 8u) range_of[rec[2]].push_back(i);
        }
        std::vector<size_t> hashes;
        for (size_t i = 0; i < records.size(); i++)
            if (!records[i].synthetic && records[i].cls == CLS_HASH && (p.prog[p.prog_offset[records[i].opcode] + 2] & PLAN_HASH_COOP_FLAG)) hashes.push_back(i);
        std::sort(hashes.begin(), hashes.end(), [&](size_t a, size_t b) { return records[a].level != records[b].level ? records[a].level < records[b].level : a < b; });
        std::vector<uint8_t> drop(records.size(), 0);
        bool any_drop = false;
        for (size_t hi : hashes) {
            PendingRecord &h = records[hi];
            const size_t at = p.prog_offset[h.opcode];
            const uint32_t n_in = p.prog[at + 3], n_out = p.prog[at + 4];
            std::vector<uint32_t> ext(2 * (size_t)n_in, 0xFFFFFFFFu);
            bool any = false;
            for (uint32_t i = 0; i < n_in; i++) {
                auto it = range_of.find(p.prog[at + 6 + 2 * i]);
                if (it == range_of.end()) continue;
                for (size_t &ri : it->second) {
                    if (ri == (size_t)-1 || records[ri].level > h.level) continue;  // (a check waits for its witness like the hash does)
                    const uint32_t *rr = &p.prog[p.prog_offset[records[ri].opcode]];
                    ext[2 * i] = rr[1];
                    ext[2 * i + 1] = rr[3];
                    drop[ri] = 1;
                    ri = (size_t)-1;
                    any = any_drop = true;
                    break;  // one check per input slot; a second RANGE on the same witness keeps its own record (or the next slot that reads it)
                }
            }
            if (!any) continue;
            const size_t len = 6 + 2 * (size_t)n_in + 2 * (size_t)n_out;
            h.prog_at = (uint32_t)p.prog.size();
            std::vector<uint32_t> copy(p.prog.begin() + at, p.prog.begin() + at + len);
            copy[2] |= PLAN_HASH_RANGE_FLAG;
            p.prog.insert(p.prog.end(), copy.begin(), copy.end());
            p.prog.insert(p.prog.end(), ext.begin(), ext.end());
        }
        if (any_drop) {
            std::vector<PendingRecord> kept;
            kept.reserve(records.size());
            for (size_t i = 0; i < records.size(); i++)
                if (!drop[i]) kept.push_back(std::move(records[i]));
            records.swap(kept);
        }
    }
    }

    void chain_hashes() {
    // =========================================================================== hash chains
    // A byte-message hash B whose inputs are the digest of another byte-message hash A plus witnesses that were already known when A was
    // launched (a hash of a hash; the links of a Merkle path) runs behind A in A's workgroup (kernels_hash.hip): the level schedule gets a
    // copy of A's record with PLAN_HASH_CHAIN_FLAG and one more word, the offset of a link [offset of B's record, per input of B: the byte
    // of A's digest it is, or NONE]; B itself leaves the level lists (its outputs count as written by A's launch). B may carry a link of its
    // own. Only the previous digest is at hand in the block, so B reads nothing of an earlier member of its chain.
    if (tune.hash_chain) {
        auto rec_at = [&](const PendingRecord &r) { return (size_t)(r.prog_at != 0xFFFFFFFFu ? r.prog_at : p.prog_offset[r.opcode]); };
        auto is_coop = [&](const PendingRecord &r) { return !r.synthetic && r.cls == CLS_HASH && (p.prog[rec_at(r) + 2] & PLAN_HASH_COOP_FLAG) && p.prog[rec_at(r) + 4] == 32u; };
        std::unordered_map<uint32_t, std::pair<size_t, uint32_t>> digest_byte;  // witness -> (record, index of the output)
        std::vector<size_t> hashes;
        for (size_t i = 0; i < records.size(); i++) {
            if (!is_coop(records[i])) continue;
            hashes.push_back(i);
            const size_t at = rec_at(records[i]);
            const uint32_t n_in = p.prog[at + 3];
            for (uint32_t k = 0; k < 32u; k++) {
                const uint32_t w = p.prog[at + 6 + 2 * n_in + 2 * k];
                if (p.producer[w] == records[i].opcode) digest_byte.emplace(w, std::make_pair(i, k));  // (an output that only compares is no source)
            }
        }
        std::sort(hashes.begin(), hashes.end(), [&](size_t a, size_t b) { return records[a].level != records[b].level ? records[a].level < records[b].level : a < b; });
        const size_t NO = (size_t)-1;
        std::vector<size_t> next(records.size(), NO), prev(records.size(), NO);
        std::vector<uint32_t> launch_level(records.size(), 0);
        for (size_t i : hashes) launch_level[i] = records[i].level;
        for (size_t bi : hashes) {
            const size_t at = rec_at(records[bi]);
            const uint32_t n_in = p.prog[at + 3];
            size_t a = NO;
            bool ok = true;
            for (uint32_t k = 0; k < n_in && ok; k++) {
                auto it = digest_byte.find(p.prog[at + 6 + 2 * k]);
                if (it == digest_byte.end() || it->second.first == bi) continue;
                const size_t cand = it->second.first;
                if (records[cand].level >= records[bi].level) continue;  // not a producer of this input on the schedule (then the check below refuses)
                if (a == NO) a = cand;
                else if (a != cand && records[cand].level > records[a].level) a = cand;  // the youngest producer is the only possible predecessor
            }
            if (a == NO || next[a] != NO) continue;
            const uint32_t head_level = launch_level[a];
            for (uint32_t k = 0; k < n_in && ok; k++) {
                const uint32_t w = p.prog[at + 6 + 2 * k];
                auto it = digest_byte.find(w);
                if (it != digest_byte.end() && it->second.first == a) continue;
                ok = level[w] + 1 <= head_level;  // known before the launch of the head: readable from the table like the head's own inputs
            }
            for (uint32_t w : records[bi].reads) {  // (compared outputs and whatever else the record reads)
                auto it = digest_byte.find(w);
                if (it != digest_byte.end() && it->second.first == a) continue;
                if (p.producer[w] == records[bi].opcode) continue;
                ok = ok && level[w] + 1 <= head_level;
            }
            if (!ok) continue;
            next[a] = bi;
            prev[bi] = a;
            launch_level[bi] = head_level;
        }
        std::vector<uint8_t> drop(records.size(), 0);
        bool any = false;
        for (size_t hi : hashes) {
            if (prev[hi] != NO || next[hi] == NO) continue;  // heads of chains only
            std::vector<size_t> chain;
            for (size_t m = hi; m != NO; m = next[m]) chain.push_back(m);
            uint32_t link_at = 0xFFFFFFFFu;  // offset of the link TO the member being emitted (from the tail backwards)
            for (size_t c = chain.size(); c-- > 0;) {
                PendingRecord &m = records[chain[c]];
                const size_t at = rec_at(m);
                const uint32_t n_in = p.prog[at + 3];
                const size_t len = 6 + 2 * (size_t)n_in + 64 + ((p.prog[at + 2] & PLAN_HASH_RANGE_FLAG) ? 2 * (size_t)n_in : 0);
                uint32_t copy_at = (uint32_t)at;
                if (c + 1 < chain.size()) {  // a successor follows: copy with the flag and the offset of its link
                    std::vector<uint32_t> copy(p.prog.begin() + at, p.prog.begin() + at + len);
                    copy[2] |= PLAN_HASH_CHAIN_FLAG;
                    copy.push_back(link_at);
                    copy_at = (uint32_t)p.prog.size();
                    p.prog.insert(p.prog.end(), copy.begin(), copy.end());
                }
                if (c > 0) {  // the link to this member: [its record, source of each input]
                    const size_t pa = chain[c - 1];
                    link_at = (uint32_t)p.prog.size();
                    p.prog.push_back(copy_at);
                    std::vector<uint32_t> from_rows;  // the inputs that are NOT bytes of the predecessor's digest: readable while the predecessor is hashed
                    for (uint32_t k = 0; k < n_in; k++) {
                        auto it = digest_byte.find(p.prog[at + 6 + 2 * k]);
                        const bool chained_in = it != digest_byte.end() && it->second.first == pa;
                        p.prog.push_back(chained_in ? it->second.second : 0xFFFFFFFFu);
                        if (!chained_in) from_rows.push_back(k);
                    }
                    p.prog.push_back((uint32_t)from_rows.size());
                    p.prog.insert(p.prog.end(), from_rows.begin(), from_rows.end());
                    drop[chain[c]] = 1;
                    any = true;
                    for (uint32_t k = 0; k < 32u; k++) {  // its outputs appear with the head's launch
                        const uint32_t w = p.prog[at + 6 + 2 * n_in + 2 * k];
                        if (p.producer[w] == m.opcode) heavy_level[w] = records[hi].level;
                    }
                } else m.prog_at = copy_at;
            }
            p.n_hash_chained += (uint32_t)chain.size() - 1;
        }
        if (any) {
            // the chained records stay in `records` (their reads and outputs count in the dependency tables below) at the head's level, but not in the level lists
            for (size_t i = 0; i < records.size(); i++)
                if (drop[i]) { records[i].level = launch_level[i]; records[i].chained = true; }
        }
    }
    }

    void merge_range_records() {
    // =========================================================================== RANGE opcodes of a level, eight to a record
    // A RANGE check is one row and a few dozen instructions: launched one lane per (opcode, instance) the kernel is bound by the chain of
    // dependent latencies every wave pays before its only load (config 3: 96 checks per instance). Merged records
    // [PK_RANGE_MULTI, first opcode, n, (opcode, witness, num_bits) x n] keep four rows in flight per lane. The exact path keeps the
    // opcode's own record.
    if (tune.range_merge) {
        std::map<uint32_t, std::vector<size_t>> by_level;
        for (size_t i = 0; i < records.size(); i++)
            if (!records[i].synthetic && records[i].cls == CLS_LIGHT && p.prog[p.prog_offset[records[i].opcode]] == PK_RANGE) by_level[records[i].level].push_back(i);
        std::vector<uint8_t> drop(records.size(), 0);
        std::vector<PendingRecord> merged;
        constexpr size_t RANGES_PER_RECORD = 8;
        for (auto &kv : by_level) {
            if (kv.second.size() < 2) continue;
            for (size_t at = 0; at < kv.second.size(); at += RANGES_PER_RECORD) {
                const size_t n = std::min(RANGES_PER_RECORD, kv.second.size() - at);
                PendingRecord m;
                m.level = kv.first;
                m.cls = CLS_LIGHT;
                m.synthetic = true;
                m.opcode = (uint32_t)p.prog.size();
                p.prog.insert(p.prog.end(), {PK_RANGE_MULTI, records[kv.second[at]].opcode, (uint32_t)n});
                for (size_t k = 0; k < n; k++) {
                    const PendingRecord &r = records[kv.second[at + k]];
                    const uint32_t *rec = &p.prog[p.prog_offset[r.opcode]];  // [PK_RANGE, opcode, witness, num_bits]
                    const uint32_t w = rec[2], bits = rec[3];
                    p.prog.insert(p.prog.end(), {r.opcode, w, bits});
                    m.reads.insert(m.reads.end(), r.reads.begin(), r.reads.end());
                    drop[kv.second[at + k]] = 1;
                }
                merged.push_back(std::move(m));
            }
        }
        if (!merged.empty()) {
            std::vector<PendingRecord> kept;
            kept.reserve(records.size());
            for (size_t i = 0; i < records.size(); i++)
                if (!drop[i]) kept.push_back(std::move(records[i]));
            for (auto &m : merged) kept.push_back(std::move(m));
            records.swap(kept);
        }
    }
    }

    void order_and_dependencies() {
    // =========================================================================== order by (level, program order), lay out
    // within a level the longest wave programs (hosts with tails, many terms) go first: blocks are dispatched in grid order,
    // and a level ends when its last wave ends
    // (config 2: 15.6 -> 15.5 ms per solve)
    std::stable_sort(gates.begin(), gates.end(), [](const PendingGate &a, const PendingGate &b) {
        if (a.level != b.level) return a.level < b.level;
        return a.words.size() > b.words.size();
    });
    std::stable_sort(records.begin(), records.end(), [](const PendingRecord &a, const PendingRecord &b) { return a.level < b.level; });
    max_level = 0;
    for (auto &g : gates) max_level = std::max(max_level, g.level);
    for (auto &r : records) max_level = std::max(max_level, r.level);
    for (auto &iv : inverses) max_level = std::max(max_level, iv.level);
    p.n_levels = max_level;
    p.level_start.assign(max_level + 1, 0);
    p.level_needs_inverse.assign(max_level + 1, 0);
    for (auto &iv : inverses) p.level_needs_inverse[iv.use_level] = std::max(p.level_needs_inverse[iv.use_level], iv.level);
    // heavy record classes run on a third stream beside the levels (batch.cpp): a level of the main stream (gates, light
    // records) or an inversion batch waits for the heavy records of level h only if it reads one of their outputs
    for (int q = 0; q < N_HEAVY_LANES; q++) {
        p.level_needs_heavy[q].assign(max_level + 1, 0);
        p.inv_needs_heavy[q].assign(max_level + 1, 0);
        for (int q2 = 0; q2 < N_HEAVY_LANES; q2++) p.lane_needs_lane[q][q2].assign(max_level + 1, 0);
        p.lane_needs_main[q].assign(max_level + 1, 0);
    }
    auto needs = [&](std::vector<uint32_t> (&tab)[N_HEAVY_LANES], uint32_t at, uint32_t w) {
        if (wlane[w]) tab[wlane[w] - 1][at] = std::max(tab[wlane[w] - 1][at], heavy_level[w]);
    };
    for (auto &g : gates) {
        const uint32_t eff = g.run_level;  // a tail runs in its host's wave, at least one level early
        for (uint32_t w : g.reads) needs(p.level_needs_heavy, eff, w);
    }
    for (auto &hr : heavy_reads) needs(p.level_needs_heavy, hr.first, hr.second);
    for (auto &iv : inverses) needs(p.inv_needs_heavy, iv.level, iv.partner);
    for (auto &r : records) {
        if (!is_heavy(r.cls)) continue;
        for (uint32_t w : r.reads) {
            if (wlane[w]) {
                    if (wlane[w] - 1 != heavy_lane(r.cls)) needs(p.lane_needs_lane[heavy_lane(r.cls)], r.level, w);
                } else if (p.producer[w] != 0xFFFFFFFFu) {  // written by a gate or a light record (an initial witness is there before the solve)
                    uint32_t &m = p.lane_needs_main[heavy_lane(r.cls)][r.level];
                    m = std::max(m, wdef[w]);
                }
        }
    }
    }

    // false: the circuit cannot run with recycled rows (p.unsupported says why)
    bool assign_rows() {
    // =========================================================================== witness-slot liveness reuse
    // A witness occupies a row of the table from the level that writes it to the level of its last reader; the rows of dead
    // witnesses go to a FIFO and are handed to the witnesses the MAIN stream defines later (gates, light records). The main stream
    // runs its levels in order, so a recycled row is safe against every earlier reader on that stream; a reader (or an asynchronous
    // writer: the outputs of heavy records) on another stream -- inversion batches, the heavy lanes, the digest leaves -- becomes one
    // more dependency of the level that recycles the row (level_needs_inverse / level_needs_heavy: usually long satisfied, the FIFO
    // keeps a row out of use for as long as it can). Heavy outputs always take fresh rows (their lanes have no edge from the other
    // asynchronous streams), the initial witnesses and `keep` are never released. The digest leaves are readers like any other:
    // a row is hashed before it is recycled.
    if (opts.reuse_slots) {
        if (p.truncated_at != 0xFFFFFFFFu || p.has_foreign_calls) {
            p.unsupported = "slot reuse needs a circuit the level kernels cover entirely and no foreign calls";
            return false;
        }
        constexpr int N_ASYNC = 1 + N_HEAVY_LANES;  // 0 = inversion batches, 1 + q = heavy lane q
        std::vector<uint32_t> last_any(nw, 0), last_on[N_ASYNC];
        for (auto &v : last_on) v.assign(nw, 0);
        auto note = [&](uint32_t w, uint32_t lvl, int async) {  // async < 0: the main stream
            last_any[w] = std::max(last_any[w], lvl);
            if (async >= 0) last_on[async][w] = std::max(last_on[async][w], lvl);
        };
        std::vector<std::vector<uint32_t>> def_main(max_level + 1), def_heavy(max_level + 1);
        for (auto &g : gates) {
            for (uint32_t w : g.reads) note(w, g.run_level, -1);
            if ((g.words[0] & 0xff) != GATE_ASSERT) def_main[g.run_level].push_back(g.words[2]);
        }
        for (auto &iv : inverses) note(iv.partner, iv.level, 0);
        for (auto &r : records) {
            const bool heavy = is_heavy(r.cls);
            const int async = heavy ? 1 + heavy_lane(r.cls) : -1;
            for (uint32_t w : r.reads) note(w, r.level, async);
            if (r.synthetic) continue;
            for (auto &slot : out_slots[r.opcode])
                if (p.producer[slot.second] == r.opcode) {
                    (heavy ? def_heavy : def_main)[r.level].push_back(slot.second);
                    if (heavy) note(slot.second, r.level, async);  // written asynchronously: recycling the row waits for the write too
                }
        }
        std::vector<uint8_t> keep(nw, 0);
        for (uint32_t i = 0; i < n_initial; i++) keep[initial_ids[i]] = 1;
        for (uint32_t w : opts.keep)
            if (w < nw) keep[w] = 1;
        p.slot_of.assign(nw, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < n_initial; i++) p.slot_of[initial_ids[i]] = p.n_slots++;
        std::vector<std::vector<uint32_t>> release(max_level + 2);
        for (uint32_t w = 0; w < nw; w++)
            if (p.producer[w] != 0xFFFFFFFFu && !keep[w]) release[std::min<uint32_t>(std::max(last_any[w], wdef[w]) + 1, max_level + 1)].push_back(w);
        std::vector<uint32_t> fifo, prev_owner;
        size_t head = 0;
        for (uint32_t L = 1; L <= max_level; L++) {
            for (uint32_t w : release[L]) {
                const uint32_t row = p.slot_of[w];
                if (row == 0xFFFFFFFFu) continue;
                if (prev_owner.size() <= row) prev_owner.resize(p.n_slots, 0xFFFFFFFFu);
                prev_owner[row] = w;
                fifo.push_back(row);
            }
            for (uint32_t w : def_main[L]) {
                if (p.slot_of[w] != 0xFFFFFFFFu) continue;
                if (head < fifo.size()) {
                    const uint32_t row = fifo[head++], old = prev_owner[row];
                    if (last_on[0][old]) p.level_needs_inverse[L] = std::max(p.level_needs_inverse[L], last_on[0][old]);
                    for (int q = 0; q < N_HEAVY_LANES; q++)
                        if (last_on[1 + q][old]) p.level_needs_heavy[q][L] = std::max(p.level_needs_heavy[q][L], last_on[1 + q][old]);
                    p.slot_of[w] = row;
                } else p.slot_of[w] = p.n_slots++;
            }
            for (uint32_t w : def_heavy[L])
                if (p.slot_of[w] == 0xFFFFFFFFu) p.slot_of[w] = p.n_slots++;
        }
        // the gate programs and the inversion jobs address rows
        auto row = [&](uint32_t w) { return w == GATE_LOCAL ? GATE_LOCAL : p.slot_of[w]; };
        for (auto &g : gates) {
            if (g.fused) continue;
            std::vector<uint32_t> &w = g.words;
            for (size_t pos = 0;;) {
                const uint32_t w0 = w[pos], w5 = w[pos + 5];
                const uint32_t np_mac = (w0 >> 8) & 0xff, nl_mac = (w0 >> 16) & 0xff;
                const uint32_t np_pos = w5 & 0xff, np_neg = (w5 >> 8) & 0xff, nl_pos = (w5 >> 16) & 0xff, nl_neg = w5 >> 24;
                if ((w0 & 0xff) != GATE_ASSERT) w[pos + 2] = row(w[pos + 2]);
                size_t t = pos + 6;
                for (uint32_t i = 0; i < np_mac; i++, t += 10) { w[t + 8] = row(w[t + 8]); w[t + 9] = row(w[t + 9]); }
                for (uint32_t i = 0; i < nl_mac; i++, t += 9) w[t + 8] = row(w[t + 8]);
                for (uint32_t i = 0; i < 2 * (np_pos + np_neg) + nl_pos + nl_neg; i++, t++) w[t] = row(w[t]);
                if (!(w0 & GATE_TAIL_FLAG)) break;
                pos = t;
            }
        }
        for (auto &iv : inverses) iv.partner = p.slot_of[iv.partner];
    }
        return true;
    }

    void lay_out() {
    p.dyn_level_start.assign(max_level + 1, 0);
    for (int k = 0; k < N_CLS; k++) p.cls_level_start[k].assign(max_level + 1, 0);
    size_t gi = 0, ri = 0, ii = 0;
    std::vector<uint32_t> width(max_level + 1, 0);
    for (uint32_t L = 1; L <= max_level; L++) {
        p.level_start[L - 1] = (uint32_t)p.gate_offset.size();
        p.dyn_level_start[L - 1] = (uint32_t)p.dyn_offset.size();
        for (int k = 0; k < N_CLS; k++) p.cls_level_start[k][L - 1] = (uint32_t)p.cls_offset[k].size();
        for (; gi < gates.size() && gates[gi].level == L; gi++) {
            if (gates[gi].fused) continue;  // emitted behind its host
            p.gate_offset.push_back((uint32_t)p.gate_stream.size());
            p.gate_stream.insert(p.gate_stream.end(), gates[gi].words.begin(), gates[gi].words.end());
            width[L]++;
        }
        for (; ii < inverses.size() && inverses[ii].level == L; ii++) {  // inversion job: [denominator witness, opcode, inverse slot]
            p.dyn_offset.push_back((uint32_t)p.gate_stream.size());
            p.gate_stream.insert(p.gate_stream.end(), {inverses[ii].partner, inverses[ii].opcode, inverses[ii].gate});
        }
        for (; ri < records.size() && records[ri].level == L; ri++) {
            const PendingRecord &r = records[ri];
            if (r.chained) continue;
            p.cls_offset[r.cls].push_back(r.prog_at != 0xFFFFFFFFu ? r.prog_at : r.synthetic ? r.opcode : p.prog_offset[r.opcode]);
            // (a Pedersen record of the level schedule parks its step sums for the shared inversion: kernels_grumpkin.hip pedersen_bundle_level_kernel)
            // (a straight-line Brillig record on the Brillig lane needs none of its VM record's scratch)
            p.cls_scratch[r.cls].push_back(r.synthetic || p.prog[p.cls_offset[r.cls].back()] == PK_BRILLIG_SL ? 0u : r.cls == CLS_PEDERSEN ? PEDERSEN_PARK_WORDS : p.prog_scratch[r.opcode]);
            width[L]++;
        }
    }
    p.level_start[max_level] = (uint32_t)p.gate_offset.size();
    p.dyn_level_start[max_level] = (uint32_t)p.dyn_offset.size();
    for (int k = 0; k < N_CLS; k++) p.cls_level_start[k][max_level] = (uint32_t)p.cls_offset[k].size();
    for (uint32_t L = 1; L <= max_level; L++) p.max_level_width = std::max(p.max_level_width, width[L]);
    }

    // =========================================================================== what each pass promises (tuning plan_validate; tests/test_plan_host.py)
    // Cheap structural invariants, one check per pass, thrown as std::logic_error (the ABI turns it into ACVM_E_INVALID with this text). The
    // semantic end-to-end check of the same plan is the hazard checker (schedule_check.cpp). plan_validate = 100 + k breaks invariant k on
    // purpose first, so that a test can see each check fire.
    [[noreturn]] void violated(const char *pass, const std::string &what) const {
        throw std::logic_error(std::string("plan invariant violated after pass ") + pass + ": " + what);
    }
    bool inject(int k) const { return tune.plan_validate == 100 + k; }
    void check_in_order_program() const {
        const char *P = "emit_in_order_program";
        if (p.prog_offset.size() != c.opcodes.size()) violated(P, "one record per opcode");
        for (size_t oi = 0; oi < p.prog_offset.size(); oi++) {
            const size_t at = p.prog_offset[oi];
            if (at + 2 > p.prog.size() || (oi && at <= p.prog_offset[oi - 1])) violated(P, "record offsets are not increasing at opcode " + std::to_string(oi));
            if (p.prog[at] > PK_BRILLIG_SL || p.prog[at] == PK_DIGEST_LEAF || p.prog[at] == PK_RANGE_MULTI || p.prog[at] == PK_BRILLIG_SL) violated(P, "opcode " + std::to_string(oi) + " has a record kind of the level schedule only");
            if (p.prog[at + 1] != oi) violated(P, "record of opcode " + std::to_string(oi) + " names another opcode");
            if (p.prog_class[oi] >= N_CLS) violated(P, "class out of range");
            for (auto &slot : out_slots[oi])
                if (slot.first == 0 || slot.first >= p.prog.size() || p.prog[slot.first - 1] != slot.second || p.prog[slot.first] > 1) violated(P, "output slot of opcode " + std::to_string(oi) + " does not sit behind its witness");
        }
    }
    void check_replay() const {
        const char *P = "replay";
        for (size_t gi = 0; gi < gates.size(); gi++) {
            const PendingGate &g = gates[gi];
            if (g.level < 1) violated(P, "a gate at level 0");
            for (uint32_t w : g.reads)
                if (w >= nw || !known[w] || level[w] >= g.level) violated(P, "gate of opcode " + std::to_string(g.words[1]) + " reads witness " + std::to_string(w) + " before the main stream may");
            if ((g.words[0] & 0xff) != GATE_ASSERT && level[g.words[2]] != g.level) violated(P, "a gate's output is not defined at the gate's level");
        }
        for (const PendingRecord &r : records) {
            const std::vector<uint32_t> &lv = is_heavy(r.cls) ? hlevel : level;
            for (uint32_t w : r.reads)
                if (w >= nw || !known[w] || (p.producer[w] != r.opcode && lv[w] >= r.level)) violated(P, "record of opcode " + std::to_string(r.opcode) + " reads witness " + std::to_string(w) + " before its stream may");
        }
        for (const PendingInverse &iv : inverses)
            if (iv.level <= level[iv.partner] || iv.use_level <= iv.level) violated(P, "inversion of opcode " + std::to_string(iv.opcode) + " is not between its denominator and its gate");
        for (uint32_t w = 0; w < nw; w++) {
            if (p.kbound[w] > GATE_K_ROW_MAX) violated(P, "witness " + std::to_string(w) + " may pass 2^256");
            if (is_scaled[w] && pinned[w]) violated(P, "pinned witness " + std::to_string(w) + " carries a scale");
            if (is_scaled[w] && !(frh::mul(ws[scale_slot[w]], wsi[scale_slot[w]]) == f_one)) violated(P, "scale x 1 / scale != 1 for witness " + std::to_string(w));
            if (p.kbound[w] != GATE_K_CANON && !is_scaled[w]) violated(P, "relaxed witness " + std::to_string(w) + " has no unscale entry");
        }
        if (p.truncated_at != 0xFFFFFFFFu)
            for (uint32_t w = 0; w < nw; w++)
                if (p.producer[w] < 0xFFFFFFFEu && p.producer[w] >= p.truncated_at) violated(P, "a witness is produced behind the truncation");
    }
    void check_pairs() {
        const char *P = "fuse_gate_pairs";
        if (inject(1) && gates.size() > 1) { gates[1].fused = true; gates[1].run_level = gates[1].level; }
        size_t n_fused = 0, n_tails = 0;
        for (const PendingGate &g : gates) {
            if (g.fused) {
                n_fused++;
                if (g.run_level >= g.level) violated(P, "a fused gate does not run before its own level");
                if ((g.words[0] & 0xff) == GATE_SOLVE_DYN) violated(P, "a gate with an inversion runs as a tail");
                continue;
            }
            if (g.run_level != g.level) violated(P, "a host runs off its level");
            n_tails += g.n_tails;
            size_t pos = 0, n_rec = 0;
            for (;;) {  // the wave program: 1 + n_tails records chained by GATE_TAIL_FLAG, the forwarded operand only behind the first
                if (pos + 6 > g.words.size()) violated(P, "a wave program runs past its words");
                const uint32_t w0 = g.words[pos], w5 = g.words[pos + 5];
                const size_t len = 6 + 10 * (size_t)((w0 >> 8) & 0xff) + 9 * (size_t)((w0 >> 16) & 0xff) + 2 * (size_t)((w5 & 0xff) + ((w5 >> 8) & 0xff)) + ((w5 >> 16) & 0xff) + (w5 >> 24);
                if (pos + len > g.words.size()) violated(P, "a wave program runs past its words");
                if (n_rec == 0)
                    for (size_t t = pos + 6; t < pos + len; t++) {
                        // (coefficient words may hold any value: only operand positions count)
                    }
                n_rec++;
                if (!(w0 & GATE_TAIL_FLAG)) { if (pos + len != g.words.size()) violated(P, "words behind the last record of a wave program"); break; }
                pos += len;
            }
            if (n_rec != 1 + g.n_tails) violated(P, "a wave program of " + std::to_string(n_rec) + " records with " + std::to_string(g.n_tails) + " tails");
        }
        if (n_fused != n_tails || n_fused != p.n_gate_pairs) violated(P, "fused gates and tails do not add up");
    }
    void check_inverse_slots() {
        const char *P = "assign_inverse_slots";
        if (inject(2) && inverses.size() > 1) inverses[1].gate = inverses[0].gate;
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> by_slot(p.n_inverse_slots);
        for (const PendingInverse &iv : inverses) {
            if (iv.gate >= p.n_inverse_slots) violated(P, "slot out of range");
            by_slot[iv.gate].push_back({iv.level, iv.use_level});
        }
        for (auto &v : by_slot) {
            std::sort(v.begin(), v.end());
            for (size_t i = 1; i < v.size(); i++)
                if (v[i].first <= v[i - 1].second) violated(P, "two inverses share a row of the table while both are alive");
        }
    }
    void check_digest_leaves() const {
        const char *P = "fold_digest_leaves";
        if (!(opts.fold_digest && p.truncated_at == 0xFFFFFFFFu)) { if (p.n_digest_segments) violated(P, "leaves without the option"); return; }
        std::vector<uint8_t> seen(nw, 0);
        uint32_t n_rec = 0;
        for (const PendingRecord &r : records) {
            if (r.cls != CLS_DIGEST) continue;
            n_rec++;
            for (uint32_t w : r.reads) {
                if (seen[w]++) violated(P, "witness " + std::to_string(w) + " is summed twice");
                if (r.level <= wdef[w]) violated(P, "a leaf runs before its witness is written");
            }
        }
        for (uint32_t w = 0; w < nw; w++)
            if ((p.producer[w] != 0xFFFFFFFFu) != (seen[w] != 0)) violated(P, "witness " + std::to_string(w) + " is assigned but not summed (or the reverse)");
        if (n_rec != p.n_digest_segments) violated(P, "leaf records and rows of the partial sums differ");
    }
    void check_range_accounting() const {
        // every RANGE opcode the generic instance executes is checked exactly once: by its own record, inside the hash that reads its byte, or in a merged record
        const char *P = "merge_range_records";
        std::vector<uint8_t> times(c.opcodes.size(), 0);
        for (const PendingRecord &r : records) {
            const size_t at = r.prog_at != 0xFFFFFFFFu ? r.prog_at : r.synthetic ? r.opcode : p.prog_offset[r.opcode];
            if (p.prog[at] == PK_RANGE) times[p.prog[at + 1]]++;
            else if (p.prog[at] == PK_RANGE_MULTI)
                for (uint32_t i = 0; i < p.prog[at + 2]; i++) times[p.prog[at + 3 + 3 * i]]++;
            else if (p.prog[at] == PK_HASH && (p.prog[at + 2] & PLAN_HASH_RANGE_FLAG)) {
                const uint32_t n_in = p.prog[at + 3], n_out = p.prog[at + 4];
                for (uint32_t i = 0; i < n_in; i++)
                    if (const uint32_t op = p.prog[at + 6 + 2 * (size_t)n_in + 2 * (size_t)n_out + 2 * i]; op != 0xFFFFFFFFu) times[op]++;
            }
        }
        const uint32_t end = p.truncated_at == 0xFFFFFFFFu ? (uint32_t)c.opcodes.size() : p.truncated_at;
        for (uint32_t oi = 0; oi < c.opcodes.size(); oi++) {
            const bool is_range = p.prog[p.prog_offset[oi]] == PK_RANGE;
            if (times[oi] != (is_range && oi < end ? 1 : 0)) violated(P, "RANGE opcode " + std::to_string(oi) + " is checked " + std::to_string(times[oi]) + " times");
        }
    }
    void check_dependencies() const {
        const char *P = "order_and_dependencies";
        for (uint32_t L = 0; L <= max_level; L++) {
            if (p.level_needs_inverse[L] && p.level_needs_inverse[L] >= L) violated(P, "a level waits for an inversion batch that is not earlier");
            for (int q = 0; q < N_HEAVY_LANES; q++) {
                if (p.level_needs_heavy[q][L] && p.level_needs_heavy[q][L] >= L) violated(P, "a main level waits for a heavy level that is not earlier");
                if (p.inv_needs_heavy[q][L] && p.inv_needs_heavy[q][L] >= L) violated(P, "an inversion batch waits for a heavy level that is not earlier");
                if (p.lane_needs_main[q][L] && p.lane_needs_main[q][L] >= L) violated(P, "a lane waits for a main level that is not earlier");
                for (int q2 = 0; q2 < N_HEAVY_LANES; q2++)
                    if (p.lane_needs_lane[q][q2][L] && p.lane_needs_lane[q][q2][L] >= L) violated(P, "a lane waits for a lane level that is not earlier");
            }
        }
        for (size_t i = 1; i < gates.size(); i++)
            if (gates[i].level < gates[i - 1].level) violated(P, "gates are not in level order");
        for (size_t i = 1; i < records.size(); i++)
            if (records[i].level < records[i - 1].level) violated(P, "records are not in level order");
    }
    void check_rows() const {
        const char *P = "assign_rows";
        if (!opts.reuse_slots) { if (!p.slot_of.empty()) violated(P, "rows without the option"); return; }
        std::vector<uint8_t> used(p.n_slots, 0);
        for (uint32_t w = 0; w < nw; w++) {
            if ((p.producer[w] != 0xFFFFFFFFu) != (p.slot_of[w] != 0xFFFFFFFFu)) violated(P, "witness " + std::to_string(w) + ": assigned witnesses and rows differ");
            if (p.slot_of[w] != 0xFFFFFFFFu) { if (p.slot_of[w] >= p.n_slots) violated(P, "row out of range"); used[p.slot_of[w]] = 1; }
        }
        for (uint32_t r = 0; r < p.n_slots; r++)
            if (!used[r]) violated(P, "row " + std::to_string(r) + " belongs to nobody");
        std::vector<uint8_t> init_row(p.n_slots, 0);
        for (uint32_t i = 0; i < n_initial; i++) init_row[p.slot_of[initial_ids[i]]]++;
        for (uint32_t w = 0; w < nw; w++)
            if (p.slot_of[w] != 0xFFFFFFFFu && init_row[p.slot_of[w]] && p.producer[w] != 0xFFFFFFFEu) violated(P, "witness " + std::to_string(w) + " recycles the row of an initial witness");
        for (uint32_t w : opts.keep)
            if (w < nw && p.slot_of[w] != 0xFFFFFFFFu)
                for (uint32_t v = 0; v < nw; v++)
                    if (v != w && p.slot_of[v] == p.slot_of[w] && wdef[v] >= wdef[w]) violated(P, "the row of kept witness " + std::to_string(w) + " is recycled");  // (it may itself have taken a dead witness's row)
    }
    void check_layout() {
        const char *P = "lay_out";
        if (inject(3) && !p.gate_offset.empty()) p.gate_offset.pop_back();
        size_t hosts = 0, listed = 0, chained = 0;
        for (const PendingGate &g : gates) hosts += !g.fused;
        for (const PendingRecord &r : records) chained += r.chained;
        for (int k = 0; k < N_CLS; k++) {
            listed += p.cls_offset[k].size();
            if (p.cls_level_start[k].size() != (size_t)max_level + 1 || p.cls_level_start[k].back() != p.cls_offset[k].size() || p.cls_scratch[k].size() != p.cls_offset[k].size()) violated(P, "a class's level table does not close its list");
            for (size_t L = 1; L < p.cls_level_start[k].size(); L++)
                if (p.cls_level_start[k][L] < p.cls_level_start[k][L - 1]) violated(P, "a class's level table goes backwards");
        }
        if (p.gate_offset.size() != hosts) violated(P, "wave programs listed " + std::to_string(p.gate_offset.size()) + ", hosts " + std::to_string(hosts));
        if (p.dyn_offset.size() != inverses.size()) violated(P, "inversion jobs and inversions differ");
        if (listed + chained != records.size()) violated(P, "records listed + chained != records");
        if (p.level_start.size() != (size_t)max_level + 1 || p.level_start.back() != p.gate_offset.size() || p.dyn_level_start.back() != p.dyn_offset.size()) violated(P, "the level tables do not close the lists");
        for (uint32_t off : p.gate_offset)
            if (off + GATE_HDR_WORDS + 1 > p.gate_stream.size()) violated(P, "a wave program starts past the stream");
    }
};

}  // namespace

Plan build_plan(const Circuit &c, const uint32_t *initial_ids, uint32_t n_initial, const PlanOpts &opts) {
    Planner b(c, initial_ids, n_initial, opts);
    b.p.tune = b.tune;
    b.p.n_opcodes = (uint32_t)c.opcodes.size();
    const bool v = b.tune.plan_validate != 0;           // every pass's promise checked on the way (tests/test_plan_host.py)
    if (!b.init()) return b.finish();                   // the dense witness table, the initial set, constants, memory blocks
    if (!b.emit_in_order_program()) return b.finish();  // `prog`: one record per opcode (exact path; non-Arithmetic records of the level path)
    if (v) b.check_in_order_program();
    b.pin_witnesses();                                  // which witnesses must stay plain canonical values
    b.replay();                                         // assigned-set replay, levels, folded gates, records, scales, bounds
    if (v) b.check_replay();
    b.fuse_gate_pairs();                                // wave programs: gates behind their producer
    if (v) b.check_pairs();
    b.assign_inverse_slots();                           // rows of the inverse table, reused
    if (v) b.check_inverse_slots();
    b.fold_digest_leaves();                             // PlanOpts::fold_digest
    if (v) b.check_digest_leaves();
    b.fuse_range_checks();                              // byte RANGE into the hash that reads the byte
    b.chain_hashes();                                   // a hash of a hash runs in its predecessor's workgroup
    b.merge_range_records();                            // eight RANGE checks to a record
    if (v) b.check_range_accounting();
    b.order_and_dependencies();                         // level order; what each stream waits for
    if (v) b.check_dependencies();
    if (!b.assign_rows()) return b.finish();            // PlanOpts::reuse_slots
    if (v) b.check_rows();
    b.lay_out();                                        // the streams and the level-major lists
    if (v) b.check_layout();
    order_level_records(b.p);                           // placement inside a level
    return b.finish();
}

}  // namespace acvm
