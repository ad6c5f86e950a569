// schedule_check.cpp -- host-only hazard checker of the level schedule (schedule.hpp). The reference executes the opcodes of an instance
// strictly in order (acvm/src/pwg/mod.rs:236-303); the level schedule runs them on up to six streams with deliberately PARTIAL waits, rows
// of the witness table recycled under slot reuse, rows of the inverse table reused, and rows whose representation depends on the consumer
// (relaxed rows, gate_eval.hpp). A missing edge would be a timing-dependent wrong witness. This file proves, for one plan and one schedule,
// without a device:
//   (1) ORDER: for every row of the witness table, of the inverse table, every memory cell, byte plane, digest leaf and class scratch
//       buffer, any two accesses of which one is a write are ordered by happens-before (stream order + event edges) -- or belong to the
//       same wave program / record of one launch; every launch is ordered behind the reset of the event words and in front of the
//       end of the solve (where the host reads the flags and the next tile's import overwrites the initial rows);
//   (2) VALUE: a read sees the write of exactly the witness the opcode's expression names: the latest write before it (unique by (1)) is
//       tagged with the witness it stored, and the reader's intended witness is derived from the ORIGINAL opcode (the in-order program's
//       record, one per opcode), not from the folded gate record;
//   (3) REPRESENTATION: every reader outside the gate kernels sees a canonical row of an unscaled witness (the digest leaves: or multiplies
//       by exactly that witness's 1 / scale); the inversion kernel's denominators are canonical (its zero test is on the stored row);
//   (4) BOUNDS: re-walking gate_eval's order of summation with the bound of each operand's actual writer, no intermediate passes the
//       domain of fr29_weak (2^261), no stored row passes 2^256 (GATE_K_ROW_MAX), and every subtracted operand fits the 2^k p it is
//       subtracted from.
// What is read and written comes from the RECORD WORDS (gate_stream, prog through cls_offset, the inversion jobs) and an independent
// in-order replay of the assigned set -- not from the planner's dependency tables (level_needs_*, lane_needs_*, slot liveness), which
// are exactly what is being checked.
#include "gate_record.hpp"
#include "schedule.hpp"
#include <array>
#include <cstdio>
#include <string>

namespace acvm {
namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
enum ResKind : int { RK_W = 0, RK_INV, RK_MEM, RK_BYTE_PLANE, RK_LEAF, RK_SCRATCH, N_RK };
const char *RES_NAME[N_RK] = {"witness row", "inverse row", "memory cell", "byte plane", "digest leaf", "class scratch"};
constexpr uint32_t K_WEAK_DOMAIN = 169u * 256u;  // fr29_weak takes any value below 2^261 = 169.3 p (fr_device.hpp)

struct Res {
    int32_t w_launch = -1;      // the launch of the latest write (-1: never written)
    uint32_t w_prog = 0;        // ... and its program (wave program / record) inside that launch
    uint32_t tag = NONE;        // what that write stored: the witness (rows), the gate's opcode (inverse rows)
    uint32_t kb = GATE_K_CANON; // bound of the stored representative in units of p / 256
    uint8_t canon = 1;          // the stored representative is the canonical one
    uint8_t r_multi = 0;        // per stream: readers of more than one program in r_launch
    int32_t r_launch[N_SCHED_STREAMS] = {-1, -1, -1, -1, -1, -1};  // per stream: the latest launch that read the resource since the last write
    uint32_t r_prog[N_SCHED_STREAMS] = {0, 0, 0, 0, 0, 0};
};
struct Launch {
    uint8_t stream, op;
    uint32_t level, seq, step;
    std::array<uint32_t, N_SCHED_STREAMS> clock;  // per stream: the latest launch of that stream (by seq) that happens before this one
};

struct Checker {
    const Plan &p;
    const LaunchLayout &lay;
    const LevelSchedule &sch;
    ScheduleReport rep;
    uint64_t base[N_RK + 1];
    std::vector<Res> res;
    std::vector<Launch> launches;
    bool reuse;
    uint32_t n_rows;
    // independent in-order replay of the assigned set over the in-order program (one record per opcode)
    std::vector<uint32_t> producer2;               // witness -> opcode that first assigns it, NONE - 1 = initial, NONE = never
    std::vector<std::vector<uint32_t>> out_of;     // opcode -> witnesses it assigns (in record order)
    std::vector<uint32_t> partner_of;              // Arithmetic opcode -> the known multiplicand of its unknown, NONE if the unknown is in a linear term
    std::vector<uint32_t> inv_slot_of;             // opcode -> inverse slot its inversion job fills (as enqueued so far)
    std::vector<uint8_t> wrote;                    // witness -> a launch of the schedule stored it
    int cur = -1;                                  // launch being walked
    uint32_t cur_prog = 0;

    Checker(const Plan &pl, const LaunchLayout &l, const LevelSchedule &s) : p(pl), lay(l), sch(s) {
        reuse = !p.slot_of.empty();
        n_rows = reuse ? p.n_slots : p.n_witnesses;
        const uint64_t sizes[N_RK] = {n_rows, p.n_inverse_slots, p.mem_cells, p.n_byte_planes, p.n_digest_segments, N_CLS};
        base[0] = 0;
        for (int k = 0; k < N_RK; k++) base[k + 1] = base[k] + sizes[k];
        res.resize(base[N_RK]);
        wrote.assign(p.n_witnesses, 0);
    }
    void finding(const std::string &what) {
        rep.ok = false;
        if (rep.n_findings++ < 32) rep.text += what + "\n";
    }
    std::string launch_name(int id) const {
        if (id < 0) return "(nothing)";
        const Launch &l = launches[id];
        static const char *sn[N_SCHED_STREAMS] = {"main", "inversions", "lane0", "lane1", "lane2", "digest"};
        char buf[160];
        if (l.op == 0xFE) snprintf(buf, sizeof buf, "launch #%d [import of the initial witnesses, stream main]", id);
        else if (l.op == 0xFF) snprintf(buf, sizeof buf, "launch #%d [end of the solve: flag count, export, next import; stream main]", id);
        else snprintf(buf, sizeof buf, "launch #%d [%s, level %u, stream %s, step %u]", id, sched_op_name(l.op), l.level + 1, sn[l.stream], l.step);
        return buf;
    }
    bool hb(int a, int b) const {  // launch a happens before launch b
        if (a < 0) return true;
        return launches[b].clock[launches[a].stream] >= launches[a].seq && a != b;
    }
    uint32_t row_of(uint32_t w) const { return reuse ? (w < p.slot_of.size() ? p.slot_of[w] : NONE) : w; }
    std::string res_name(int kind, uint64_t idx) const {
        char buf[96];
        snprintf(buf, sizeof buf, "%s %llu", RES_NAME[kind], (unsigned long long)idx);
        return buf;
    }

    // ---- (1) order, (2) value
    Res *touch(int kind, uint64_t idx, const char *who) {
        if (idx >= base[kind + 1] - base[kind]) {
            finding(std::string(who) + ": " + res_name(kind, idx) + " is outside its table, in " + launch_name(cur));
            return nullptr;
        }
        rep.n_accesses++;
        return &res[base[kind] + idx];
    }
    // want_tag: the witness (opcode for inverse rows) whose value the reader means to see, NONE = any
    Res *read(int kind, uint64_t idx, uint32_t want_tag, const char *who) {
        Res *r = touch(kind, idx, who);
        if (!r) return nullptr;
        const uint8_t s = launches[cur].stream;
        if (r->w_launch < 0) finding(std::string(who) + " reads " + res_name(kind, idx) + ", which nothing has written, in " + launch_name(cur));
        else if (r->w_launch == cur) {
            if (r->w_prog != cur_prog) finding(std::string(who) + " reads " + res_name(kind, idx) + " written by another record of the SAME launch (no order inside a launch): " + launch_name(cur));
        } else if (!hb(r->w_launch, cur))
            finding("RAW: " + std::string(who) + " reads " + res_name(kind, idx) + " in " + launch_name(cur) + " but its writer " + launch_name(r->w_launch) + " is not ordered before it");
        if (want_tag != NONE && r->w_launch >= 0 && r->tag != want_tag) {
            char buf[128];
            snprintf(buf, sizeof buf, " (holds %u, the reader wants %u)", r->tag, want_tag);
            finding("VALUE: " + std::string(who) + " reads " + res_name(kind, idx) + " in " + launch_name(cur) + " but the latest write, " + launch_name(r->w_launch) + ", stored something else" + buf);
        }
        if (r->r_launch[s] == cur) { if (r->r_prog[s] != cur_prog) r->r_multi |= (uint8_t)(1u << s); }
        else { r->r_launch[s] = cur; r->r_prog[s] = cur_prog; r->r_multi &= (uint8_t)~(1u << s); }
        return r;
    }
    Res *write(int kind, uint64_t idx, uint32_t tag, const char *who) {
        Res *r = touch(kind, idx, who);
        if (!r) return nullptr;
        if (r->w_launch == cur) {
            if (r->w_prog != cur_prog) finding("WAW: two records of " + launch_name(cur) + " write " + res_name(kind, idx));
        } else if (r->w_launch >= 0 && !hb(r->w_launch, cur))
            finding("WAW: " + std::string(who) + " writes " + res_name(kind, idx) + " in " + launch_name(cur) + " but the earlier writer " + launch_name(r->w_launch) + " is not ordered before it");
        for (int s = 0; s < N_SCHED_STREAMS; s++) {
            const int rl = r->r_launch[s];
            if (rl < 0) continue;
            if (rl == cur) {
                if (r->r_prog[s] != cur_prog || (r->r_multi >> s & 1)) finding("WAR: " + launch_name(cur) + " has one record reading and another writing " + res_name(kind, idx));
            } else if (!hb(rl, cur))
                finding("WAR: " + std::string(who) + " overwrites " + res_name(kind, idx) + " in " + launch_name(cur) + " but its reader " + launch_name(rl) + " is not ordered before it");
            r->r_launch[s] = -1;
        }
        r->r_multi = 0;
        r->w_launch = cur;
        r->w_prog = cur_prog;
        r->tag = tag;
        if (kind == RK_W && tag < wrote.size() && launches[cur].op < 0xFE) wrote[tag] = 1;
        r->kb = GATE_K_CANON;
        r->canon = 1;
        return r;
    }

    // ---- expressions of the in-order program: [n_mul, n_lin, qc, (coef, l, r) x n_mul, (coef, -1/coef, w) x n_lin]
    size_t expr_len(size_t at) const { return 3 + 3 * (size_t)p.prog[at] + 3 * (size_t)p.prog[at + 1]; }
    template <class F>
    void expr_witnesses(size_t at, F fn) const {  // every witness the device loads for this expression (zero-coefficient terms are skipped: ops_common.hpp expr_value)
        const uint32_t n_mul = p.prog[at], n_lin = p.prog[at + 1];
        size_t t = at + 3;
        for (uint32_t i = 0; i < n_mul; i++, t += 3)
            if (p.prog[t] != COEF_ZERO) { fn(p.prog[t + 1]); fn(p.prog[t + 2]); }
        for (uint32_t i = 0; i < n_lin; i++, t += 3)
            if (p.prog[t] != COEF_ZERO) fn(p.prog[t + 2]);
    }

    // A record of the in-order program / of the level lists as accesses. rd(w): the record loads witness w; out(w, flag): insert_value --
    // flag 1 compares (a read), flag 0 stores; mem_r / mem_w (first cell, n); plane(w): a byte-message hash input (row or byte plane);
    // leaf(row); digest inputs go through dig(w, unscale row). Returns false on a record kind it does not know.
    struct Sink {
        virtual void rd(uint32_t w) = 0;
        virtual void out(uint32_t w, uint32_t flag) = 0;
        virtual void mem_r(uint32_t first, uint32_t n) = 0;
        virtual void mem_w(uint32_t first, uint32_t n) = 0;
        virtual void byte_in(uint32_t w) = 0;
        virtual void dig(uint32_t w, uint32_t unscale_row) = 0;
        virtual void leaf(uint32_t row) = 0;
        virtual ~Sink() {}
    };
    bool walk_record(size_t at, Sink &k, bool level_list) const {
        const std::vector<uint32_t> &g = p.prog;
        auto E = [&](size_t e) { expr_witnesses(e, [&](uint32_t w) { k.rd(w); }); return e + expr_len(e); };
        switch (g[at]) {
        case PK_ARITH: E(at + 2); return true;  // (exact path only: the level schedule runs the folded gate stream)
        case PK_RANGE: k.rd(g[at + 2]); return true;
        case PK_RANGE_MULTI:
            for (uint32_t i = 0; i < g[at + 2]; i++) k.rd(g[at + 3 + 3 * i + 1]);
            return true;
        case PK_LOGIC: k.rd(g[at + 3]); k.rd(g[at + 4]); k.out(g[at + 7], g[at + 8]); return true;
        case PK_HASH: {
            for (size_t rec = at;;) {  // the record, then the members of the chain it heads (they run in its workgroup)
                const uint32_t fw = g[rec + 2], n_in = g[rec + 3], n_out = g[rec + 4];
                const bool coop = (fw & PLAN_HASH_COOP_FLAG) != 0;
                const uint32_t *src = nullptr;  // a chained member: per input, the byte of the predecessor's digest it is (read from LDS), or NONE
                if (rec != at) src = &g[link_src];
                for (uint32_t i = 0; i < n_in; i++) {
                    if (src && src[i] != NONE) continue;
                    if (coop) k.byte_in(g[rec + 6 + 2 * i]);
                    else k.rd(g[rec + 6 + 2 * i]);
                }
                if (g[rec + 5] != NONE) k.rd(g[rec + 5]);  // Keccak256VariableLength: the message size
                for (uint32_t i = 0; i < n_out; i++) k.out(g[rec + 6 + 2 * n_in + 2 * i], g[rec + 6 + 2 * n_in + 2 * i + 1]);
                if (!(fw & PLAN_HASH_CHAIN_FLAG) || !level_list) return true;
                const size_t link = g[rec + 6 + 2 * (size_t)n_in + 2 * (size_t)n_out + ((fw & PLAN_HASH_RANGE_FLAG) ? 2 * (size_t)n_in : 0)];
                rec = g[link];
                link_src = link + 1;
            }
        }
        case PK_PEDERSEN: {
            const uint32_t n_in = g[at + 3];
            for (uint32_t i = 0; i < n_in; i++) k.rd(g[at + 8 + i]);
            k.out(g[at + 4], g[at + 5]); k.out(g[at + 6], g[at + 7]);
            return true;
        }
        case PK_FIXED_BASE: k.rd(g[at + 2]); k.rd(g[at + 3]); k.out(g[at + 4], g[at + 5]); k.out(g[at + 6], g[at + 7]); return true;
        case PK_SCHNORR: {
            k.rd(g[at + 2]); k.rd(g[at + 3]);
            for (uint32_t i = 0; i < g[at + 4] + g[at + 5]; i++) k.rd(g[at + 8 + i]);
            k.out(g[at + 6], g[at + 7]);
            return true;
        }
        case PK_ECDSA: {
            const uint32_t n = g[at + 3] + g[at + 4] + g[at + 5] + g[at + 6];
            for (uint32_t i = 0; i < n; i++) k.rd(g[at + 9 + i]);
            k.out(g[at + 7], g[at + 8]);
            return true;
        }
        case PK_ZERO_OUT: {
            const uint32_t n_in = g[at + 2], n_out = g[at + 3];
            for (uint32_t i = 0; i < n_in; i++) k.rd(g[at + 4 + i]);
            for (uint32_t i = 0; i < n_out; i++) k.out(g[at + 4 + n_in + 2 * i], g[at + 4 + n_in + 2 * i + 1]);
            return true;
        }
        case PK_QUOTIENT: {
            size_t e = at + 7;
            e = E(e); e = E(e);
            if (g[at + 6]) E(e);
            k.out(g[at + 2], g[at + 3]); k.out(g[at + 4], g[at + 5]);
            return true;
        }
        case PK_TO_LE_RADIX: {
            const uint32_t n_out = g[at + 3];
            E(at + 4 + 2 * (size_t)n_out);
            for (uint32_t i = 0; i < n_out; i++) k.out(g[at + 4 + 2 * i], g[at + 4 + 2 * i + 1]);
            return true;
        }
        case PK_PERM_SORT: {
            const uint32_t n = g[at + 2], tuple = g[at + 3], n_sort_by = g[at + 4], n_bits = g[at + 5];
            size_t e = at + 6 + n_sort_by + 2 * (size_t)n_bits;
            for (uint64_t i = 0; i < (uint64_t)n * tuple; i++) e = E(e);
            for (uint32_t i = 0; i < n_bits; i++) k.out(g[at + 6 + n_sort_by + 2 * i], g[at + 6 + n_sort_by + 2 * i + 1]);
            return true;
        }
        case PK_MEM_INIT:
            for (uint32_t i = 0; i < g[at + 3]; i++) k.rd(g[at + 4 + i]);
            k.mem_w(g[at + 2], g[at + 3]);
            return true;
        case PK_MEM_OP: {
            // the index is a per-instance value: a read may touch any readable cell of the block, a write any cell
            size_t e = at + 9;
            e = E(e);              // operation
            e = E(e);              // index
            const size_t e_val = e;
            e += expr_len(e);
            if (g[at + 5]) E(e);   // predicate
            if (g[at + 6] == 1) { k.mem_r(g[at + 2], g[at + 4]); k.out(g[at + 7], g[at + 8]); }
            else { E(e_val); k.mem_w(g[at + 2], g[at + 3]); }
            return true;
        }
        case PK_BRILLIG: {
            const uint32_t has_pred = g[at + 2], n_inputs = g[at + 3], n_outputs = g[at + 4];
            size_t e = at + 12;
            if (has_pred) e = E(e);
            for (uint32_t i = 0; i < n_inputs; i++) {
                const uint32_t n = g[e + 1];
                e += 2;
                for (uint32_t j = 0; j < n; j++) e = E(e);
            }
            for (uint32_t i = 0; i < n_outputs; i++) {
                const uint32_t n = g[e + 1];
                e += 2;
                for (uint32_t j = 0; j < n; j++, e += 2) k.out(g[e], g[e + 1]);
            }
            return true;
        }
        case PK_BRILLIG_SL: {
            const uint32_t has_pred = g[at + 2], n_in = g[at + 3], n_out = g[at + 4];
            size_t e = at + 6;
            if (has_pred) e = E(e);
            for (uint32_t i = 0; i < n_in; i++) e = E(e);
            for (uint32_t i = 0; i < n_out; i++, e += 2) k.out(g[e], g[e + 1]);
            return true;
        }
        case PK_DIGEST_LEAF:
            for (uint32_t i = 0; i < g[at + 2]; i++) k.dig(g[at + 3 + 2 * i], g[at + 3 + 2 * i + 1]);
            k.leaf(g[at + 1]);
            return true;
        default: return false;
        }
    }
    mutable size_t link_src = 0;

    // ---- the independent replay: which opcode assigns which witness (insert_value, pwg/mod.rs:338-357), over the in-order program
    void replay() {
        const uint32_t nw = p.n_witnesses;
        producer2.assign(nw, NONE);
        out_of.assign(p.n_opcodes, {});
        partner_of.assign(p.n_opcodes, NONE);
        inv_slot_of.assign(p.n_opcodes, NONE);
        std::vector<uint8_t> known(nw, 0);
        for (uint32_t w : p.initial_ids)
            if (w < nw) { known[w] = 1; producer2[w] = NONE - 1; }
        struct Outs : Sink {
            Checker &c; std::vector<uint8_t> &known; uint32_t oi;
            Outs(Checker &cc, std::vector<uint8_t> &kn) : c(cc), known(kn), oi(0) {}
            void rd(uint32_t) override {}
            void out(uint32_t w, uint32_t flag) override {
                if (w >= known.size()) { c.finding("replay: output witness beyond the table at opcode " + std::to_string(oi)); return; }
                if ((flag != 0) != (known[w] != 0)) c.finding("replay: opcode " + std::to_string(oi) + " output witness " + std::to_string(w) + ": the record's 'already assigned' flag disagrees with the in-order replay");
                if (!known[w]) { known[w] = 1; c.producer2[w] = oi; c.out_of[oi].push_back(w); }
            }
            void mem_r(uint32_t, uint32_t) override {}
            void mem_w(uint32_t, uint32_t) override {}
            void byte_in(uint32_t) override {}
            void dig(uint32_t, uint32_t) override {}
            void leaf(uint32_t) override {}
        } outs(*this, known);
        const uint32_t end = p.truncated_at == NONE ? p.n_opcodes : p.truncated_at;
        for (uint32_t oi = 0; oi < end; oi++) {
            const size_t at = p.prog_offset[oi];
            outs.oi = oi;
            if (p.prog[at] == PK_ARITH) {  // ArithmeticSolver::solve for the generic instance: exactly one unknown (or none: a constraint)
                const size_t e = at + 2;
                const uint32_t n_mul = p.prog[e], n_lin = p.prog[e + 1];
                uint32_t unk = NONE, partner = NONE, n_unk = 0;
                size_t t = e + 3;
                for (uint32_t i = 0; i < n_mul; i++, t += 3) {
                    if (p.prog[t] == COEF_ZERO) continue;
                    const uint32_t l = p.prog[t + 1], r = p.prog[t + 2];
                    const bool kl = known[l], kr = known[r];
                    if (kl && kr) continue;
                    if (!kl && !kr) { n_unk += 2; continue; }
                    n_unk++; unk = kl ? r : l; partner = kl ? l : r;
                }
                for (uint32_t i = 0; i < n_lin; i++, t += 3) {
                    if (p.prog[t] == COEF_ZERO) continue;
                    if (!known[p.prog[t + 2]]) { n_unk++; unk = p.prog[t + 2]; partner = NONE; }
                }
                if (n_unk > 1) { finding("replay: Arithmetic opcode " + std::to_string(oi) + " has more than one unknown but the plan is not truncated there"); continue; }
                if (n_unk == 1) { known[unk] = 1; producer2[unk] = oi; out_of[oi].push_back(unk); partner_of[oi] = partner; }
            } else if (!walk_record(at, outs, false)) finding("replay: unknown record kind at opcode " + std::to_string(oi));
        }
        for (uint32_t w = 0; w < nw; w++)
            if (producer2[w] != p.producer[w])
                finding("replay: witness " + std::to_string(w) + " is first assigned by opcode " + std::to_string(producer2[w]) + " in order, the plan says " + std::to_string(p.producer[w]));
    }

    // ---- a record of the level lists on the device
    struct DeviceSink : Sink {
        Checker &c; const char *who; bool is_digest = false;
        DeviceSink(Checker &cc, const char *w) : c(cc), who(w) {}
        Res *row_read(uint32_t w, bool need_canon) {
            if (w >= c.p.n_witnesses) { c.finding(std::string(who) + ": witness " + std::to_string(w) + " beyond the table"); return nullptr; }
            const uint32_t row = c.row_of(w);
            if (row == NONE) { c.finding(std::string(who) + " reads witness " + std::to_string(w) + ", which has no row of the table, in " + c.launch_name(c.cur)); return nullptr; }
            Res *r = c.read(RK_W, row, w, who);
            if (r && need_canon) {
                if (!r->canon) c.finding("REPRESENTATION: " + std::string(who) + " reads witness " + std::to_string(w) + " (row " + std::to_string(row) + ") as it is in " + c.launch_name(c.cur) + " but its writer " + c.launch_name(r->w_launch) + " stored a relaxed row");
                if (c.p.unscale_index[w] != NONE) c.finding("REPRESENTATION: " + std::string(who) + " reads witness " + std::to_string(w) + " as it is in " + c.launch_name(c.cur) + " but the witness is stored scaled");
            }
            return r;
        }
        void rd(uint32_t w) override { row_read(w, true); }
        void out(uint32_t w, uint32_t flag) override {
            if (flag) { row_read(w, true); return; }  // compared, never overwritten
            const uint32_t row = c.row_of(w);
            if (w >= c.p.n_witnesses || row == NONE) { c.finding(std::string(who) + " writes witness " + std::to_string(w) + ", which has no row, in " + c.launch_name(c.cur)); return; }
            c.write(RK_W, row, w, who);
        }
        void mem_r(uint32_t first, uint32_t n) override { for (uint32_t i = 0; i < n; i++) c.read(RK_MEM, first + i, NONE, who); }
        void mem_w(uint32_t first, uint32_t n) override { for (uint32_t i = 0; i < n; i++) c.write(RK_MEM, first + i, NONE, who); }
        void byte_in(uint32_t w) override {
            const uint32_t pl = c.p.byte_plane_of.empty() || w >= c.p.byte_plane_of.size() ? NONE : c.p.byte_plane_of[w];
            if (pl == NONE) { row_read(w, true); return; }
            c.read(RK_BYTE_PLANE, pl, w, who);
        }
        void dig(uint32_t w, uint32_t unscale_row) override {
            Res *r = row_read(w, false);
            if (w >= c.p.n_witnesses) return;
            if (unscale_row != c.p.unscale_index[w]) c.finding("REPRESENTATION: a digest leaf multiplies witness " + std::to_string(w) + " by the wrong 1 / scale");
            if (r && unscale_row == NONE && !r->canon) c.finding("REPRESENTATION: a digest leaf reads witness " + std::to_string(w) + " as it is but " + c.launch_name(r->w_launch) + " stored a relaxed row");
        }
        void leaf(uint32_t row) override { c.write(RK_LEAF, row, NONE, who); }
    };

    // ---- (4) a gate wave program: accesses + gate_eval's bounds walk (gate_eval.hpp), operand bounds from the rows' actual writers
    void walk_wave_program(uint32_t gate_index) {
        const std::vector<uint32_t> &gs = p.gate_stream;
        size_t pos = p.gate_offset[gate_index];
        uint32_t local_kb = GATE_K_CANON, local_witness = NONE;
        bool host = true;
        for (;;) {
            if (pos + 6 > gs.size()) { finding("gate record runs past the stream"); return; }
            const uint32_t w0 = gs[pos], kind = w0 & 0xff, opcode = gs[pos + 1], w5 = gs[pos + 5];
            const uint32_t np_mac = (w0 >> 8) & 0xff, nl_mac = (w0 >> 16) & 0xff, n_mac = np_mac + nl_mac;
            const uint32_t np_pos = w5 & 0xff, np_neg = (w5 >> 8) & 0xff, nl_pos = (w5 >> 16) & 0xff, nl_neg = w5 >> 24;
            const uint32_t sub_k = (w0 >> GATE_SUBK_SHIFT) & 3u;
            rep.n_records++;
            char who[64];
            snprintf(who, sizeof who, "gate of opcode %u", opcode);
            if (opcode >= p.n_opcodes || p.prog[p.prog_offset[opcode]] != PK_ARITH) { finding(std::string(who) + " is not an Arithmetic opcode"); return; }
            // the witnesses the ORIGINAL expression names, less the unknown and its multiplicand (that one is read by the inversion job)
            const uint32_t out_w = out_of[opcode].empty() ? NONE : out_of[opcode][0];
            std::vector<uint32_t> cand;
            {
                const size_t e = p.prog_offset[opcode] + 2;
                const uint32_t n_mul = p.prog[e], n_lin = p.prog[e + 1];
                size_t t = e + 3;
                for (uint32_t i = 0; i < n_mul; i++, t += 3) {
                    if (p.prog[t] == COEF_ZERO) continue;
                    const uint32_t l = p.prog[t + 1], r = p.prog[t + 2];
                    if (l == out_w || r == out_w) continue;
                    cand.push_back(l); cand.push_back(r);
                }
                for (uint32_t i = 0; i < n_lin; i++, t += 3)
                    if (p.prog[t] != COEF_ZERO && p.prog[t + 2] != out_w) cand.push_back(p.prog[t + 2]);
                std::sort(cand.begin(), cand.end());
                cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
            }
            std::vector<uint8_t> matched(cand.size(), 0);
            bool uses_local = false;
            // an operand row -> the bound of what it holds (checked against the expression's witnesses on the way)
            auto operand = [&](uint32_t row) -> uint32_t {
                if (row == GATE_LOCAL) {
                    uses_local = true;
                    if (host) finding(std::string(who) + " reads the wave's forwarded value but heads its wave program");
                    return local_kb;
                }
                uint32_t want = NONE;
                for (size_t i = 0; i < cand.size(); i++)
                    if (row_of(cand[i]) == row) {
                        if (want != NONE && want != cand[i]) finding("VALUE: " + std::string(who) + ": witnesses " + std::to_string(want) + " and " + std::to_string(cand[i]) + " of one expression share row " + std::to_string(row));
                        want = cand[i];
                        matched[i] = 1;
                    }
                if (want == NONE) { finding("VALUE: " + std::string(who) + " reads row " + std::to_string(row) + ", which holds none of its expression's witnesses, in " + launch_name(cur)); }
                Res *r = read(RK_W, row, want, who);
                return r ? r->kb : GATE_K_CANON;
            };
            uint32_t hk = 0, hw = 0;
            auto weak_check = [&](const char *where) {
                if (hk > K_WEAK_DOMAIN) finding("BOUNDS: " + std::string(who) + ": the running sum reaches " + std::to_string(hk) + " / 256 p " + where + ", past the domain of fr29_weak");
            };
            auto room = [&](uint32_t weight) {
                if (hw + weight > GATE_H_MAX) { weak_check("before a side-sum reduction"); hk = GATE_K_WEAK; hw = GATE_H_AFTER_WEAK; }
                hw += weight;
            };
            if (gs[pos + 3] != GATE_COEF_ZERO) { hk = GATE_K_CANON; hw = 16; }
            const size_t t0 = pos + 6, tp = t0 + 10 * (size_t)np_mac + 9 * (size_t)nl_mac;
            size_t t = tp + 2 * (size_t)np_pos;
            const size_t rec_end = t + 2 * (size_t)np_neg + nl_pos + nl_neg;
            if (rec_end > gs.size()) { finding("gate record runs past the stream"); return; }
            for (uint32_t i = 0; i < np_neg; i++, t += 2) { operand(gs[t]); operand(gs[t + 1]); room(33); hk += 2 * GATE_K_CANON; }
            for (uint32_t i = 0; i < nl_pos; i++, t += 1) { const uint32_t kb = operand(gs[t]); room(16); hk += kb; }
            for (uint32_t i = 0; i < nl_neg; i++, t += 1) {
                const uint32_t kb = operand(gs[t]);
                const uint32_t k = sub_k < 1 ? 1 : sub_k;
                if (kb > (GATE_K_CANON << k)) finding("BOUNDS: " + std::string(who) + " subtracts a row bounded by " + std::to_string(kb) + " / 256 p from 2^" + std::to_string(k) + " p");
                room(33);
                hk += GATE_K_CANON << k;
            }
            auto mac_k = [&](uint32_t im) {  // the im-th multiplied term: coefficient (canonical) x (product | witness)
                if (im < np_mac) {
                    const size_t c = t0 + 10 * (size_t)im;
                    const uint32_t ka = operand(gs[c + 8]), kb = operand(gs[c + 9]);
                    return gate_k_product(GATE_K_CANON + gate_k_product(ka, kb), GATE_K_CANON);
                }
                const size_t c = t0 + 10 * (size_t)np_mac + 9 * (size_t)(im - np_mac);
                return gate_k_product(operand(gs[c + 8]), GATE_K_CANON);
            };
            auto pp_k = [&](uint32_t ip) { const uint32_t ka = operand(gs[tp + 2 * (size_t)ip]), kb = operand(gs[tp + 2 * (size_t)ip + 1]); return gate_k_product(ka, kb); };
            uint32_t im = 0, ip = 0, n_red = 0;
            auto reduction = [&](uint32_t adds) {
                hk += GATE_K_CANON + adds;
                if (++n_red == GATE_REDUCTIONS_PER_WEAK) { weak_check("before the periodic reduction"); hk = GATE_K_WEAK; n_red = 0; }
            };
            for (; ip < np_pos && im < n_mac; ip++, im++) { const uint32_t a = pp_k(ip), b = mac_k(im); reduction(a + b); }
            for (; im < n_mac; im += 2) { const uint32_t a = mac_k(im), b = n_mac - im == 1 ? 0u : mac_k(im + 1); reduction(a + b); }
            for (; ip < np_pos; ip += 2) { const uint32_t a = pp_k(ip), b = np_pos - ip == 1 ? 0u : pp_k(ip + 1); reduction(a + b); }
            uint32_t acc_k = hk;
            if (kind == GATE_SOLVE_DYN) {
                if (w0 & GATE_PRESUM_WEAK) { weak_check("before the product with the inverse"); acc_k = GATE_K_WEAK; }
                else if (acc_k > 8 * GATE_K_CANON) finding("BOUNDS: " + std::string(who) + ": the sum (" + std::to_string(acc_k) + " / 256 p) meets the inverse without a reduction");
                read(RK_INV, gs[pos + 4], opcode, who);
                if (inv_slot_of[opcode] != gs[pos + 4]) finding("VALUE: " + std::string(who) + " reads inverse row " + std::to_string(gs[pos + 4]) + " but its inversion job fills row " + std::to_string(inv_slot_of[opcode]));
                acc_k = GATE_K_CANON + gate_k_product(acc_k, GATE_K_INVERSE);
            }
            // every witness of the expression was an operand -- but for the one the wave forwards in registers
            uint32_t forwarded = NONE, n_unmatched = 0;
            for (size_t i = 0; i < cand.size(); i++)
                if (!matched[i]) { forwarded = cand[i]; n_unmatched++; }
            if (uses_local) {
                if (n_unmatched != 1) finding("VALUE: " + std::string(who) + " takes a forwarded operand but " + std::to_string(n_unmatched) + " witnesses of its expression are not among its rows");
                else if (forwarded != local_witness) finding("VALUE: " + std::string(who) + " wants witness " + std::to_string(forwarded) + " forwarded, the wave holds witness " + std::to_string(local_witness));
            } else if (n_unmatched) finding("VALUE: " + std::string(who) + " never reads witness " + std::to_string(forwarded) + " of its expression");
            uint32_t out_kb = GATE_K_CANON;
            bool out_canon = true;
            if (kind == GATE_ASSERT) {
                weak_check("at the end of a constraint");
                if (out_w != NONE) finding("VALUE: " + std::string(who) + " is a constraint record but the opcode assigns witness " + std::to_string(out_w));
            } else {
                const uint32_t mode = (w0 >> GATE_OUT_SHIFT) & 3u;
                if (mode == GATE_OUT_ASIS) {
                    if (acc_k > GATE_K_ROW_MAX) finding("BOUNDS: " + std::string(who) + " stores its sum as it is but the sum may reach " + std::to_string(acc_k) + " / 256 p, past 2^256");
                    out_kb = acc_k; out_canon = false;
                } else if (mode == GATE_OUT_WEAK) { hk = acc_k; weak_check("before the final reduction"); out_kb = GATE_K_WEAK; out_canon = false; }
                else { hk = acc_k; weak_check("before the final reduction"); }
                if (out_w == NONE) finding("VALUE: " + std::string(who) + " stores a witness but the opcode assigns none in order");
                else {
                    if (row_of(out_w) != gs[pos + 2]) finding("VALUE: " + std::string(who) + " writes row " + std::to_string(gs[pos + 2]) + ", witness " + std::to_string(out_w) + " lives in row " + std::to_string(row_of(out_w)));
                    if (!out_canon && p.unscale_index[out_w] == NONE) finding("REPRESENTATION: " + std::string(who) + " stores a relaxed row for witness " + std::to_string(out_w) + ", which no reader unscales");
                    if (Res *r = write(RK_W, gs[pos + 2], out_w, who)) { r->kb = out_kb; r->canon = out_canon; }
                }
                if ((kind == GATE_SOLVE_DYN) != (partner_of[opcode] != NONE)) finding("VALUE: " + std::string(who) + ": the record's kind disagrees with where the unknown stands in the expression");
            }
            if (!(w0 & GATE_TAIL_FLAG)) break;
            if (host || (w0 & GATE_SETLOCAL_FLAG)) { local_kb = out_kb; local_witness = out_w; }
            host = false;
            pos = rec_end;
        }
    }

    void walk_inversion(uint32_t job) {
        const size_t at = p.dyn_offset[job];
        const uint32_t den_row = p.gate_stream[at], opcode = p.gate_stream[at + 1], slot = p.gate_stream[at + 2];
        rep.n_records++;
        char who[64];
        snprintf(who, sizeof who, "inversion job of opcode %u", opcode);
        if (opcode >= p.n_opcodes || partner_of[opcode] == NONE) { finding(std::string(who) + ": the opcode has no multiplicand to invert"); return; }
        const uint32_t partner = partner_of[opcode];
        if (row_of(partner) != den_row) finding("VALUE: " + std::string(who) + " reads row " + std::to_string(den_row) + ", its denominator (witness " + std::to_string(partner) + ") lives in row " + std::to_string(row_of(partner)));
        if (Res *r = read(RK_W, den_row, partner, who))
            if (!r->canon) finding("REPRESENTATION: " + std::string(who) + " tests a relaxed row for zero (witness " + std::to_string(partner) + ", written by " + launch_name(r->w_launch) + ")");
        if (Res *r = write(RK_INV, slot, opcode, who)) { r->kb = GATE_K_INVERSE; r->canon = 0; }
        inv_slot_of[opcode] = slot;
    }

    void walk_class_records(const SchedStep &st, int cls, uint32_t first, uint32_t count) {
        const bool scratch = st.op == SO_HASH || st.op == SO_GRUMPKIN || st.op == SO_BRILLIG || st.op == SO_PEDERSEN;
        uint64_t prev_end = 0;
        for (uint32_t r = first; r < first + count; r++) {
            if (r >= p.cls_offset[cls].size()) { finding("launch runs past its class's record list: " + launch_name(cur)); return; }
            const size_t at = p.cls_offset[cls][r];
            cur_prog = 0x40000000u + r;
            rep.n_records++;
            char who[64];
            snprintf(who, sizeof who, "record at prog[%zu] (kind %u, opcode %u)", at, p.prog[at], p.prog[at] == PK_DIGEST_LEAF || p.prog[at] == PK_RANGE_MULTI ? NONE : p.prog[at + 1]);
            DeviceSink sink(*this, who);
            if (!walk_record(at, sink, true)) finding(std::string(who) + ": unknown record kind");
            if (scratch) {  // the records of a launch own disjoint pieces of the class's scratch buffer
                const uint64_t off = lay.scratch_off[cls][2 * (size_t)r], words = lay.scratch_off[cls][2 * (size_t)r + 1];
                if (words && off < prev_end && r > first) finding("SCRATCH: two records of " + launch_name(cur) + " share scratch words");
                if (off + words > lay.scratch_words[cls]) finding("SCRATCH: a record's scratch lies outside its class's buffer in " + launch_name(cur));
                prev_end = std::max(prev_end, off + words);
            }
        }
        if (scratch && prev_end) { cur_prog = 0x7FFFFFFFu; write(RK_SCRATCH, cls, NONE, "class scratch"); }
    }

    ScheduleReport run(uint32_t drop_wait) {
        replay();
        std::array<uint32_t, N_SCHED_STREAMS> zero{};
        std::array<std::array<uint32_t, N_SCHED_STREAMS>, N_SCHED_STREAMS> clock{};  // per stream: what happens before its next launch
        uint32_t seq[N_SCHED_STREAMS] = {0, 0, 0, 0, 0, 0};
        const uint32_t n_events = 2 * p.n_levels + 1 + 4 * p.n_levels;
        std::vector<std::array<uint32_t, N_SCHED_STREAMS>> ev(n_events, zero);
        std::vector<uint8_t> ev_set(n_events, 0);
        auto new_launch = [&](uint8_t stream, uint8_t op, uint32_t level, uint32_t step) {
            Launch l;
            l.stream = stream; l.op = op; l.level = level; l.step = step;
            l.seq = ++seq[stream];
            clock[stream][stream] = l.seq;
            l.clock = clock[stream];
            l.clock[stream] = l.seq - 1;  // (hb(): strictly earlier launches of its own stream)
            launches.push_back(l);
            cur = (int)launches.size() - 1;
            cur_prog = 0;
            rep.n_launches++;
        };
        // the import that precedes the solve on the main stream: the initial witnesses' rows and byte planes
        new_launch(SS_MAIN, 0xFE, 0, 0);
        for (uint32_t w : p.initial_ids) {
            const uint32_t row = row_of(w);
            if (row == NONE) { finding("initial witness " + std::to_string(w) + " has no row"); continue; }
            write(RK_W, row, w, "import");
            if (!p.byte_plane_of.empty() && p.byte_plane_of[w] != NONE) write(RK_BYTE_PLANE, p.byte_plane_of[w], w, "import");
        }
        int reset_launch = -1;
        uint32_t wait_index = 0;
        for (size_t si = 0; si < sch.steps.size(); si++) {
            const SchedStep &st = sch.steps[si];
            if (st.stream >= N_SCHED_STREAMS) { finding("step on an unknown stream"); continue; }
            if (st.kind == SK_RECORD) {
                if (st.event >= n_events) { finding("record of an event outside the handle's pools"); continue; }
                ev[st.event] = clock[st.stream];
                ev_set[st.event] = 1;
                continue;
            }
            if (st.kind == SK_WAIT) {
                rep.n_waits++;
                if (wait_index++ == drop_wait) {  // the mutation: this wait was never enqueued
                    bool implied = st.event < n_events && ev_set[st.event];
                    for (int q = 0; q < N_SCHED_STREAMS && implied; q++) implied = ev[st.event][q] <= clock[st.stream][q];
                    char buf[160];
                    snprintf(buf, sizeof buf, "(mutation: wait #%u dropped -- step %zu, stream %u waits for event %u%s)\n", drop_wait, si, st.stream, st.event,
                             implied ? "; everything behind that event was already ordered before this stream: a redundant wait" : "");
                    rep.text += buf;
                    continue;
                }
                if (st.event >= n_events || !ev_set[st.event]) { finding("step " + std::to_string(si) + " waits for an event that has not been recorded in this solve (a no-op on the device)"); continue; }
                for (int q = 0; q < N_SCHED_STREAMS; q++) clock[st.stream][q] = std::max(clock[st.stream][q], ev[st.event][q]);
                continue;
            }
            new_launch(st.stream, st.op, st.level, (uint32_t)si);
            if (st.op == SO_EVENT_RESET) { reset_launch = cur; continue; }
            if (reset_launch < 0 || !hb(reset_launch, cur)) finding("EVENT WORDS: " + launch_name(cur) + " may flag instances before the event words are reset");
            switch (st.op) {
            case SO_GATES: case SO_GATES_LIGHT:
                if ((uint64_t)st.first + st.count > p.gate_offset.size()) { finding("gate launch runs past the gate list"); break; }
                for (uint32_t g = st.first; g < st.first + st.count; g++) { cur_prog = g; walk_wave_program(g); }
                if (st.op == SO_GATES_LIGHT) walk_class_records(st, CLS_LIGHT, st.first2, st.count2);
                break;
            case SO_INVERSE:
                if ((uint64_t)st.first + st.count > p.dyn_offset.size()) { finding("inversion launch runs past the job list"); break; }
                // (a wave walks a chunk of jobs in order and parks prefix products in the jobs' own rows: one program per chunk; chunks own disjoint rows)
                for (uint32_t j = st.first; j < st.first + st.count; j++) { cur_prog = 0x20000000u + j; walk_inversion(j); }
                break;
            case SO_TRUNCATE: break;
            default: walk_class_records(st, st.cls, st.first, st.count); break;
            }
        }
        // the end of the solve on the main stream: the host reads the flag count, results and kept witnesses leave, the next tile's import overwrites the
        // initial rows and the byte planes -- every launch of every stream must be ordered before it
        new_launch(SS_MAIN, 0xFF, p.n_levels, (uint32_t)sch.steps.size());
        for (int id = 0; id + 1 < (int)launches.size(); id++)
            if (!hb(id, cur)) { finding("JOIN: " + launch_name(id) + " is not ordered before the end of the solve"); break; }
        for (uint32_t w : p.initial_ids) {
            const uint32_t row = row_of(w);
            if (row == NONE) continue;
            write(RK_W, row, w, "next import");
            if (!p.byte_plane_of.empty() && p.byte_plane_of[w] != NONE) write(RK_BYTE_PLANE, p.byte_plane_of[w], w, "next import");
        }
        // every witness the plan says the level path assigns was written by it (an opcode the schedule lost would show here)
        for (uint32_t w = 0; w < p.n_witnesses; w++)
            if (producer2[w] < NONE - 1 && !wrote[w]) { finding("COVERAGE: witness " + std::to_string(w) + " is assigned by opcode " + std::to_string(producer2[w]) + " in order but no launch of the schedule writes it"); break; }
        char tail[160];
        snprintf(tail, sizeof tail, "%llu launches, %llu waits, %llu records, %llu accesses checked; %u finding(s)", (unsigned long long)rep.n_launches, (unsigned long long)rep.n_waits,
                 (unsigned long long)rep.n_records, (unsigned long long)rep.n_accesses, rep.n_findings);
        rep.text += tail;
        return rep;
    }
};

}  // namespace

ScheduleReport check_level_schedule(const Plan &p, const LaunchLayout &lay, const LevelSchedule &s, uint32_t drop_wait) {
    Checker c(p, lay, s);
    return c.run(drop_wait);
}

}  // namespace acvm
