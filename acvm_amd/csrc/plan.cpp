// plan.cpp -- static replay of the reference's assigned-set bookkeeping + levelisation (see plan.hpp).
#include "plan.hpp"
#include <algorithm>
#include <array>
#include <chrono>
#include <map>

namespace acvm {
namespace {

struct ConstPool {
    std::vector<FrH> &pool;
    std::map<std::array<uint64_t, 4>, uint32_t> index;
    FrH one = frh::one(), minus_one = frh::neg(frh::one());
    explicit ConstPool(std::vector<FrH> &p) : pool(p) {}
    uint32_t coef(const FrH &c) {  // multiplicative coefficient
        if (c == one) return COEF_ONE;
        if (c == minus_one) return COEF_MINUS_ONE;
        return intern(c);
    }
    uint32_t constant(const FrH &c) {  // additive constant
        if (c.is_zero()) return COEF_ZERO;
        return intern(c);
    }
    uint32_t intern(const FrH &c) {
        std::array<uint64_t, 4> k = {c.l[0], c.l[1], c.l[2], c.l[3]};
        auto it = index.find(k);
        if (it != index.end()) return it->second;
        uint32_t id = (uint32_t)pool.size();
        pool.push_back(c);
        index.emplace(k, id);
        return id;
    }
};

struct PendingGate {
    uint32_t level;
    uint32_t opcode;
    std::vector<uint32_t> words;
};

}  // namespace

Plan build_plan(const Circuit &c, const uint32_t *initial_ids, uint32_t n_initial) {
    auto t0 = std::chrono::steady_clock::now();
    Plan p;
    p.n_opcodes = (uint32_t)c.opcodes.size();
    uint32_t nw = c.max_witness + 1;
    for (uint32_t i = 0; i < n_initial; i++) nw = std::max(nw, initial_ids[i] + 1);
    p.n_witnesses = nw;
    p.initial_ids.assign(initial_ids, initial_ids + n_initial);
    p.producer.assign(nw, 0xFFFFFFFFu);
    std::vector<uint8_t> known(nw, 0);
    std::vector<uint32_t> level(nw, 0);
    for (uint32_t i = 0; i < n_initial; i++) {
        known[initial_ids[i]] = 1;
        p.producer[initial_ids[i]] = 0xFFFFFFFEu;
    }
    ConstPool pool(p.constants);
    std::vector<PendingGate> gates;
    gates.reserve(c.opcodes.size());

    // ---- in-order (exact kernel) program: original expressions, no folding
    p.slow_offset.reserve(c.opcodes.size());
    for (uint32_t oi = 0; oi < c.opcodes.size(); oi++) {
        const Opcode &o = c.opcodes[oi];
        p.slow_offset.push_back((uint32_t)p.slow_stream.size());
        if (o.kind != OP_ARITHMETIC) {
            if (p.unsupported.empty())
                p.unsupported = "opcode " + std::to_string(oi) + ": kind " + std::to_string(o.kind) + " has no kernel yet";
            p.slow_stream.push_back(0xFFFFFFFFu);
            continue;
        }
        auto &s = p.slow_stream;
        s.push_back(OP_ARITHMETIC);
        s.push_back((uint32_t)o.expr.mul.size());
        s.push_back((uint32_t)o.expr.lin.size());
        s.push_back(pool.constant(o.expr.qc));
        for (auto &t : o.expr.mul) {
            s.push_back(t.c.is_zero() ? COEF_ZERO : pool.coef(t.c));
            s.push_back(t.l);
            s.push_back(t.r);
        }
        for (auto &t : o.expr.lin) {
            // coefficient, -1/coefficient (so that a solved witness costs one multiplication), witness
            s.push_back(t.c.is_zero() ? COEF_ZERO : pool.coef(t.c));
            s.push_back(t.c.is_zero() ? COEF_ZERO : pool.coef(frh::neg(frh::inverse(t.c))));
            s.push_back(t.w);
        }
    }

    // ---- level-parallel program for the generic instance
    for (uint32_t oi = 0; oi < c.opcodes.size() && p.truncated_at == 0xFFFFFFFFu; oi++) {
        const Opcode &o = c.opcodes[oi];
        if (o.kind != OP_ARITHMETIC) { p.truncated_at = oi; break; }
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
            break;
        }
        PendingGate g;
        g.opcode = oi;
        uint32_t lvl = 0;
        std::vector<uint32_t> reads;
        auto rd = [&](uint32_t w) { lvl = std::max(lvl, level[w]); reads.push_back(w); };
        for (auto &t : prods) { rd(t.a); rd(t.b); }
        for (auto &t : lins) rd(t.a);
        uint32_t kind = GATE_ASSERT;
        FrH scale = frh::one();
        bool scaled = false;
        if (n_unknown == 1) {
            // out = -(sum)/coeff (arithmetic.rs:120) or -(sum)/(c*partner) (:86): fold -1/coeff into every coefficient
            scale = frh::neg(frh::inverse(unk_coef));
            scaled = true;
            if (unk_is_folded) { kind = GATE_SOLVE_DYN; rd(unk_partner); }
            else kind = GATE_SOLVE;
        }
        // zero-coefficient products / linear terms contribute exactly 0: drop them from the device program
        auto sc = [&](const FrH &x) { return scaled ? frh::mul(x, scale) : x; };
        std::vector<uint32_t> pw, lw;
        uint32_t np = 0, nl = 0;
        for (auto &t : prods) {
            if (t.c.is_zero()) continue;
            pw.push_back(pool.coef(sc(t.c))); pw.push_back(t.a); pw.push_back(t.b);
            np++;
        }
        for (auto &t : lins) {
            if (t.c.is_zero()) continue;
            lw.push_back(pool.coef(sc(t.c))); lw.push_back(t.a);
            nl++;
        }
        if (np > 255 || nl > 255) { p.truncated_at = oi; break; }
        g.level = lvl + 1;
        g.words = {kind | np << 8 | nl << 16, oi, kind == GATE_ASSERT ? 0u : unk_w, pool.constant(sc(e.qc)),
                   kind == GATE_SOLVE_DYN ? unk_partner : 0u};
        g.words.insert(g.words.end(), pw.begin(), pw.end());
        g.words.insert(g.words.end(), lw.begin(), lw.end());
        std::sort(reads.begin(), reads.end());
        reads.erase(std::unique(reads.begin(), reads.end()), reads.end());
        uint64_t bytes = 32ull * (reads.size() + (kind == GATE_ASSERT ? 0 : 1));
        p.algorithmic_bytes += bytes;
        (kind == GATE_SOLVE_DYN ? p.dyn_algorithmic_bytes : p.arith_algorithmic_bytes) += bytes;
        if (kind != GATE_ASSERT) {
            known[unk_w] = 1;
            level[unk_w] = g.level;
            p.producer[unk_w] = oi;
        }
        if (kind == GATE_SOLVE_DYN) p.n_dyn_gates++;
        else p.n_fast_gates++;
        gates.push_back(std::move(g));
    }

    // ---- order by (level, program order) and lay out
    std::stable_sort(gates.begin(), gates.end(), [](const PendingGate &a, const PendingGate &b) { return a.level < b.level; });
    uint32_t max_level = 0;
    for (auto &g : gates) max_level = std::max(max_level, g.level);
    p.level_start.assign(max_level + 1, 0);
    p.dyn_level_start.assign(max_level + 1, 0);
    size_t gi = 0;
    for (uint32_t L = 1; L <= max_level; L++) {
        p.level_start[L - 1] = (uint32_t)p.gate_offset.size();
        p.dyn_level_start[L - 1] = (uint32_t)p.dyn_offset.size();
        for (; gi < gates.size() && gates[gi].level == L; gi++) {
            bool dyn = (gates[gi].words[0] & 0xff) == GATE_SOLVE_DYN;
            (dyn ? p.dyn_offset : p.gate_offset).push_back((uint32_t)p.gate_stream.size());
            p.gate_stream.insert(p.gate_stream.end(), gates[gi].words.begin(), gates[gi].words.end());
        }
    }
    p.level_start[max_level] = (uint32_t)p.gate_offset.size();
    p.dyn_level_start[max_level] = (uint32_t)p.dyn_offset.size();
    for (size_t l = 0; l + 1 < p.level_start.size(); l++)
        p.max_level_width = std::max(p.max_level_width, p.level_start[l + 1] - p.level_start[l] + p.dyn_level_start[l + 1] - p.dyn_level_start[l]);
    p.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return p;
}

}  // namespace acvm
