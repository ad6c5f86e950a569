// Host-side execution of the Arithmetic level program: the planner's gate records (plan.cpp) run through the SAME record evaluation that
// arith_level_kernel runs (gate_eval.hpp is __host__ __device__) over a witness table in host memory, with the planner's bounds for the
// relaxed rows CHECKED at every store and every stored row replaced by an ADVERSARIAL representative of its residue: the largest
// value + t p below the bound the planner claims for it (and below 2^256), so that every later gate meets the worst operand its record was
// sized for. Run by tests/test_gate_eval_on_host.py, which compares the canonical values with the CPU oracle's witness map.
//   gate_host_test <in> <out> [seed]
//   in:  u32 n_circuit_bytes, circuit bytes, u32 n_ids, ids, u32 B, B x n_ids x 32 big-endian values
//   out: u32 n_witnesses, u32 B, then per instance: u32 flagged (first opcode that left the generic path or 0xFFFFFFFF), n_witnesses x (u8 produced, 32 bytes BE)
// Arithmetic-only circuits (the level program of every other opcode class needs the device).
#include "../acvm_amd/csrc/fr_device.hpp"
#include "../acvm_amd/csrc/fr_host.hpp"
#include "../acvm_amd/csrc/gate_eval.hpp"
#include "../acvm_amd/csrc/plan.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
using namespace acvm;

static uint64_t rng_state = 1;
static uint64_t sm() {
    rng_state += 0x9E3779B97F4A7C15ULL;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static Fr dev_of(const FrH &h) {
    const FrH a = frh::to_device_form(h);
    Fr r;
    for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)a.l[i]; r.v[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
    return r;
}
static FrH host_of(const Fr &d) {  // any representative below 2^256 of a device-form residue -> the planner's form
    uint64_t x[4];
    for (int i = 0; i < 4; i++) x[i] = (uint64_t)d.v[2 * i] | (uint64_t)d.v[2 * i + 1] << 32;
    while (frh::geq_p(x)) frh::sub4(x, x, frh::P);
    FrH h;
    memcpy(h.l, x, sizeof x);
    return frh::from_device_form(h);
}
// 320-bit helpers on 10 x u32 (values up to 2^261 times small factors)
struct Big { uint32_t v[10]; };
static Big big_of29(const Fr29 &a) {  // sum a.v[i] 2^(29 i), limbs normalised below the top one (the top one below 2^32)
    Big r{};
    unsigned __int128 acc = 0;
    int bits = 0, out = 0;
    for (int i = 0; i < 9; i++) {
        acc |= (unsigned __int128)a.v[i] << bits;
        bits += i < 8 ? 29 : 32;
        while (bits >= 32 && out < 10) { r.v[out++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
    }
    while (out < 10) { r.v[out++] = (uint32_t)acc; acc >>= 32; }
    return r;
}
static Big big_mul_small(const Big &a, uint32_t k) {
    Big r;
    uint64_t c = 0;
    for (int i = 0; i < 10; i++) { c += (uint64_t)a.v[i] * k; r.v[i] = (uint32_t)c; c >>= 32; }
    return r;
}
static Big big_p() {
    Big r{};
    for (int i = 0; i < 8; i++) r.v[i] = fr_p(i);
    return r;
}
static bool big_lt(const Big &a, const Big &b) {
    for (int i = 9; i >= 0; i--)
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
    return false;
}
static Big big_add(const Big &a, const Big &b) {
    Big r;
    uint64_t c = 0;
    for (int i = 0; i < 10; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    return r;
}
static Fr29 fr29_of_big(const Big &a) {  // below 2^261
    Fr29 r;
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = a.v[w] | (uint64_t)a.v[w + 1] << 32;
        r.v[i] = (uint32_t)(two >> sh) & 0x1fffffffu;
    }
    return r;
}

struct HostLoader {
    const std::vector<Fr> *W, *Inv;
    const std::vector<uint32_t> *consts;
    uint32_t B, j;
    bool force_any;
    Fr29 load(uint32_t slot) const { return fr29_from((*W)[(size_t)slot * B + j]); }
    Fr29 load_inverse(uint32_t slot) const { return fr29_from((*Inv)[(size_t)slot * B + j]); }
    GateWords constant(uint32_t idx) const { return consts->data() + (size_t)idx * 8; }
    bool any(bool x) const { return x || force_any; }  // (a lane whose neighbours need the last subtraction runs it too)
};

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in out [seed]\n", argv[0]); return 2; }
    if (argc > 3) rng_state = strtoull(argv[3], nullptr, 0);
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t at = 0;
    auto u32 = [&]() { uint32_t x; memcpy(&x, &in[at], 4); at += 4; return x; };
    const uint32_t nc = u32();
    std::string err;
    auto circ = circuit_from_bytes(&in[at], nc, err);
    at += nc;
    if (!circ) { fprintf(stderr, "circuit: %s\n", err.c_str()); return 1; }
    const uint32_t n_ids = u32();
    std::vector<uint32_t> ids(n_ids);
    for (auto &x : ids) x = u32();
    const uint32_t B = u32();
    const uint8_t *vals = &in[at];
    Plan p = build_plan(*circ, ids.data(), n_ids);
    if (p.truncated_at != 0xFFFFFFFFu || p.n_other_records) { fprintf(stderr, "not a covered arithmetic-only circuit\n"); return 1; }
    const uint32_t nw = p.n_witnesses;
    std::vector<Fr> W((size_t)nw * B, fr_zero()), Inv((size_t)std::max(p.n_inverse_slots, 1u) * B, fr_zero());
    // rows nothing has written yet hold garbage on the device: any 256-bit pattern
    for (auto &x : W)
        for (int i = 0; i < 8; i++) x.v[i] = (uint32_t)sm();
    std::vector<uint32_t> consts(p.constants.size() * 8);
    for (size_t i = 0; i < p.constants.size(); i++) {
        const Fr d = dev_of(p.constants[i]);
        memcpy(&consts[8 * i], d.v, 32);
    }
    std::vector<uint32_t> flagged(B, 0xFFFFFFFFu);
    for (uint32_t j = 0; j < B; j++)
        for (uint32_t k = 0; k < n_ids; k++) W[(size_t)ids[k] * B + j] = dev_of(frh::from_be_bytes32_reduce(vals + ((size_t)j * n_ids + k) * 32, 32));
    const Big P = big_p();
    uint64_t n_store = 0, n_bumped = 0, fails = 0;
    uint32_t worst_slack = 0xFFFFFFFFu;
    for (uint32_t L = 0; L < p.n_levels; L++) {
        // the inversion batch of this level (inverse_batch_kernel): 1 / stored denominator, zero denominators leave the generic path
        for (uint32_t q = p.dyn_level_start[L]; q < p.dyn_level_start[L + 1]; q++) {
            const uint32_t *g = &p.gate_stream[p.dyn_offset[q]];
            for (uint32_t j = 0; j < B; j++) {
                const Fr den = W[(size_t)g[0] * B + j];
                if (p.kbound[g[0]] != GATE_K_CANON) { fails++; printf("FAIL denominator w%u is not canonical (kbound %u)\n", g[0], p.kbound[g[0]]); }
                FrH d = host_of(den);
                if (fr_is_zero(den)) { flagged[j] = std::min(flagged[j], g[1]); d = frh::one(); }
                // the device's table holds a representative below 1.4 p: take the canonical one or canonical + p at random
                Fr inv = dev_of(frh::inverse(d));
                if (sm() & 1) { Fr t; fr_add256(t, inv, fr_modulus()); if (big_lt(big_mul_small(big_of29(fr29_from(t)), 256), big_mul_small(P, GATE_K_INVERSE))) inv = t; }
                Inv[(size_t)g[2] * B + j] = inv;
            }
        }
        for (uint32_t q = p.level_start[L]; q < p.level_start[L + 1]; q++) {
            for (uint32_t j = 0; j < B; j++) {
                const uint32_t *g = &p.gate_stream[p.gate_offset[q]];
                HostLoader ld{&W, &Inv, &consts, B, j, (sm() & 3) == 0};
                Fr29 local = fr29_from(fr_zero());
                bool host = true;
                for (;;) {
                    const uint32_t w0 = g[0], kind = w0 & 0xff, opcode = g[1], out = g[2];
                    Fr29 acc = gate_eval(ld, g, local);
                    const Big v = big_of29(acc);
                    if (kind == 0) {
                        uint32_t z = 0;
                        for (int i = 0; i < 9; i++) z |= acc.v[i];
                        if (!big_lt(v, P)) { fails++; printf("FAIL assert value of opcode %u is not canonical\n", opcode); }
                        if (z) flagged[j] = std::min(flagged[j], opcode);
                    } else {
                        // the planner's bound for this row, and the 256 bits of the row
                        const uint32_t kb = p.kbound[out];
                        n_store++;
                        bool limbs_ok = true;
                        for (int i = 0; i < 8; i++) limbs_ok &= acc.v[i] < (1u << 29);
                        if (!limbs_ok || !big_lt(big_mul_small(v, 256), big_mul_small(P, kb)) || v.v[8] || v.v[9]) {
                            if (flagged[j] == 0xFFFFFFFFu) { fails++; printf("FAIL opcode %u w%u instance %u: value exceeds the planner's bound %u / 256 p (mode %u)\n", opcode, out, j, kb, (w0 >> GATE_OUT_SHIFT) & 3u); }
                        } else {
                            // slack of the bound: how many p / 256 below it the value is (statistics)
                            Big b = v;
                            // adversarial representative: add p while it stays below the bound and below 2^256
                            for (;;) {
                                const Big nb = big_add(b, P);
                                if (nb.v[8] || nb.v[9] || !big_lt(big_mul_small(nb, 256), big_mul_small(P, kb))) break;
                                b = nb;
                                n_bumped++;
                            }
                            acc = fr29_of_big(b);
                        }
                        W[(size_t)out * B + j] = fr29_pack(acc);
                    }
                    if (!(w0 & GATE_TAIL_FLAG)) break;
                    if (host || (w0 & GATE_SETLOCAL_FLAG)) local = acc;
                    host = false;
                    g += gate_record_words(g);
                }
            }
        }
    }
    (void)worst_slack;
    // canonical values: stored x 1 / scale for the scaled (= relaxed) witnesses
    FILE *o = fopen(argv[2], "wb");
    fwrite(&nw, 4, 1, o);
    fwrite(&B, 4, 1, o);
    for (uint32_t j = 0; j < B; j++) {
        fwrite(&flagged[j], 4, 1, o);
        for (uint32_t w = 0; w < nw; w++) {
            const uint8_t produced = p.producer[w] != 0xFFFFFFFFu;
            uint8_t be[32] = {0};
            if (produced) {
                const Fr s = W[(size_t)w * B + j];
                if (p.unscale_index[w] == 0xFFFFFFFFu && p.kbound[w] != GATE_K_CANON) { fails++; printf("FAIL w%u is relaxed but has no unscale entry\n", w); }
                FrH x = host_of(s);
                if (p.unscale_index[w] != 0xFFFFFFFFu) x = frh::mul(x, p.unscale[p.unscale_index[w]]);
                else if (!fr_eq(s, dev_of(x))) { fails++; printf("FAIL w%u is stored unreduced although readers take it as it is\n", w); }
                uint64_t c[4];
                frh::to_canonical(x, c);
                for (int i = 0; i < 4; i++)
                    for (int k = 0; k < 8; k++) be[31 - 8 * i - k] = (uint8_t)(c[i] >> (8 * k));
            }
            fwrite(&produced, 1, 1, o);
            fwrite(be, 32, 1, o);
        }
    }
    fclose(o);
    printf("stores %llu, representatives raised by p %llu times, out modes asis/weak/canon %u/%u/%u, largest record bound %u/256 p\n", (unsigned long long)n_store,
           (unsigned long long)n_bumped, p.n_gate_out_mode[0], p.n_gate_out_mode[1], p.n_gate_out_mode[2], p.max_gate_bound);
    printf(fails ? "FAILED %llu\n" : "OK\n", (unsigned long long)fails);
    return fails ? 1 : 0;
}
