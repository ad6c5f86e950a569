// display.cpp -- see display.hpp
#include "display.hpp"
#include <cstring>

namespace acvm {
namespace {

struct U256 {
    uint64_t l[4];
};
bool is_zero(const U256 &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
// a / d and a % d for a small divisor
uint32_t divmod_small(U256 &a, uint32_t d) {
    unsigned __int128 r = 0;
    for (int i = 3; i >= 0; i--) {
        const unsigned __int128 cur = (r << 64) | a.l[i];
        a.l[i] = (uint64_t)(cur / d);
        r = cur % d;
    }
    return (uint32_t)r;
}
std::string decimal(U256 a) {
    if (is_zero(a)) return "0";
    std::string s;
    while (!is_zero(a)) {
        uint32_t chunk = divmod_small(a, 1000000000u);
        for (int k = 0; k < 9 && (!is_zero(a) || chunk); k++) {
            s.push_back((char)('0' + chunk % 10));
            chunk /= 10;
        }
    }
    return std::string(s.rbegin(), s.rend());
}
unsigned bits(const U256 &a) {
    for (int i = 3; i >= 0; i--)
        if (a.l[i]) return 64u * i + (64u - (unsigned)__builtin_clzll(a.l[i]));
    return 0;
}
unsigned popcount(const U256 &a) { return (unsigned)(__builtin_popcountll(a.l[0]) + __builtin_popcountll(a.l[1]) + __builtin_popcountll(a.l[2]) + __builtin_popcountll(a.l[3])); }
bool low_bits_zero(const U256 &a, unsigned k) {  // a % 2^k == 0, k <= 64
    return k == 64 ? a.l[0] == 0 : (a.l[0] & ((1ull << k) - 1)) == 0;
}
U256 shr(const U256 &a, unsigned k) {  // k <= 64
    U256 r;
    if (k == 64) {
        r.l[0] = a.l[1]; r.l[1] = a.l[2]; r.l[2] = a.l[3]; r.l[3] = 0;
        return r;
    }
    for (int i = 0; i < 4; i++) r.l[i] = (a.l[i] >> k) | (i < 3 && k ? a.l[i + 1] << (64 - k) : 0);
    return r;
}
// generic_ark.rs:476-503
std::string superscript(uint64_t n) {
    static const char *digits[10] = {"⁰", "¹", "²", "³", "⁴", "⁵", "⁶", "⁷", "⁸", "⁹"};
    if (n < 10) return digits[n];
    return superscript(n / 10) + superscript(n % 10);
}

}  // namespace

std::string field_display(const FrH &x) {
    U256 number, minus;
    frh::to_canonical(x, number.l);
    if (is_zero(number)) return "0";
    frh::to_canonical(frh::neg(x), minus.l);
    // "Check if the negative version is smaller to represent": compared by the LENGTH of the decimal strings
    const bool negative = decimal(minus).size() < decimal(number).size();
    const U256 small = negative ? minus : number;
    std::string out = negative ? "-" : "";
    if (popcount(small) == 1) {  // a power of two
        const unsigned bit = bits(small) - 1;
        if (bit < 4) return out + std::to_string(1u << bit);
        return out + "2" + superscript(bit);
    }
    for (unsigned power : {64u, 32u, 16u, 8u, 4u})
        if (low_bits_zero(small, power)) return out + "2" + superscript(power) + "×" + decimal(shr(small, power));
    return out + decimal(small);
}

std::string expression_display(const Expr &e) {
    // to_witness (expression/mod.rs:158-172): no mul terms, one linear term with coefficient one, constant zero
    if (e.mul.empty() && e.lin.size() == 1 && e.lin[0].c == frh::one() && e.qc.is_zero()) return "x" + std::to_string(e.lin[0].w);
    std::string s = "%EXPR [ ";
    for (const MulTerm &t : e.mul) s += "(" + field_display(t.c) + ", _" + std::to_string(t.l) + ", _" + std::to_string(t.r) + ") ";
    for (const LinTerm &t : e.lin) s += "(" + field_display(t.c) + ", _" + std::to_string(t.w) + ") ";
    s += field_display(e.qc) + " ]%";
    return s;
}

}  // namespace acvm
