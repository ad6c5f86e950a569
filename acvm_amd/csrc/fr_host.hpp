// fr_host.hpp -- host-side BN254-Fr arithmetic used by the static planner (constant folding of circuit
// coefficients: Montgomery conversion, negation, inverses of constant divisors). Product code: it shares
// nothing with oracle/. Semantics follow acir_field::FieldElement (acir_field/src/generic_ark.rs:242-283,
// 360-406): canonical residues mod p, inverse(0) == 0.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace acvm {

typedef unsigned __int128 u128;

struct FrH {
    uint64_t l[4];  // Montgomery form, R = 2^256
    bool operator==(const FrH &o) const { return memcmp(l, o.l, 32) == 0; }
    bool operator!=(const FrH &o) const { return !(*this == o); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
};

namespace frh {
static constexpr uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                  0x30644e72e131a029ULL};
static constexpr uint64_t N0INV = 0xc2e1f593efffffffULL;  // -p^-1 mod 2^64
// R mod p, R^2 mod p (checked at start-up by frh::self_check)
static constexpr uint64_t R1[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL,
                                   0x0e0a77c19a07df2fULL};
static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL,
                                   0x0216d0b17f4e44a5ULL};

inline bool geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > P[i]) return true;
        if (a[i] < P[i]) return false;
    }
    return true;
}
inline uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
inline uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
inline FrH mul(const FrH &a, const FrH &b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * N0INV;
        c = (u128)m * P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq_p(t)) sub4(t, t, P);
    FrH r;
    memcpy(r.l, t, 32);
    return r;
}
inline FrH zero() { return FrH{{0, 0, 0, 0}}; }
inline FrH one() { return FrH{{R1[0], R1[1], R1[2], R1[3]}}; }
inline FrH add(const FrH &a, const FrH &b) {
    FrH r;
    uint64_t c = add4(r.l, a.l, b.l);
    if (c || geq_p(r.l)) sub4(r.l, r.l, P);
    return r;
}
inline FrH sub(const FrH &a, const FrH &b) {
    FrH r;
    if (sub4(r.l, a.l, b.l)) add4(r.l, r.l, P);
    return r;
}
inline FrH neg(const FrH &a) {
    if (a.is_zero()) return a;
    FrH r;
    sub4(r.l, P, a.l);
    return r;
}
inline FrH from_canonical(const uint64_t v[4]) {
    FrH t, r2;
    memcpy(t.l, v, 32);
    memcpy(r2.l, R2, 32);
    return mul(t, r2);
}
inline void to_canonical(const FrH &a, uint64_t out[4]) {
    FrH o{{1, 0, 0, 0}};
    FrH t = mul(a, o);
    memcpy(out, t.l, 32);
}
inline FrH from_u64(uint64_t v) {
    uint64_t t[4] = {v, 0, 0, 0};
    return from_canonical(t);
}
// from_be_bytes_reduce for <= 32 bytes (generic_ark.rs:281-283)
inline FrH from_be_bytes32_reduce(const uint8_t *b, size_t len) {
    uint64_t v[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < len; i++) {
        size_t pos = len - 1 - i;
        v[pos / 8] |= (uint64_t)b[i] << (8 * (pos % 8));
    }
    while (geq_p(v)) sub4(v, v, P);
    return from_canonical(v);
}
// from_be_bytes_reduce for any length (generic_ark.rs:281-283): Horner over 32-byte blocks, most significant first
inline FrH from_be_bytes_reduce(const uint8_t *b, size_t len) {
    if (len <= 32) return from_be_bytes32_reduce(b, len);
    const size_t head = len % 32 ? len % 32 : 32;
    FrH acc = from_be_bytes32_reduce(b, head);
    const FrH r2{{R2[0], R2[1], R2[2], R2[3]}};  // Montgomery form of R = 2^256
    for (size_t off = head; off < len; off += 32) acc = add(mul(acc, r2), from_be_bytes32_reduce(b + off, 32));
    return acc;
}
inline FrH pow_pm2(const FrH &a) {  // a^(p-2): Fermat inverse (planner only; a handful of calls per circuit)
    uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
    FrH r = one();
    for (int i = 253; i >= 0; i--) {
        r = mul(r, r);
        if ((e[i / 64] >> (i % 64)) & 1) r = mul(r, a);
    }
    return r;
}
// Binary extended Euclid on the Montgomery representative u = aR (an integer below p): u^-1 = a^-1 R^-1, and one Montgomery
// product with R^3 returns a^-1 R. ~4x faster than the Fermat power, which matters because the planner folds -1/coefficient
// into every solved gate (one inversion per distinct coefficient; 1 M-gate circuits plan in seconds instead of half a minute).
inline FrH inverse(const FrH &a) {  // inverse(0) == 0
    if (a.is_zero()) return a;
    uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    memcpy(u, a.l, 32);
    memcpy(v, P, 32);
    auto is_one = [](const uint64_t x[4]) { return x[0] == 1 && !(x[1] | x[2] | x[3]); };
    auto halve = [](uint64_t x[4]) {
        for (int i = 0; i < 3; i++) x[i] = x[i] >> 1 | x[i + 1] << 63;
        x[3] >>= 1;
    };
    auto halve_mod = [&](uint64_t x[4]) {  // x / 2 mod p; x + p < 2^255
        if (x[0] & 1) add4(x, x, P);
        halve(x);
    };
    auto geq = [](const uint64_t x[4], const uint64_t y[4]) {
        for (int i = 3; i >= 0; i--) {
            if (x[i] > y[i]) return true;
            if (x[i] < y[i]) return false;
        }
        return true;
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u[0] & 1)) { halve(u); halve_mod(x1); }
        while (!(v[0] & 1)) { halve(v); halve_mod(x2); }
        if (geq(u, v)) {
            sub4(u, u, v);
            if (sub4(x1, x1, x2)) add4(x1, x1, P);
        } else {
            sub4(v, v, u);
            if (sub4(x2, x2, x1)) add4(x2, x2, P);
        }
    }
    FrH r, r2, r3;
    memcpy(r.l, is_one(u) ? x1 : x2, 32);
    memcpy(r2.l, R2, 32);
    r3 = mul(r2, r2);  // R^2 * R^2 * R^-1 = R^3
    return mul(r, r3);
}
// The device keeps Montgomery representatives with R = 2^261 (fr_device.hpp); the planner computes with R = 2^256.
// x * 2^261 mod p is the R = 2^256 representative of 32 x, and back.
inline FrH to_device_form(const FrH &a) { return mul(a, from_u64(32)); }
inline FrH from_device_form(const FrH &a) {
    static const FrH inv32 = inverse(from_u64(32));
    return mul(a, inv32);
}
inline bool self_check() {
    // R1 = 2^256 mod p, R2 = R1^2 mod p: verify by doubling
    uint64_t x[4] = {1, 0, 0, 0};
    auto dbl = [&]() {
        uint64_t c = add4(x, x, x);
        if (c || geq_p(x)) sub4(x, x, P);
    };
    for (int i = 0; i < 256; i++) dbl();
    if (memcmp(x, R1, 32)) return false;
    for (int i = 0; i < 256; i++) dbl();
    if (memcmp(x, R2, 32)) return false;
    if ((uint64_t)(P[0] * (0 - N0INV)) != 1) return false;
    // the Euclidean inverse against the Fermat power on a few values, and a * a^-1 == 1
    FrH t = from_u64(5);
    for (int i = 0; i < 8; i++) {
        const FrH inv = inverse(t), fermat = pow_pm2(t), prod = mul(t, inv);
        if (memcmp(inv.l, fermat.l, 32) || memcmp(prod.l, R1, 32)) return false;
        t = add(mul(t, t), from_u64(0x9E3779B97F4A7C15ULL + i));
    }
    return true;
}
}  // namespace frh
}  // namespace acvm
