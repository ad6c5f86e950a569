// grumpkin_host.cpp -- one-time construction of the Grumpkin lookup tables the device kernels of
// Pedersen / FixedBaseScalarMul / SchnorrVerify read (ops_grumpkin.hpp). Product code, shares nothing with oracle/.
//
// What the reference delegates to barretenberg's acvm_backend.wasm (barretenberg_blackbox_solver/src/wasm/
// {pedersen.rs:14-35, scalar_mul.rs:17-65, schnorr.rs:68-103}; algorithms restated in SURVEY.md Appendix A):
//   * curve y^2 = x^3 - 17 over BN254-Fr, generator G = (1, sqrt(-16)) (scalar_mul.rs:77-78)
//   * 30 derived generators D[i]: x = keccak256(be64(seed) || 0^24) read as a little-endian integer, top bit = parity
//     of y, seeds 1, 2, ... skipping non-residues (SURVEY A.2 derive_generators)
//   * plookup Pedersen: 9-bit slices index tables of k * D[i], k = 1..512
//   * fixed-base multiplications (G, and D[0], D[3], D[6] of the Schnorr hash ladder) use 8-bit window tables
//     T[w][d-1] = d * 2^(8w) * P, so a 256-bit scalar costs 32 mixed additions and no doubling.
#include "grumpkin_host.hpp"
#include "tuning.hpp"
#include <cstdlib>
#include "fr_host.hpp"
#include <hip/hip_runtime.h>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace acvm {
void launch_grumpkin_win16_table(hipStream_t s, const GrumpkinTables &T, uint4 *out);  // kernels_grumpkin.hip
namespace {

// ---- Keccak-256 (Keccak-f[1600], rate 136, pad 0x01 .. 0x80), host
void keccak_f(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    auto rotl = [](uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; };
    for (int r = 0; r < 24; r++) {
        uint64_t C[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int i = 0; i < 25; i++) s[i] ^= C[(i % 5 + 4) % 5] ^ rotl(C[(i % 5 + 1) % 5], 1);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(s[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) s[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[r];
    }
}
void keccak256_32(const uint8_t in[32], uint8_t out[32]) {  // single-block message of 32 bytes
    uint64_t s[25] = {0};
    uint8_t blk[136] = {0};
    memcpy(blk, in, 32);
    blk[32] = 0x01;
    blk[135] |= 0x80;
    for (int i = 0; i < 17; i++) {
        uint64_t v = 0;
        for (int k = 0; k < 8; k++) v |= (uint64_t)blk[8 * i + k] << (8 * k);
        s[i] ^= v;
    }
    keccak_f(s);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(s[i / 8] >> (8 * (i % 8)));
}

// ---- field helpers
FrH pow_u(const FrH &a, const uint64_t e[4]) {
    FrH r = frh::one();
    for (int i = 255; i >= 0; i--) {
        r = frh::mul(r, r);
        if ((e[i / 64] >> (i % 64)) & 1) r = frh::mul(r, a);
    }
    return r;
}
// Tonelli-Shanks (p - 1 = 2^28 * t, 5 is a non-residue). false if a is not a square.
bool sqrt_fr(const FrH &a, FrH &out) {
    if (a.is_zero()) { out = a; return true; }
    uint64_t pm1[4] = {frh::P[0] - 1, frh::P[1], frh::P[2], frh::P[3]};
    uint64_t t[4], half[4], t1h[4];
    for (int i = 0; i < 4; i++) t[i] = (pm1[i] >> 28) | (i < 3 ? pm1[i + 1] << 36 : 0);
    for (int i = 0; i < 4; i++) half[i] = (pm1[i] >> 1) | (i < 3 ? pm1[i + 1] << 63 : 0);
    uint64_t tp1[4] = {t[0] + 1, t[1], t[2], t[3]};
    for (int i = 0; i < 4; i++) t1h[i] = (tp1[i] >> 1) | (i < 3 ? tp1[i + 1] << 63 : 0);
    const FrH one = frh::one();
    if (pow_u(a, half) != one) return false;
    FrH c = pow_u(frh::from_u64(5), t), x = pow_u(a, t1h), b = pow_u(a, t);
    int m = 28;
    while (b != one) {
        int i = 0;
        FrH b2 = b;
        while (b2 != one) { b2 = frh::mul(b2, b2); i++; }
        FrH e = c;
        for (int k = 0; k < m - i - 1; k++) e = frh::mul(e, e);
        x = frh::mul(x, e);
        c = frh::mul(e, e);
        b = frh::mul(b, c);
        m = i;
    }
    out = x;
    return true;
}

// ---- curve arithmetic (Jacobian, a = 0)
struct Aff { FrH x, y; };
struct Jac { FrH X, Y, Z; bool inf() const { return Z.is_zero(); } };
Jac jinf() { return Jac{frh::one(), frh::one(), frh::zero()}; }
Jac from_aff(const Aff &p) { return Jac{p.x, p.y, frh::one()}; }
Jac dbl(const Jac &p) {
    if (p.inf() || p.Y.is_zero()) return jinf();
    using namespace frh;
    FrH A = mul(p.X, p.X), B = mul(p.Y, p.Y), C = mul(B, B);
    FrH t = add(p.X, B);
    t = sub(sub(mul(t, t), A), C);
    FrH D = add(t, t), E = add(add(A, A), A), F = mul(E, E);
    FrH X3 = sub(sub(F, D), D);
    FrH C8 = add(C, C);
    C8 = add(C8, C8);
    C8 = add(C8, C8);
    FrH Y3 = sub(mul(E, sub(D, X3)), C8);
    FrH Z3 = mul(p.Y, p.Z);
    Z3 = add(Z3, Z3);
    return Jac{X3, Y3, Z3};
}
Jac addj(const Jac &p, const Jac &q) {
    if (p.inf()) return q;
    if (q.inf()) return p;
    using namespace frh;
    FrH Z1Z1 = mul(p.Z, p.Z), Z2Z2 = mul(q.Z, q.Z);
    FrH U1 = mul(p.X, Z2Z2), U2 = mul(q.X, Z1Z1);
    FrH S1 = mul(mul(p.Y, q.Z), Z2Z2), S2 = mul(mul(q.Y, p.Z), Z1Z1);
    FrH H = sub(U2, U1), r = sub(S2, S1);
    if (H.is_zero()) return r.is_zero() ? dbl(p) : jinf();
    FrH I = add(H, H);
    I = mul(I, I);
    FrH J = mul(H, I);
    r = add(r, r);
    FrH V = mul(U1, I);
    FrH X3 = sub(sub(sub(mul(r, r), J), V), V);
    FrH t = mul(S1, J);
    FrH Y3 = sub(mul(r, sub(V, X3)), add(t, t));
    FrH Z3 = add(p.Z, q.Z);
    Z3 = mul(sub(sub(mul(Z3, Z3), Z1Z1), Z2Z2), H);
    return Jac{X3, Y3, Z3};
}
// batch normalisation (Montgomery's trick); every point must be finite
void to_affine_batch(const std::vector<Jac> &in, std::vector<Aff> &out) {
    size_t n = in.size();
    out.resize(n);
    std::vector<FrH> prefix(n);
    FrH acc = frh::one();
    for (size_t i = 0; i < n; i++) { prefix[i] = acc; acc = frh::mul(acc, in[i].Z); }
    FrH inv = frh::inverse(acc);
    for (size_t i = n; i-- > 0;) {
        FrH zi = frh::mul(inv, prefix[i]);
        inv = frh::mul(inv, in[i].Z);
        FrH zi2 = frh::mul(zi, zi);
        out[i].x = frh::mul(in[i].X, zi2);
        out[i].y = frh::mul(in[i].Y, frh::mul(zi2, zi));
    }
}

struct HostTables {
    std::vector<uint32_t> words;  // all tables, 16 u32 per affine point (x then y, Montgomery limbs little-endian)
    size_t ped_off, win_off, small_off, skew_off;
    bool ok = false;
};

void put_point(std::vector<uint32_t> &w, size_t idx, const Aff &p) {  // device Montgomery form (R = 2^261)
    const FrH x = frh::to_device_form(p.x), y = frh::to_device_form(p.y);
    memcpy(&w[idx * 16], x.l, 32);
    memcpy(&w[idx * 16 + 8], y.l, 32);
}

HostTables build_host_tables() {
    HostTables T;
    const FrH b_coef = frh::neg(frh::from_u64(17));
    // G = (1, y), y^2 = -16 (scalar_mul.rs:77-78)
    static const uint8_t gy_be[32] = {0, 0, 0, 0, 0, 0, 0, 0x02, 0xcf, 0x13, 0x5e, 0x75, 0x06, 0xa4, 0x5d, 0x63,
                                      0x2d, 0x27, 0x0d, 0x45, 0xf1, 0x18, 0x12, 0x94, 0x83, 0x3f, 0xc4, 0x8d, 0x82, 0x3f, 0x27, 0x2c};
    Aff G{frh::one(), frh::from_be_bytes32_reduce(gy_be, 32)};
    if (frh::mul(G.y, G.y) != frh::add(frh::one(), b_coef)) return T;
    // derived generators
    std::vector<Aff> gens;
    for (uint64_t seed = 1; gens.size() < GRUMPKIN_N_GENERATORS && seed < 1000; seed++) {
        uint8_t buf[32] = {0}, h[32], rev[32];
        for (int i = 0; i < 8; i++) buf[i] = (uint8_t)(seed >> (8 * (7 - i)));
        keccak256_32(buf, h);
        int y_bit = h[31] >> 7;
        for (int i = 0; i < 32; i++) rev[i] = h[31 - i];  // digest read as a little-endian integer
        rev[0] &= 0x7f;
        FrH x = frh::from_be_bytes32_reduce(rev, 32);
        FrH yy = frh::add(frh::mul(frh::mul(x, x), x), b_coef), y;
        if (!sqrt_fr(yy, y)) continue;
        uint64_t yc[4];
        frh::to_canonical(y, yc);
        if ((int)(yc[0] & 1) != y_bit) y = frh::neg(y);
        gens.push_back(Aff{x, y});
    }
    if (gens.size() != GRUMPKIN_N_GENERATORS) return T;

    std::vector<Jac> all;
    // Pedersen tables: k * D[i], k = 1..512
    T.ped_off = 0;
    for (uint32_t i = 0; i < GRUMPKIN_N_GENERATORS; i++) {
        Jac g = from_aff(gens[i]), acc = g;
        for (uint32_t k = 1; k <= GRUMPKIN_PED_ENTRIES; k++) {
            all.push_back(acc);
            acc = addj(acc, g);
        }
    }
    // window tables: bases G, D[0], D[3], D[6]
    T.win_off = all.size();
    const Aff bases[GRUMPKIN_N_WINDOW_BASES] = {G, gens[0], gens[3], gens[6]};
    for (uint32_t t = 0; t < GRUMPKIN_N_WINDOW_BASES; t++) {
        Jac base = from_aff(bases[t]);
        for (uint32_t w = 0; w < 32; w++) {
            Jac acc = base;
            for (uint32_t d = 1; d <= 255; d++) {
                all.push_back(acc);
                acc = addj(acc, base);
            }
            base = acc;  // 256 * base = 2^(8(w+1)) * P
        }
    }
    // small ladder tables: k * D[3j+1], k = 1..15
    T.small_off = all.size();
    for (uint32_t j = 0; j < 3; j++) {
        Jac g = from_aff(gens[3 * j + 1]), acc = g;
        for (uint32_t k = 1; k <= 15; k++) {
            all.push_back(acc);
            acc = addj(acc, g);
        }
    }
    T.skew_off = all.size();
    for (uint32_t j = 0; j < 3; j++) all.push_back(from_aff(gens[3 * j + 2]));
    for (auto &p : all)
        if (p.inf()) return T;
    std::vector<Aff> aff;
    to_affine_batch(all, aff);
    T.words.resize(aff.size() * 16);
    for (size_t i = 0; i < aff.size(); i++) put_point(T.words, i, aff[i]);
    // spot check: every 97th point is on the curve
    for (size_t i = 0; i < aff.size(); i += 97) {
        FrH l = frh::mul(aff[i].y, aff[i].y), r = frh::add(frh::mul(frh::mul(aff[i].x, aff[i].x), aff[i].x), b_coef);
        if (l != r) return T;
    }
    T.ok = true;
    return T;
}

// ---- the per-device table sets. One set per HIP device, built on first use and kept until the caller releases it
// (acvm_device_release_tables, or the last handle of the device with tuning tables_keep = 0). Builds take the SET's lock only -- the handles
// acvm_node_new creates side by side on eight devices build their tables in parallel -- and run on the set's own stream, which is
// synchronised instead of the device: a handle already solving on the device is not stalled by a second handle's first use of a table.
std::once_flag g_host_once;
HostTables g_host;
const HostTables &host_tables() {
    std::call_once(g_host_once, [] { g_host = build_host_tables(); });
    return g_host;
}

struct DeviceTableSet {
    std::mutex mu;
    hipStream_t build = nullptr;
    bool base_built = false;
    uint32_t *base_alloc = nullptr;   // ped | win | small | skew, one allocation
    GrumpkinTables t{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t *ecdsa = nullptr;        // generator tables of the two ECDSA curves (kernels_ecdsa.hip)
    uint32_t users = 0;               // live batch handles (and probes in flight) that hold pointers into the set
};
std::mutex g_sets_mu;  // guards the map only
std::map<int, std::unique_ptr<DeviceTableSet>> g_sets;

DeviceTableSet *table_set(int dev) {
    std::lock_guard<std::mutex> lk(g_sets_mu);
    auto &p = g_sets[dev];
    if (!p) p = std::make_unique<DeviceTableSet>();
    return p.get();
}
DeviceTableSet *current_set(int *dev_out = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev_out) *dev_out = dev;
    return table_set(dev);
}
bool ensure_stream(DeviceTableSet &S) { return S.build || hipStreamCreateWithFlags(&S.build, hipStreamNonBlocking) == hipSuccess; }

// (S.mu held) the four host-built tables and, with tuning win16, the 16-bit window tables
bool ensure_base(DeviceTableSet &S) {
    if (S.base_built) return true;
    const HostTables &H = host_tables();
    if (!H.ok || !ensure_stream(S)) return false;
    uint32_t *d = nullptr;
    if (hipMalloc((void **)&d, H.words.size() * 4) != hipSuccess) return false;
    if (hipMemcpyAsync(d, H.words.data(), H.words.size() * 4, hipMemcpyHostToDevice, S.build) != hipSuccess || hipStreamSynchronize(S.build) != hipSuccess) {
        hipFree(d);
        return false;
    }
    S.base_alloc = d;
    GrumpkinTables &t = S.t;
    t.ped = (const uint4 *)(d + H.ped_off * 16);
    t.win = (const uint4 *)(d + H.win_off * 16);
    t.small = (const uint4 *)(d + H.small_off * 16);
    t.skew = (const uint4 *)(d + H.skew_off * 16);
    t.ped2 = nullptr;
    t.win16 = nullptr;
    t.pedw = nullptr;
    {   // 16-bit window tables of the fixed bases (s * G of SchnorrVerify and FixedBaseScalarMul, the three generators of the hash
        // ladder): memory for arithmetic, like the pair table -- 16 mixed additions per 256-bit scalar instead of 32. Without the
        // memory the 8-bit windows stay in use.
        uint4 *w16 = nullptr;
        const size_t entries = (size_t)GRUMPKIN_N_WINDOW_BASES * GRUMPKIN_WIN16_STRIDE;
        if (tuning().win16 && hipMalloc((void **)&w16, entries * 64) == hipSuccess) {
            launch_grumpkin_win16_table(S.build, t, w16);
            if (hipGetLastError() == hipSuccess && hipStreamSynchronize(S.build) == hipSuccess) t.win16 = w16;
            else hipFree(w16);
        }
    }
    S.base_built = true;
    return true;
}

void free_set(DeviceTableSet &S) {  // (S.mu held, no users)
    for (const void *p : {(const void *)S.base_alloc, (const void *)S.t.ped2, (const void *)S.t.win16, (const void *)S.t.pedw, (const void *)S.ecdsa})
        if (p) hipFree(const_cast<void *>(p));
    if (S.build) hipStreamDestroy(S.build);
    S.build = nullptr;
    S.base_alloc = nullptr;
    S.ecdsa = nullptr;
    S.base_built = false;
    S.t = GrumpkinTables{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
}

}  // namespace

void launch_pedersen_pair_table(hipStream_t s, const GrumpkinTables &T, uint4 *out);  // kernels_grumpkin.hip
void launch_pedersen_window_table(hipStream_t s, const GrumpkinTables &T, uint4 *out, uint32_t *n_infinite);  // kernels_grumpkin.hip
void launch_ecdsa_gtable(hipStream_t s, uint32_t *out);  // kernels_ecdsa.hip
size_t ecdsa_gtable_bytes();

bool grumpkin_tables(GrumpkinTables *out) {
    DeviceTableSet *S = current_set();
    if (!S) return false;
    std::lock_guard<std::mutex> lk(S->mu);
    if (!ensure_base(*S)) return false;
    *out = S->t;
    return true;
}

bool grumpkin_pair_table(GrumpkinTables *out) {
    DeviceTableSet *S = current_set();
    if (!S) return false;
    std::lock_guard<std::mutex> lk(S->mu);
    if (!ensure_base(*S)) return false;
    GrumpkinTables &t = S->t;
    if (!t.ped2) {
        uint4 *d = nullptr;
        const size_t entries = (size_t)30 << GRUMPKIN_PED2_LOG2;
        if (hipMalloc((void **)&d, entries * 64) != hipSuccess) return false;
        launch_pedersen_pair_table(S->build, t, d);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(S->build) != hipSuccess) { hipFree(d); return false; }
        t.ped2 = d;
    }
    *out = t;
    return true;
}

bool grumpkin_window_table(GrumpkinTables *out) {
    DeviceTableSet *S = current_set();
    if (!S) return false;
    std::lock_guard<std::mutex> lk(S->mu);
    if (!ensure_base(*S)) return false;
    GrumpkinTables &t = S->t;
    if (!t.pedw) {
        uint4 *d = nullptr;
        uint32_t *d_bad = nullptr, bad = 1;
        const size_t entries = (size_t)2 * GRUMPKIN_PEDW_WINDOWS << GRUMPKIN_PEDW_BITS;
        // The 23.6 GB table is an optimisation of one kernel; the 503 MB pair table serves the same kernel. It is built only where it leaves room:
        // a quarter of the device's memory must stay free behind it (the handle that asks still allocates its inverse rows and side tables, and on a
        // shared GPU -- several processes, several handles -- a table that just fits starves whoever allocates next).
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < entries * 64 + total_b / 4) return false;
        if (hipMalloc((void **)&d, entries * 64) != hipSuccess) return false;
        if (hipMalloc((void **)&d_bad, 4) != hipSuccess || hipMemsetAsync(d_bad, 0, 4, S->build) != hipSuccess) { hipFree(d); hipFree(d_bad); return false; }
        launch_pedersen_window_table(S->build, t, d, d_bad);
        const bool ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, S->build) == hipSuccess &&
                        hipStreamSynchronize(S->build) == hipSuccess && bad == 0;  // (no entry may be the identity)
        hipFree(d_bad);
        if (!ok) { hipFree(d); return false; }
        t.pedw = d;
    }
    *out = t;
    return true;
}

const uint32_t *ecdsa_generator_tables() {
    DeviceTableSet *S = current_set();
    if (!S) return nullptr;
    std::lock_guard<std::mutex> lk(S->mu);
    if (S->ecdsa) return S->ecdsa;
    if (!ensure_stream(*S)) return nullptr;
    uint32_t *d = nullptr;
    if (hipMalloc((void **)&d, ecdsa_gtable_bytes()) != hipSuccess) return nullptr;
    launch_ecdsa_gtable(S->build, d);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(S->build) != hipSuccess) { hipFree(d); return nullptr; }
    S->ecdsa = d;
    return d;
}

void device_tables_retain(int device) {
    DeviceTableSet *S = table_set(device);
    std::lock_guard<std::mutex> lk(S->mu);
    S->users++;
}
void device_tables_unref(int device) {
    DeviceTableSet *S = table_set(device);
    std::lock_guard<std::mutex> lk(S->mu);
    if (S->users) S->users--;
    if (!S->users && !tuning().tables_keep) {
        int prev = 0;
        const bool have_prev = hipGetDevice(&prev) == hipSuccess;
        if (hipSetDevice(device) == hipSuccess) free_set(*S);
        if (have_prev) hipSetDevice(prev);
    }
}
// 0 = freed (or nothing to free), 1 = handles of the device still hold the tables, -1 = the device could not be selected
int device_tables_free(int device, size_t *bytes_freed) {
    DeviceTableSet *S = table_set(device);
    std::lock_guard<std::mutex> lk(S->mu);
    if (bytes_freed) *bytes_freed = 0;
    if (S->users) return 1;
    if (!S->base_built && !S->ecdsa) return 0;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    if (hipSetDevice(device) != hipSuccess) return -1;
    size_t before = 0, after = 0, total = 0;
    hipMemGetInfo(&before, &total);
    free_set(*S);
    hipMemGetInfo(&after, &total);
    if (bytes_freed) *bytes_freed = after > before ? after - before : 0;
    if (have_prev) hipSetDevice(prev);
    return 0;
}
// bytes the fixed tables of a Grumpkin / ECDSA circuit would ADD to the current device (what is not built yet), for memory sizing
size_t device_tables_missing_bytes(bool grumpkin, bool pedersen_level, bool window_table, bool ecdsa) {
    DeviceTableSet *S = current_set();
    if (!S) return 0;
    std::lock_guard<std::mutex> lk(S->mu);
    size_t need = 0;
    if (grumpkin && !S->base_built) need += host_tables().words.size() * 4 + (tuning().win16 ? (size_t)GRUMPKIN_N_WINDOW_BASES * GRUMPKIN_WIN16_STRIDE * 64 : 0);
    if (grumpkin && pedersen_level) {
        if (window_table && !S->t.pedw) need += ((size_t)2 * GRUMPKIN_PEDW_WINDOWS << GRUMPKIN_PEDW_BITS) * 64;
        else if (!window_table && !S->t.ped2) need += ((size_t)30 << GRUMPKIN_PED2_LOG2) * 64;
    }
    if (ecdsa && !S->ecdsa) need += ecdsa_gtable_bytes();
    return need;
}

// host copy of a table point (tests / self check): which = 0 ped, 1 win, 2 small, 3 skew
bool grumpkin_host_point(uint32_t which, uint32_t index, uint8_t out_be[64]) {
    const HostTables &g_host = host_tables();
    if (!g_host.ok) return false;
    size_t off = which == 0 ? g_host.ped_off : which == 1 ? g_host.win_off : which == 2 ? g_host.small_off : g_host.skew_off;
    size_t idx = off + index;
    if ((idx + 1) * 16 > g_host.words.size()) return false;
    for (int c = 0; c < 2; c++) {
        FrH v;
        memcpy(v.l, &g_host.words[idx * 16 + 8 * c], 32);
        uint64_t can[4];
        frh::to_canonical(frh::from_device_form(v), can);
        for (int i = 0; i < 32; i++) out_be[32 * c + 31 - i] = (uint8_t)(can[i / 8] >> (8 * (i % 8)));
    }
    return true;
}

}  // namespace acvm
