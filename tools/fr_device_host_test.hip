// Host-side execution of the device field library (fr_device.hpp is __host__ __device__) against the
// planner's independent 4x64 implementation (fr_host.hpp). Run by tests/test_fr_device_on_host.py.
#include "../acvm_amd/csrc/fr_device.hpp"
#include "../acvm_amd/csrc/fr_host.hpp"
#include <cstdio>
#include <cstdlib>
using namespace acvm;

static uint64_t sm(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static Fr to_dev(const FrH &h) {  // device Montgomery form (R = 2^261) of the planner's value
    const FrH a = frh::to_device_form(h);
    Fr r;
    for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)a.l[i]; r.v[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
    return r;
}
static bool same(const Fr &d, const FrH &hh) {
    const FrH h = frh::to_device_form(hh);
    for (int i = 0; i < 4; i++)
        if (d.v[2 * i] != (uint32_t)h.l[i] || d.v[2 * i + 1] != (uint32_t)(h.l[i] >> 32)) return false;
    return true;
}
int main() {
    if (!frh::self_check()) { printf("FAIL self_check\n"); return 1; }
    uint64_t s = 12345;
    int fails = 0;
    FrH edge[6];
    edge[0] = frh::zero(); edge[1] = frh::one(); edge[2] = frh::neg(frh::one());
    uint64_t t128[4] = {0, 0, 1, 0}; edge[3] = frh::from_canonical(t128);
    uint64_t t253[4] = {0, 0, 0, 1ULL << 61}; edge[4] = frh::from_canonical(t253);
    edge[5] = frh::from_u64(5);
    for (int it = 0; it < 3000; it++) {
        FrH a, b;
        if (it < 36) { a = edge[it / 6]; b = edge[it % 6]; }
        else {
            uint64_t x[4] = {sm(s), sm(s), sm(s), sm(s) >> 3}, y[4] = {sm(s), sm(s), sm(s), sm(s) >> 3};
            while (frh::geq_p(x)) frh::sub4(x, x, frh::P);
            while (frh::geq_p(y)) frh::sub4(y, y, frh::P);
            a = frh::from_canonical(x); b = frh::from_canonical(y);
        }
        Fr da = to_dev(a), db = to_dev(b);
        if (!same(fr_mul(da, db), frh::mul(a, b))) { fails++; printf("mul mismatch %d\n", it); }
        if (!fr_eq(fr_mul(da, db), fr_mul_portable(da, db))) { fails++; printf("mul vs cios mismatch %d\n", it); }
        {   // working form with unreduced inputs: (a + b + a) * (b + b) through limb-wise sums
            Fr29 x = fr29_from(da), y = fr29_from(db), sx, sy;
            for (int i = 0; i < 9; i++) { sx.v[i] = 2 * x.v[i] + y.v[i]; sy.v[i] = 2 * y.v[i]; }
            for (int i = 0; i < 8; i++) { sx.v[i + 1] += sx.v[i] >> 29; sx.v[i] &= 0x1fffffffu; sy.v[i + 1] += sy.v[i] >> 29; sy.v[i] &= 0x1fffffffu; }
            Fr29 pr = fr29_cond_sub_p(fr29_mul(sx, sy));
            FrH want = frh::mul(frh::add(frh::add(a, a), b), frh::add(b, b));
            if (!same(fr29_pack(pr), want)) { fails++; printf("lazy mul mismatch %d\n", it); }
        }
        {   // weak reduction of a lazy sum: x_j = c_j + m_j p (normalised), v = their limb-wise sum (up to 8 x 16 p), reduced in one step
            Fr29 v;
            for (int i = 0; i < 9; i++) v.v[i] = 0;
            FrH want = frh::zero();
            const int n = 1 + (int)(sm(s) % 7);  // (limbs of the sum stay below 2^32)
            for (int jx = 0; jx < n; jx++) {
                const FrH c = (jx & 1) ? b : a;
                Fr29 x = fr29_from(to_dev(c));
                const int m = (int)(sm(s) % 16);  // + m p as 8 p, 4 p, 2 p, p
                for (int bit = 3; bit >= 0; bit--)
                    if (m >> bit & 1)
                        for (int i = 0; i < 9; i++) x.v[i] += bit ? fr_kp29(bit, i) : fr_p29(i);
                x = fr29_norm(x);
                for (int i = 0; i < 9; i++) v.v[i] += x.v[i];
                want = frh::add(want, c);
            }
            const Fr29 r = fr29_weak(fr29_norm(v));
            bool ok = true;
            for (int i = 0; i < 8; i++) ok = ok && r.v[i] <= 0x1fffffffu;
            Fr29 two_p;  // below 2 p: one conditional subtraction makes it canonical
            for (int i = 0; i < 9; i++) two_p.v[i] = fr_kp29(1, i);
            int cmp = 0;
            for (int i = 8; i >= 0 && !cmp; i--) cmp = r.v[i] < two_p.v[i] ? -1 : r.v[i] > two_p.v[i] ? 1 : 0;
            ok = ok && cmp < 0;
            if (!ok || !same(fr29_pack(fr29_cond_sub_p(r)), want)) { fails++; printf("weak reduction mismatch %d\n", it); }
        }
        {   // truncated reduction: low 29 bits of the canonical value
            Fr one_c = fr_zero(); one_c.v[0] = 1;
            const Fr c = fr_mul(da, one_c);
            if (fr29_redc_low(fr29_from(da)) != (c.v[0] & 0x1fffffffu)) { fails++; printf("redc_low mismatch %d\n", it); }
        }
        if (!same(fr_add(da, db), frh::add(a, b))) { fails++; printf("add mismatch %d\n", it); }
        if (!same(fr_sub(da, db), frh::sub(a, b))) { fails++; printf("sub mismatch %d\n", it); }
        if (!same(fr_neg(da), frh::neg(a))) { fails++; printf("neg mismatch %d\n", it); }
        if (it < 3000) {
            Fr inv = fr_inv(da);
            if (!same(inv, frh::inverse(a))) { fails++; printf("inv mismatch %d\n", it); }
            if (!same(fr_inv_eea(da), frh::inverse(a))) { fails++; printf("inv_eea mismatch %d\n", it); }
        }
        {   // lazy working-form arithmetic at the edges of its contracts: x = a + a (< 2p), y = b + b
            Fr29 x = fr29_norm(fr29_dbll(fr29_from(da))), y = fr29_norm(fr29_dbll(fr29_from(db)));
            const FrH a2 = frh::add(a, a), b2 = frh::add(b, b);
            // 8 (x * y) + 4p - y, reduced two ways
            Fr29 m = fr29_mul(x, y);
            Fr29 t = fr29_norm(fr29_subl(fr29_norm(fr29_dbll(fr29_dbll(fr29_dbll(m)))), y, 1));   // < 8 * 1.1 + 2 = 10.8p, limbs renormalised before the subtraction
            Fr29 u = fr29_mul(t, x);                                                      // inputs < 16p
            FrH want = frh::mul(frh::sub(frh::mul(frh::from_u64(8), frh::mul(a2, b2)), b2), a2);
            if (!same(fr29_pack(fr29_canon(u)), want)) { fails++; printf("lazy chain mismatch %d\n", it); }
            // subtraction below zero and back: (x - y + 2p) - x + 4p == 2p... == -y mod p
            Fr29 d1 = fr29_norm(fr29_subl(x, y, 1));
            Fr29 d2 = fr29_norm(fr29_subl(d1, x, 2));
            if (!same(fr29_pack(fr29_canon(d2)), frh::neg(b2))) { fails++; printf("lazy sub mismatch %d\n", it); }
            if (fr29_is_zero_mod_p(fr29_lt2p(fr29_norm(fr29_subl(x, x, 1)))) != true) { fails++; printf("zero test mismatch %d\n", it); }
            if (fr29_is_zero_mod_p(fr29_mul(x, y)) != (a.is_zero() || b.is_zero())) { fails++; printf("zero test 2 mismatch %d\n", it); }
        }
        {   // squaring, also of an unreduced operand (3a + b, limbs renormalised)
            const Fr29 x = fr29_from(da), y = fr29_from(db);
            if (!same(fr29_pack(fr29_canon(fr29_sqr(x))), frh::mul(a, a))) { fails++; printf("sqr mismatch %d\n", it); }
            const Fr29 z = fr29_norm(fr29_addl(fr29_addl(fr29_dbll(x), x), y));
            const FrH zh = frh::add(frh::add(frh::add(a, a), a), b);
            if (!same(fr29_pack(fr29_canon(fr29_sqr(z))), frh::mul(zh, zh))) { fails++; printf("lazy sqr mismatch %d\n", it); }
        }
        {   // sums of products with one reduction: a*b + b*b + (a+a)*a and a*a + b*a
            const Fr29 x = fr29_from(da), y = fr29_from(db), x2 = fr29_norm(fr29_dbll(x));
            const Fr29 l3[3] = {x, y, x2}, r3[3] = {y, y, x};
            FrH want = frh::add(frh::add(frh::mul(a, b), frh::mul(b, b)), frh::mul(frh::add(a, a), a));
            if (!same(fr29_pack(fr29_canon(fr29_dot<3>(l3, r3))), want)) { fails++; printf("dot3 mismatch %d\n", it); }
            const Fr29 l2[2] = {x, y}, r2[2] = {x, x};
            want = frh::add(frh::mul(a, a), frh::mul(b, a));
            if (!same(fr29_pack(fr29_canon(fr29_dot<2>(l2, r2))), want)) { fails++; printf("dot2 mismatch %d\n", it); }
            const Fr29 l1[1] = {x}, r1[1] = {y};
            if (!same(fr29_pack(fr29_canon(fr29_dot<1>(l1, r1))), frh::mul(a, b))) { fails++; printf("dot1 mismatch %d\n", it); }
        }
        if (fr_is_zero(da) != a.is_zero()) { fails++; printf("is_zero mismatch %d\n", it); }
    }
    // 5^-1 = 0x135b5294...6667 (acvm_js/test/shared/foreign_call.ts)
    {
        Fr one_c = fr_zero(); one_c.v[0] = 1;
        Fr inv5 = fr_mul(fr_inv(to_dev(frh::from_u64(5))), one_c);  // out of Montgomery form: the canonical integer
        const uint32_t expect[8] = {0xc6666667u, 0xe7f3fbd4u, 0xca4a2d06u, 0xa9ae5ce9u, 0x33cd568bu, 0x49b9b57cu, 0x5a13d9aau, 0x135b5294u};
        for (int i = 0; i < 8; i++) if (inv5.v[i] != expect[i]) { fails++; printf("inv5 limb %d mismatch\n", i); }
    }
    if (!same(fr_one(), frh::one())) { fails++; printf("one mismatch\n"); }
    printf(fails ? "FAIL %d\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
