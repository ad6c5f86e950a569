// Host-side execution of the ECDSA curve arithmetic of the device (acvm_amd/csrc/secp_device.hpp is __host__ __device__): reads one request per
// line from stdin, prints one answer per line; tests/test_secp_device_on_host.py compares with Python integers. Nothing is launched.
//   <c> mul a b | sqr a | add a b | sub a b | inv a | ninv a | nmul a b | sqrt a        -> hex
//   <c> dbl X Y Z | addaff X Y Z x y                                                  -> X Y Z (Jacobian, hex)
//   0 split k                                                                         -> |k1| neg1 |k2| neg2 (secp256k1's endomorphism split)
//   <c> gtab j d                                                                      -> x y
//   <c> verify r s x y_odd n_msg z                                                    -> result panic      (table of the generator built on first use;
//                                                                                        build with -DSECP_GWIN_BITS=8: the device's 16-bit table takes minutes on the host)
#include "../acvm_amd/csrc/secp_device.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
using namespace acvm;

static Fr parse(const char *h) {
    Fr r = fr_zero();
    const size_t n = strlen(h);
    for (size_t i = 0; i < n && i < 64; i++) {
        const char ch = h[n - 1 - i];
        const uint32_t d = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        r.v[i / 8] |= d << (4 * (i % 8));
    }
    return r;
}
static void put(const Fr &a) {
    for (int i = 7; i >= 0; i--) printf("%08x", a.v[i]);
}
template <int C>
static const uint32_t *gtable() {
    static std::vector<uint32_t> t;
    if (t.empty()) {
        t.assign(SECP_GTABLE_WORDS, 0u);
        for (uint32_t j = 0; j < SECP_GWINDOWS; j++)
            for (uint32_t d = 1; d < (1u << SECP_GWIN); d++) {
                const SAff e = secp_gtable_entry<C>(j, d);
                for (int k = 0; k < 8; k++) { t[((j << SECP_GWIN) + d) * 16 + k] = e.x.v[k]; t[((j << SECP_GWIN) + d) * 16 + 8 + k] = e.y.v[k]; }
            }
    }
    return t.data();
}
template <int C>
static void serve(const std::vector<std::string> &w) {
    const std::string &op = w[1];
    auto A = [&](size_t i) { return parse(w[i].c_str()); };
    if (op == "mul") put(sp_mul<C>(A(2), A(3)));
    else if (op == "sqr") put(sp_sqr<C>(A(2)));
    else if (op == "add") put(sp_add<C>(A(2), A(3)));
    else if (op == "sub") put(sp_sub<C>(A(2), A(3)));
    else if (op == "inv") put(sp_inv<C>(A(2)));
    else if (op == "ninv") put(sn_inv<C>(A(2)));
    else if (op == "nmul") put(sn_mul<C>(A(2), A(3)));
    else if (op == "sqrt") put(sp_sqrt_candidate<C>(A(2)));
    // (points enter the domain form of the curve routines -- the residue itself for secp256k1, the Montgomery residue for secp256r1 -- and leave it for the answer)
    else if (op == "dbl") { const SJac r = sj_dbl<C>(SJac{sp_enter<C>(A(2)), sp_enter<C>(A(3)), sp_enter<C>(A(4))}); put(sp_leave<C>(r.X)); printf(" "); put(sp_leave<C>(r.Y)); printf(" "); put(sp_leave<C>(r.Z)); }
    else if (op == "addaff") {
        const SJac r = sj_add_aff<C>(SJac{sp_enter<C>(A(2)), sp_enter<C>(A(3)), sp_enter<C>(A(4))}, SAff{sp_store<C>(sp_enter<C>(A(5))), sp_store<C>(sp_enter<C>(A(6)))});
        put(sp_leave<C>(r.X)); printf(" "); put(sp_leave<C>(r.Y)); printf(" "); put(sp_leave<C>(r.Z));
    }
    else if (op == "gtab") { const SAff e = secp_gtable_entry<C>((uint32_t)atoi(w[2].c_str()), (uint32_t)atoi(w[3].c_str())); put(sp_leave<C>(sp_load(e.x))); printf(" "); put(sp_leave<C>(sp_load(e.y))); }
    else if (op == "verify") {
        uint32_t panic = 0;
        const uint32_t ok = secp_verify<C>(A(2), A(3), A(4), (uint32_t)atoi(w[5].c_str()), (uint32_t)atoi(w[6].c_str()), A(7), gtable<C>(), &panic);
        printf("%u %u", ok, panic);
    } else if (op == "verifyy") {  // the same with the whole of public_key_y given (the decompression shortcut)
        uint32_t panic = 0;
        const Fr y = A(5);
        const uint32_t ok = secp_verify<C>(A(2), A(3), A(4), y.v[0] & 1u, (uint32_t)atoi(w[6].c_str()), A(7), gtable<C>(), &panic, &y);
        printf("%u %u", ok, panic);
    } else if (op == "split") {
        const SecpSplit sp = secp256k1_split_lambda(A(2));
        put(sp.k1); printf(" %d ", sp.neg1 ? 1 : 0); put(sp.k2); printf(" %d", sp.neg2 ? 1 : 0);
    } else printf("?");
    printf("\n");
}
int main() {
    char line[4096];
    while (fgets(line, sizeof line, stdin)) {
        std::vector<std::string> w;
        for (char *t = strtok(line, " \n"); t; t = strtok(nullptr, " \n")) w.push_back(t);
        if (w.size() < 2) continue;
        if (w[0] == "0") serve<0>(w); else serve<1>(w);
    }
    return 0;
}
