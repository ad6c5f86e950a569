// tools/narrow_probe.hip -- reproduction of the hipcc (ROCm 7.2, gfx950) miscompilation that acvm_amd/csrc/secp_device.hpp works around with s29_opaque:
// a^4 = sqr(sqr(a)) in the 29-bit working form of secp256k1, on the device WITHOUT the barrier, with the limbs of the first square laundered selectively,
// against the same header run on the host.   hipcc --offload-arch=gfx950 -O3 tools/narrow_probe.hip -o tools/narrow_probe && tools/narrow_probe
#define S29_PROBE_NO_OPAQUE
#include "../acvm_amd/csrc/secp_device.hpp"
#include <cstdio>
#include <vector>
using namespace acvm;

template <unsigned MASK>  // limb i of the first square passes an empty asm iff bit i is set
__global__ void a4_kernel(const uint32_t *in, uint32_t *out, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Fr a;
    for (int i = 0; i < 8; i++) a.v[i] = in[t * 8 + i];
    S29 x = s29_sqr<0>(fr29_from(a));
#pragma unroll
    for (int i = 0; i < 9; i++)
        if (MASK >> i & 1u) asm volatile("" : "+v"(x.v[i]));
    const S29 y = s29_canon<0>(s29_sqr<0>(x));
    for (int i = 0; i < 9; i++) out[t * 9 + i] = y.v[i];
}
template <unsigned MASK>
static int run(const char *what, const uint32_t *d_in, uint32_t *d_out, const std::vector<uint32_t> &in, uint32_t n) {
    hipLaunchKernelGGL(a4_kernel<MASK>, dim3((n + 63) / 64), dim3(64), 0, 0, d_in, d_out, n);
    std::vector<uint32_t> out(n * 9);
    if (hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int bad = 0, first = -1;
    for (uint32_t t = 0; t < n; t++) {
        Fr a;
        for (int i = 0; i < 8; i++) a.v[i] = in[t * 8 + i];
        const S29 want = s29_canon<0>(s29_sqr<0>(s29_sqr<0>(fr29_from(a))));
        bool same = true;
        for (int i = 0; i < 9; i++) same = same && want.v[i] == out[t * 9 + i];
        if (!same) { bad++; if (first < 0) first = (int)t; }
    }
    printf("%-44s wrong results: %d of %u", what, bad, n);
    if (first >= 0) {
        Fr a;
        for (int i = 0; i < 8; i++) a.v[i] = in[first * 8 + i];
        const S29 want = s29_canon<0>(s29_sqr<0>(s29_sqr<0>(fr29_from(a))));
        printf("   first: item %d, limbs that differ:", first);
        for (int i = 0; i < 9; i++)
            if (want.v[i] != out[first * 9 + i]) printf(" %d (device %08x host %08x)", i, out[first * 9 + i], want.v[i]);
    }
    printf("\n");
    return bad;
}
int main() {
    const uint32_t n = 4096;
    std::vector<uint32_t> in(n * 8);
    uint64_t s = 0x1234567;
    for (auto &w : in) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; w = (uint32_t)(s >> 32); }
    for (uint32_t t = 0; t < n; t++) in[t * 8 + 7] >>= 1;  // below p
    const uint32_t pm1[8] = {0xfffffc2eu, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    for (int i = 0; i < 8; i++) in[i] = pm1[i];  // item 0: p - 1
    uint32_t *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc((void **)&d_in, in.size() * 4) != hipSuccess || hipMalloc((void **)&d_out, n * 9 * 4) != hipSuccess) { printf("no device\n"); return 2; }
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    run<0x000u>("no limb laundered", d_in, d_out, in, n);
    run<0x1ffu>("every limb laundered (what the library does)", d_in, d_out, in, n);
    run<0x100u>("limb 8 only (24 bits known)", d_in, d_out, in, n);
    run<0x008u>("limb 3 only (29 bits + a carry)", d_in, d_out, in, n);
    run<0x0f7u>("limbs 0..2, 4..7 (29 bits known)", d_in, d_out, in, n);
    run<0x108u>("limbs 3 and 8", d_in, d_out, in, n);
    return 0;
}
