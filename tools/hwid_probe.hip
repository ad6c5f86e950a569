// hwid_probe.hip -- where do the waves of 4-wave workgroups land? Prints, for a grid of 1024 x 256 threads, how the hardware places wave q
// of a block (SIMD id, wave slot) and how many wave-0s share a (CU, SIMD).   hipcc --offload-arch=gfx950 -O2 hwid_probe.hip -o hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t spin) {
    const uint32_t q = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    uint32_t x = hw + lane;
    for (uint32_t i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;  // keep the blocks resident together
    if (lane == 0) { out[(blockIdx.x * 4 + q) * 2] = hw; out[(blockIdx.x * 4 + q) * 2 + 1] = (xcc & 0xf) | (x & 0x80000000u); }
}
int main() {
    const uint32_t nb = 1024;
    uint32_t *d;
    hipMalloc(&d, nb * 4 * 2 * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 20000u);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (uint32_t b = 0; b < 8; b++) {
        printf("block %u:", b);
        for (uint32_t q = 0; q < 4; q++) { uint32_t hw = h[(b * 4 + q) * 2]; printf("  q%u hw=%08x simd=%u slot=%u cu=%u se=%u xcc=%u", q, hw, (hw >> 4) & 3, hw & 15, (hw >> 8) & 15, (hw >> 13) & 7, h[(b * 4 + q) * 2 + 1] & 0xf); }
        printf("\n");
    }
    std::map<uint64_t, std::vector<uint32_t>> per_cu;  // (xcc, se, sh, cu) -> blocks
    std::map<uint64_t, int> w0_simd;
    for (uint32_t b = 0; b < nb; b++) {
        uint32_t hw = h[(b * 4) * 2], xcc = h[(b * 4) * 2 + 1] & 0xf;
        uint64_t cu = ((uint64_t)xcc << 16) | ((hw >> 8) & 0xff);
        per_cu[cu].push_back(b);
        w0_simd[(cu << 4) | ((hw >> 4) & 3)]++;
    }
    printf("distinct CUs: %zu\n", per_cu.size());
    int hist[16] = {0};
    for (auto &kv : w0_simd) hist[kv.second < 15 ? kv.second : 15]++;
    printf("wave-0s per (CU, SIMD) histogram:");
    for (int i = 1; i < 16; i++) if (hist[i]) printf(" %d:%d", i, hist[i]);
    printf("\n");
    int shown = 0;
    for (auto &kv : per_cu) {
        if (shown++ >= 4) break;
        printf("cu %llx blocks:", (unsigned long long)kv.first);
        for (uint32_t b : kv.second) { printf(" %u[", b); for (uint32_t q = 0; q < 4; q++) printf("%u/%u ", (h[(b * 4 + q) * 2] >> 4) & 3, h[(b * 4 + q) * 2] & 15); printf("]"); }
        printf("\n");
    }
    return 0;
}
