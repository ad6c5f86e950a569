#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
// mad_hazard_probe.hip -- do dependent v_mad_u64_u32 need a wait state between them on gfx950? hipcc puts an s_nop behind every asm statement whose result the
// next instruction reads, and never schedules its own dependent multiply-adds... except that it does (kernels.hip's ISA has pairs back to back). Here: four
// dependent multiply-adds and a shift inside ONE asm block, 64 times over, against the same chain in C and against the block with s_nop 0 between the
// instructions: 0 of 262 144 lanes differ -- the hardware interlocks, the s_nop is the compiler's caution about asm (fr_blocks.inc relies on it).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mad_hazard_probe tools/mad_hazard_probe.hip && tools/mad_hazard_probe
__global__ void k(const uint32_t *in, uint64_t *out_asm, uint64_t *out_c, uint64_t *out_nop) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t a = in[4 * t], b = in[4 * t + 1], c = in[4 * t + 2], d = in[4 * t + 3];
    uint64_t x = a, y = a, z = a;
    for (int i = 0; i < 64; i++) {
        uint64_t cy;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %5, %2, %0\n\t"
                     "v_lshrrev_b64 %0, 3, %0"
                     : "+v"(x), "=&s"(cy) : "v"(a), "v"(b), "v"(c), "v"(d));
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0\n\ts_nop 0\n\tv_mad_u64_u32 %0, %1, %3, %4, %0\n\ts_nop 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\ts_nop 0\n\tv_mad_u64_u32 %0, %1, %5, %2, %0\n\ts_nop 0\n\t"
                     "v_lshrrev_b64 %0, 3, %0"
                     : "+v"(z), "=&s"(cy) : "v"(a), "v"(b), "v"(c), "v"(d));
        y = (uint64_t)a * b + y;
        y = (uint64_t)b * c + y;
        y = (uint64_t)c * d + y;
        y = (uint64_t)d * a + y;
        y >>= 3;
    }
    out_asm[t] = x; out_c[t] = y; out_nop[t] = z;
}
int main() {
    const int n = 256 * 1024;
    std::vector<uint32_t> h(4 * n);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
    uint32_t *in; uint64_t *o1, *o2, *o3;
    hipMalloc(&in, 16 * n); hipMalloc(&o1, 8 * n); hipMalloc(&o2, 8 * n); hipMalloc(&o3, 8 * n);
    hipMemcpy(in, h.data(), 16 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(in, o1, o2, o3);
    std::vector<uint64_t> a(n), b(n), c(n);
    hipMemcpy(a.data(), o1, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(c.data(), o3, 8 * n, hipMemcpyDeviceToHost);
    size_t bad = 0, badn = 0;
    for (int i = 0; i < n; i++) { bad += a[i] != b[i]; badn += c[i] != b[i]; }
    printf("dependent mads back to back, no wait state: %zu of %d lanes differ from the C chain; with s_nop 0: %zu differ\n", bad, n, badn);
    return 0;
}
