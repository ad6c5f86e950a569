// tools/copybench.hip -- what does the HBM / fabric path of an MI355X give to a streaming kernel? Variants of a 16 B/lane copy
// (grid size, loads in flight per lane, non-temporal hints), a read-only and a write-only stream, and a 2-read-1-write stream
// shaped like arith_level_kernel (two operand rows far apart, one output row).   hipcc --offload-arch=gfx950 -O3 copybench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t V4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_u(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (; i + (U - 1) * 256 < n; i += stride) {
        uint4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) { const V4 t = __builtin_nontemporal_load((const V4 *)&in[i + k * 256]); v[k] = make_uint4(t.x, t.y, t.z, t.w); }
            else v[k] = in[i + k * 256];
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) { V4 t = {v[k].x, v[k].y, v[k].z, v[k].w}; __builtin_nontemporal_store(t, (V4 *)&out[i + k * 256]); }
            else out[i + k * 256] = v[k];
        }
    }
}
// one block = one contiguous tile, no grid-stride loop (the shape of a level kernel: grid covers the data exactly)
template <int U>
__global__ void __launch_bounds__(256) copy_tile(const uint4 *__restrict__ in, uint4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = in[i + k * 256];
#pragma unroll
    for (int k = 0; k < U; k++) out[i + k * 256] = v[k];
}
__global__ void __launch_bounds__(256) read_only(const uint4 *__restrict__ in, uint32_t *__restrict__ sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) write_only(uint4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// out[r] = a[r] ^ b[r'] for two source rows: 2 reads + 1 write per 16 B, like a gate with two operands
__global__ void __launch_bounds__(256) two_in_one_out(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint4 x = a[i], y = b[i];
    out[i] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
}

template <class F>
static float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    uint4 *a, *b, *c;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes)); CHECK(hipMemset(c, 3, bytes));
    auto rep = [&](const char *name, float ms, double moved) { printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, moved / ms / 1e6); };
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64]; snprintf(nm, 64, "copy grid-stride U=1, %d blocks", g);
        rep(nm, time_ms([&] { copy_u<1, false><<<g, 256>>>(a, b, n); }), 2.0 * bytes);
    }
    rep("copy grid-stride U=2, 2048 blocks", time_ms([&] { copy_u<2, false><<<2048, 256>>>(a, b, n); }), 2.0 * bytes);
    rep("copy grid-stride U=4, 2048 blocks", time_ms([&] { copy_u<4, false><<<2048, 256>>>(a, b, n); }), 2.0 * bytes);
    rep("copy grid-stride U=8, 1024 blocks", time_ms([&] { copy_u<8, false><<<1024, 256>>>(a, b, n); }), 2.0 * bytes);
    rep("copy grid-stride U=4 non-temporal, 2048", time_ms([&] { copy_u<4, true><<<2048, 256>>>(a, b, n); }), 2.0 * bytes);
    rep("copy grid-stride U=1 non-temporal, 2048", time_ms([&] { copy_u<1, true><<<2048, 256>>>(a, b, n); }), 2.0 * bytes);
    rep("copy one tile per block U=1", time_ms([&] { copy_tile<1><<<(unsigned)(n / 256), 256>>>(a, b); }), 2.0 * bytes);
    rep("copy one tile per block U=4", time_ms([&] { copy_tile<4><<<(unsigned)(n / 1024), 256>>>(a, b); }), 2.0 * bytes);
    rep("copy one tile per block U=8", time_ms([&] { copy_tile<8><<<(unsigned)(n / 2048), 256>>>(a, b); }), 2.0 * bytes);
    uint32_t *sink; CHECK(hipMalloc(&sink, 4));
    rep("read only, 4096 blocks", time_ms([&] { read_only<<<4096, 256>>>(a, sink, n); }), 1.0 * bytes);
    rep("write only, 4096 blocks", time_ms([&] { write_only<<<4096, 256>>>(b, n); }), 1.0 * bytes);
    rep("2 reads + 1 write, one tile per block", time_ms([&] { two_in_one_out<<<(unsigned)(n / 256), 256>>>(a, c, b); }), 3.0 * bytes);
    rep("hipMemcpyDtoD", time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }), 2.0 * bytes);
    return 0;
}
