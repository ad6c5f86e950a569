// layoutbench.hip -- does the half-row layout of the witness table cost HBM efficiency? One lane = one instance; per "gate" every wave reads K
// pseudo-random rows and writes one, like arith_level_kernel without the arithmetic, in two layouts:
//   split:  row r = [lo half: B x 16 B][hi half: B x 16 B]   (fr_load: two 1 KiB segments a wave, B x 16 B apart)   <- the product's layout
//   joint:  row r = B x 32 B                                  (two 16-byte loads per lane, 32 B apart: one 2 KiB segment a wave)
// nontemporal stores / loads as in the product.   hipcc --offload-arch=gfx950 -O3 tools/layoutbench.hip -o tools/layoutbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <bool JOINT, int K>
__global__ void __launch_bounds__(256) gates(uint4 *__restrict__ W, uint64_t B, uint32_t n_rows, uint32_t out0, uint32_t seed) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t g = blockIdx.y;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t r = mix(seed + g * 16 + k) % out0;  // a row below the rows this launch writes
        const uint4 *lo = JOINT ? W + ((uint64_t)r * B + j) * 2 : W + (uint64_t)r * 2 * B + j;
        const uint4 *hi = JOINT ? lo + 1 : lo + B;
        uint4 x, y;
        x.x = __builtin_nontemporal_load(&lo->x); x.y = __builtin_nontemporal_load(&lo->y); x.z = __builtin_nontemporal_load(&lo->z); x.w = __builtin_nontemporal_load(&lo->w);
        y.x = __builtin_nontemporal_load(&hi->x); y.y = __builtin_nontemporal_load(&hi->y); y.z = __builtin_nontemporal_load(&hi->z); y.w = __builtin_nontemporal_load(&hi->w);
        a.x ^= x.x; a.y += x.y; a.z ^= x.z; a.w += x.w; b.x ^= y.x; b.y += y.y; b.z ^= y.z; b.w += y.w;
    }
    const uint32_t r = out0 + g;
    uint4 *lo = JOINT ? W + ((uint64_t)r * B + j) * 2 : W + (uint64_t)r * 2 * B + j;
    uint4 *hi = JOINT ? lo + 1 : lo + B;
    __builtin_nontemporal_store(a.x, &lo->x); __builtin_nontemporal_store(a.y, &lo->y); __builtin_nontemporal_store(a.z, &lo->z); __builtin_nontemporal_store(a.w, &lo->w);
    __builtin_nontemporal_store(b.x, &hi->x); __builtin_nontemporal_store(b.y, &hi->y); __builtin_nontemporal_store(b.z, &hi->z); __builtin_nontemporal_store(b.w, &hi->w);
}

template <bool JOINT>
static void run(uint4 *W, uint64_t B, uint32_t n_rows, const char *name) {
    const uint32_t gates_per_level = 270, levels = 20, first = n_rows - gates_per_level * levels;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        for (uint32_t L = 0; L < levels; L++)
            hipLaunchKernelGGL((gates<JOINT, 2>), dim3((unsigned)(B / 256), gates_per_level), dim3(256), 0, 0, W, B, n_rows, first + L * gates_per_level, 77u + L);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)levels * gates_per_level * B * 32.0 * 3.0;  // 2 rows read + 1 written
        printf("%s rep %d: %.3f ms for %u levels x %u gates x %llu instances: %.0f GB/s (2R + 1W)\n", name, rep, ms, levels, gates_per_level, (unsigned long long)B, bytes / ms / 1e6);
    }
}
int main() {
    const uint64_t B = 1 << 17;
    const uint32_t n_rows = 10016;  // 42 GB like the bench's tile
    uint4 *W;
    CHECK(hipMalloc(&W, (size_t)n_rows * 2 * B * sizeof(uint4)));
    CHECK(hipMemset(W, 1, (size_t)n_rows * 2 * B * sizeof(uint4)));
    run<false>(W, B, n_rows, "split (product)");
    run<true>(W, B, n_rows, "joint          ");
    run<false>(W, B, n_rows, "split (product)");
    run<true>(W, B, n_rows, "joint          ");
    return 0;
}
