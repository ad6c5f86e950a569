// probes.cpp -- measurement and self-test entry points of the C ABI (include/acvm_amd.h): the device self test of the field library, the
// back-to-back product probes behind the ALU rooflines, the streaming ceiling behind the HBM roofline, the component probes of the Grumpkin kernels.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include "batch_internal.hpp"

int acvm_selftest(uint32_t n, uint64_t seed) {
    uint32_t *d = nullptr, h = 0;
    HIPCHK(hipMalloc((void **)&d, 4));
    HIPCHK(hipMemset(d, 0, 4));
    launch_fr_selftest(nullptr, seed, n, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    hipFree(d);
    return (int)h;
}

// Peak of the ALU roofline: back-to-back Montgomery products on every SIMD, `waves_per_simd` chains interleaved.
// field: 0 = BN254-Fr in the 29-bit working form (fr29_mul), 1 / 2 = the base field of secp256k1 / secp256r1 (sp_mul, sp_sqr in turn)
static int product_rate(uint32_t field, uint32_t iters, uint32_t waves_per_simd, double *per_s, uint64_t *n_products) {
    if (!per_s || !iters || !waves_per_simd || field > 2) return set_err(ACVM_E_INVALID, "bad argument");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    const uint32_t blocks = (uint32_t)prop.multiProcessorCount * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
    uint32_t *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {  // the first run warms the clocks up
        hipEventRecord(e0, nullptr);
        if (field == 0) launch_modmul_rate(nullptr, d, blocks, iters);
        else launch_secp_rate(nullptr, field - 1, d, blocks, iters);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    HIPCHK(hipGetLastError());
    const double n = (double)blocks * 256.0 * iters * 2.0;
    *per_s = n / (best * 1e-3);
    if (n_products) *n_products = (uint64_t)n;
    return 0;
}
int acvm_debug_modmul_rate(uint32_t iters, uint32_t waves_per_simd, double *modmul_per_s, uint64_t *n_modmul) try {
    return product_rate(0, iters, waves_per_simd, modmul_per_s, n_modmul);
} catch (...) { return set_err(ACVM_E_DEVICE, "probe failed"); }
int acvm_debug_secp_rate(uint32_t curve, uint32_t iters, uint32_t waves_per_simd, double *products_per_s, uint64_t *n_products) try {
    if (curve > 1) return set_err(ACVM_E_INVALID, "curve: 0 = secp256k1, 1 = secp256r1");
    return product_rate(1 + curve, iters, waves_per_simd, products_per_s, n_products);
} catch (...) { return set_err(ACVM_E_DEVICE, "probe failed"); }

// The measured streaming ceiling beside the spec peak of the HBM roofline: two rows of `bytes` read and one written by a kernel with the
// gate kernel's access shape (kernels.hip stream_rate_kernel), best of four; bytes moved = 3 x bytes.
int acvm_debug_stream_rate(size_t bytes, double *gb_per_s) {
    if (!gb_per_s || bytes < (1u << 20)) return set_err(ACVM_E_INVALID, "bad argument");
    const uint64_t n = bytes / 16 / 256 * 256;
    uint4 *buf[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; k++)
        if (hipMalloc((void **)&buf[k], n * 16) != hipSuccess) {
            for (int q = 0; q < k; q++) hipFree(buf[q]);
            return set_err(ACVM_E_DEVICE, "hipMalloc failed");
        }
    for (int k = 0; k < 3; k++) HIPCHK(hipMemset(buf[k], k + 1, n * 16));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0, nullptr);
        launch_stream_rate(nullptr, buf[0], buf[1], buf[2], n);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    for (int k = 0; k < 3; k++) hipFree(buf[k]);
    HIPCHK(hipGetLastError());
    *gb_per_s = 3.0 * (double)(n * 16) / (best * 1e-3) / 1e9;
    return 0;
}

// Component probes of the Grumpkin kernels for the parity tests: what = 0 host table point (param = table << 24 | index),
// 1 device hash_single(in[0], parity = param), 2 device hash-ladder compress(in[0..n_in)), 3 device fixed_base_mul(table
// base param, integer in[0]), 4 device table point. in: n_in x 32 bytes big-endian; out: 64 bytes (x || y) big-endian.
int acvm_debug_secp(uint32_t curve, uint32_t what, const uint8_t *in_be32, uint32_t n_items, uint8_t *out_be32) try {
    static const uint32_t WIN[12] = {2, 1, 2, 2, 1, 1, 3, 5, 1, 2, 2, 1}, WOUT[12] = {1, 1, 1, 1, 1, 1, 3, 3, 1, 1, 1, 1};
    if (curve > 1u || what > 11u || !in_be32 || !out_be32) return set_err(ACVM_E_INVALID, "bad argument");
    if (!n_items) return 0;
    const uint32_t wi = WIN[what], wo = WOUT[what];
    std::vector<uint32_t> in((size_t)n_items * wi * 8, 0), out((size_t)n_items * wo * 8, 0);
    for (size_t i = 0; i < (size_t)n_items * wi; i++)
        for (int k = 0; k < 32; k++) in[8 * i + k / 4] |= (uint32_t)in_be32[32 * i + 31 - k] << (8 * (k % 4));
    uint32_t *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_in, in.size() * 4));
    HIPCHK(hipMalloc((void **)&d_out, out.size() * 4));
    HIPCHK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_out, 0, out.size() * 4));
    launch_secp_probe(nullptr, curve, what, d_in, n_items, wi, wo, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    hipFree(d_in);
    hipFree(d_out);
    for (size_t i = 0; i < (size_t)n_items * wo; i++)
        for (int k = 0; k < 32; k++) out_be32[32 * i + 31 - k] = (uint8_t)(out[8 * i + k / 4] >> (8 * (k % 4)));
    return 0;
} ABI_CATCH

int acvm_debug_grumpkin(uint32_t what, uint32_t param, const uint8_t *in_be32, uint32_t n_in, uint8_t *out_be64) try {
    if (!out_be64) return set_err(ACVM_E_INVALID, "null argument");
    if (what == 0) return grumpkin_host_point(param >> 24, param & 0xffffffu, out_be64) ? 0 : set_err(ACVM_E_INVALID, "bad table index");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    // the probe holds the device's tables while it runs, from BEFORE it asks for their addresses: with tables_keep = 0 another handle's
    // destruction between the two would free what was just handed out (acvm_device_release_tables refuses meanwhile)
    struct Hold {
        int d;
        explicit Hold(int dev_) : d(dev_) { device_tables_retain(d); }
        ~Hold() { device_tables_unref(d); }
    } hold(dev);
    GrumpkinTables tabs;
    if (!grumpkin_tables(&tabs)) return set_err(ACVM_E_DEVICE, "could not build the Grumpkin tables on the device");
    const GrumpkinTables *t = &tabs;
    std::vector<uint32_t> in(8 * (n_in ? n_in : 1), 0), out(16, 0);
    for (uint32_t i = 0; i < n_in; i++)
        for (int k = 0; k < 32; k++) in[8 * i + k / 4] |= (uint32_t)in_be32[32 * i + 31 - k] << (8 * (k % 4));
    uint32_t *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_in, in.size() * 4));
    HIPCHK(hipMalloc((void **)&d_out, 64));
    HIPCHK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_out, 0, 64));
    launch_grumpkin_probe(nullptr, *t, what, param, d_in, n_in, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out.data(), d_out, 64, hipMemcpyDeviceToHost));
    hipFree(d_in);
    hipFree(d_out);
    for (int c = 0; c < 2; c++)
        for (int k = 0; k < 32; k++) out_be64[32 * c + 31 - k] = (uint8_t)(out[8 * c + k / 4] >> (8 * (k % 4)));
    return 0;
} ABI_CATCH


