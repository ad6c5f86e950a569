// batch.cpp -- the C ABI (include/acvm_amd.h) and the batch driver: one handle = one circuit plan, one
// device-resident witness table W[slot][half][instance], one HIP stream. Mirrors the call shape of
// acvm::pwg::ACVM (acvm/src/pwg/mod.rs:145-304) for B instances at once.
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
#include "batch.hpp"
#include "display.hpp"

static thread_local std::string g_last_error;
int set_err(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

static void format_message(acvm_batch *b, uint32_t j, const SlowResult &sr, acvm_result_t &r);
static void fill_result(acvm_batch *b, uint32_t j, acvm_result_t &r);

template <class T>
static int upload(T **dst, const std::vector<T> &src) {
    size_t bytes = (src.size() ? src.size() : 1) * sizeof(T);
    HIPCHK(hipMalloc((void **)dst, bytes));
    if (!src.empty()) HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// staging arena: `bytes` of device memory valid until the next stage_reserve of this batch (256-byte aligned carving by the caller)
static int stage_reserve(acvm_batch *b, size_t bytes) {
    if (bytes <= b->stage_cap) return 0;
    if (b->d_stage) { hipFree(b->d_stage); b->d_stage = nullptr; b->stage_cap = 0; }
    const size_t cap = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
    HIPCHK(hipMalloc((void **)&b->d_stage, cap));
    b->stage_cap = cap;
    return 0;
}
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// forget every resolved result (a new ACVM: set_initial_witness / reset)
static void clear_fc_store(acvm_batch *b) {
    for (auto &sl : b->fc_slots)
        if (!sl.inst.empty()) { sl.inst.clear(); sl.dirty = true; }
}


extern "C" {

const char *acvm_last_error(void) { return g_last_error.c_str(); }
int acvm_abi_version(void) { return ACVM_AMD_ABI_VERSION; }

int acvm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int acvm_set_device(int device) {
    HIPCHK(hipSetDevice(device));
    return 0;
}
int acvm_device_synchronize(void) {
    HIPCHK(hipDeviceSynchronize());
    return 0;
}
int acvm_device_arch(char *out, size_t out_len) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    snprintf(out, out_len, "%s", prop.gcnArchName);
    return 0;
}

void *acvm_device_malloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { set_err(ACVM_E_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
int acvm_device_free(void *p) {
    if (p) HIPCHK(hipFree(p));
    return 0;
}
int acvm_device_upload(void *dst_device, const void *src_host, size_t bytes) {
    if (bytes && (!dst_device || !src_host)) return set_err(ACVM_E_INVALID, "null argument");
    if (bytes) HIPCHK(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return 0;
}

long long acvm_device_release_tables(int device) {
    if (device < 0 || device >= acvm_device_count()) return set_err(ACVM_E_INVALID, "device index out of range");
    size_t freed = 0;
    const int rc = device_tables_free(device, &freed);
    if (rc == 1) return set_err(ACVM_E_STATE, "batch handles of device " + std::to_string(device) + " still use its lookup tables: free them first");
    if (rc < 0) return set_err(ACVM_E_DEVICE, "could not select device " + std::to_string(device));
    return (long long)freed;
}

int acvm_tuning_set(const char *key, long long value) {
    if (!tuning_set(key, (int64_t)value)) return set_err(ACVM_E_INVALID, std::string("unknown tuning key ") + (key ? key : "(null)"));
    return 0;
}
int acvm_tuning_get(const char *key, long long *value) {
    int64_t v = 0;
    if (!value || !tuning_get(key, &v)) return set_err(ACVM_E_INVALID, std::string("unknown tuning key ") + (key ? key : "(null)"));
    *value = (long long)v;
    return 0;
}
const char *acvm_tuning_key(unsigned index) { return tuning_key(index); }

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
int acvm_debug_grumpkin(uint32_t what, uint32_t param, const uint8_t *in_be32, uint32_t n_in, uint8_t *out_be64) try {
    if (!out_be64) return set_err(ACVM_E_INVALID, "null argument");
    if (what == 0) return grumpkin_host_point(param >> 24, param & 0xffffffu, out_be64) ? 0 : set_err(ACVM_E_INVALID, "bad table index");
    GrumpkinTables tabs;
    if (!grumpkin_tables(&tabs)) return set_err(ACVM_E_DEVICE, "could not build the Grumpkin tables on the device");
    const GrumpkinTables *t = &tabs;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    struct Hold {  // the probe holds the device's tables while it runs (acvm_device_release_tables refuses meanwhile)
        int d;
        explicit Hold(int dev_) : d(dev_) { device_tables_retain(d); }
        ~Hold() { device_tables_unref(d); }
    } hold(dev);
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

acvm_circuit_t *acvm_circuit_from_bytes(const uint8_t *bytes, size_t len) try {
    if (!bytes) { set_err(ACVM_E_INVALID, "null circuit bytes"); return nullptr; }
    if (!frh::self_check()) { set_err(ACVM_E_INVALID, "field constants self-check failed"); return nullptr; }
    std::string err;
    auto c = circuit_from_bytes(bytes, len, err);
    if (!c) { set_err(ACVM_E_MALFORMED, err); return nullptr; }
    auto *h = new acvm_circuit;
    h->c = std::move(c);
    return h;
} ABI_CATCH_PTR
void acvm_circuit_free(acvm_circuit_t *c) { delete c; }
uint32_t acvm_circuit_num_opcodes(const acvm_circuit_t *c) { return c ? (uint32_t)c->c->opcodes.size() : 0; }
uint32_t acvm_circuit_num_witnesses(const acvm_circuit_t *c) { return c ? c->c->max_witness + 1 : 0; }
int acvm_circuit_opcode_kinds(const acvm_circuit_t *c, uint32_t first, uint32_t n, uint32_t *kinds) {
    if (!c || (n && !kinds)) return set_err(ACVM_E_INVALID, "null argument");
    const std::vector<Opcode> &ops = c->c->opcodes;
    if ((uint64_t)first + n > ops.size()) return set_err(ACVM_E_INVALID, "opcode range out of bounds");
    for (uint32_t i = 0; i < n; i++) {
        const Opcode &o = ops[first + i];
        kinds[2 * i] = o.kind;
        kinds[2 * i + 1] = o.kind == OP_BLACKBOX ? o.bb->func : o.kind == OP_DIRECTIVE ? o.dir->kind : o.kind == OP_BRILLIG ? (uint32_t)o.brillig->bytecode.size()
                           : o.kind == OP_MEMORY_OP || o.kind == OP_MEMORY_INIT ? o.block_id : 0u;
    }
    return 0;
}

static void plan_stats(const Plan &p, acvm_stats_t *out) {
    memset(out, 0, sizeof *out);
    out->n_opcodes = p.n_opcodes;
    out->n_witnesses = p.n_witnesses;
    out->n_levels = p.n_levels;
    out->n_fast_gates = p.n_fast_gates;
    out->n_dyn_gates = p.n_dyn_gates;
    out->max_level_width = p.max_level_width;
    out->algorithmic_bytes_per_instance = p.algorithmic_bytes;
    out->arith_algorithmic_bytes_per_instance = p.arith_algorithmic_bytes;
    out->dyn_algorithmic_bytes_per_instance = p.dyn_algorithmic_bytes;
    out->plan_ms = p.plan_ms;
    out->n_other_records = p.n_other_records;
    out->truncated_at = p.truncated_at;
    out->n_gate_pairs = p.n_gate_pairs;
    out->n_inverse_slots = p.n_inverse_slots;
    out->n_scaled_witnesses = (uint32_t)p.scaled_ids.size();
    out->n_table_rows = p.slot_of.empty() ? p.n_witnesses : p.n_slots;
    out->n_digest_segments = p.n_digest_segments;
    out->n_brillig_inlined = p.n_brillig_inlined;
    out->n_hash_chained = p.n_hash_chained;
    for (uint32_t L = 0; L + 1 < p.level_start.size(); L++) out->n_arith_launches += (p.level_start[L + 1] - p.level_start[L] + 65534) / 65535;
    for (int k = 0; k < 4; k++) out->class_algorithmic_bytes_per_instance[k] = p.cls_algorithmic_bytes[k];
    out->class_algorithmic_bytes_per_instance[CLS_GRUMPKIN] += p.cls_algorithmic_bytes[CLS_PEDERSEN] + p.cls_algorithmic_bytes[CLS_ECDSA] +
                                                               p.cls_algorithmic_bytes[CLS_HOSTBB];
}

// Host-only: levelise the circuit against a set of initial witness ids without touching a device (plan statistics, and
// whether the circuit holds an opcode no kernel implements). Returns 0, or ACVM_E_UNSUPPORTED with the reason as the error text.
int acvm_circuit_plan_stats(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, acvm_stats_t *out) {
    return acvm_circuit_plan_stats_ex(c, initial_ids, n_initial, 0, nullptr, 0, out);
}
int acvm_circuit_plan_stats_ex(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids,
                               uint32_t n_keep, acvm_stats_t *out) try {
    if (!c || !out || (n_initial && !initial_ids) || (n_keep && !keep_ids)) return set_err(ACVM_E_INVALID, "null argument");
    PlanOpts opts;
    opts.fold_digest = (flags & (ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) != 0;
    opts.reuse_slots = (flags & ACVM_BATCH_REUSE_SLOTS) != 0;
    opts.keep.assign(keep_ids, keep_ids + n_keep);
    Plan p = build_plan(*c->c, initial_ids, n_initial, opts);
    plan_stats(p, out);
    if (!p.unsupported.empty()) return set_err(ACVM_E_UNSUPPORTED, p.unsupported);
    return 0;
} ABI_CATCH

// The two fixed field elements of the witness-map digest (include/acvm_amd.h acvm_batch_digest): Blake2s-256 of the ASCII strings
// "acvm_amd witness map digest: g" / "... h", read as big-endian integers and reduced modulo p.
static const uint8_t DIGEST_G[32] = {0x23, 0x35, 0x53, 0x18, 0xdb, 0xff, 0xab, 0x2f, 0xb7, 0x72, 0x11, 0x7c, 0x57, 0x5c, 0x61, 0xb1,
                                     0x79, 0xf8, 0xc9, 0x83, 0x3c, 0x83, 0xba, 0x65, 0x59, 0x7e, 0x17, 0x3c, 0x35, 0xc4, 0xbb, 0xe3};
static const uint8_t DIGEST_H[32] = {0x28, 0x25, 0x78, 0x33, 0xe7, 0x23, 0x7f, 0xbd, 0x29, 0x7c, 0x55, 0x74, 0x6b, 0xe0, 0xa3, 0xa9,
                                     0x8a, 0x2a, 0x89, 0x8d, 0x8a, 0xb1, 0x0b, 0xe0, 0x05, 0xaa, 0x2f, 0xdf, 0x9c, 0x60, 0x11, 0xa4};
// device tables of the digest: g^(w+1), g^(w+1) / scale_w for the scaled witnesses, h^(w+1), and the h-sum of the planner's assigned set
static int ensure_digest_tables(acvm_batch *b) {
    if (b->d_fp_g) return 0;
    const Plan &p = b->plan;
    const uint32_t nw = p.n_witnesses;
    const FrH g = frh::from_be_bytes32_reduce(DIGEST_G, 32), h = frh::from_be_bytes32_reduce(DIGEST_H, 32);
    std::vector<uint32_t> tg((size_t)std::max<uint32_t>(nw, 1) * 8), th((size_t)std::max<uint32_t>(nw, 1) * 8), tgs(std::max<size_t>(p.unscale.size(), 1) * 8), hgen(8);
    FrH gp = g, hp = h, hsum = frh::zero();
    auto put = [](std::vector<uint32_t> &v, size_t i, const FrH &x) {
        const FrH d = frh::to_device_form(x);
        memcpy(&v[8 * i], d.l, 32);
    };
    for (uint32_t w = 0; w < nw; w++) {
        put(tg, w, gp);
        put(th, w, hp);
        if (p.unscale_index[w] != 0xFFFFFFFFu) put(tgs, p.unscale_index[w], frh::mul(gp, p.unscale[p.unscale_index[w]]));
        if (p.producer[w] != 0xFFFFFFFFu) hsum = frh::add(hsum, hp);
        gp = frh::mul(gp, g);
        hp = frh::mul(hp, h);
    }
    put(hgen, 0, hsum);
    if (int rc = upload(&b->d_fp_g, tg)) return rc;
    if (int rc = upload(&b->d_fp_h, th)) return rc;
    if (int rc = upload(&b->d_fp_gs, tgs)) return rc;
    if (int rc = upload(&b->d_fp_hgen, hgen)) return rc;
    b->fp = DigestTables{b->d_fp_g, b->d_fp_gs, b->d_fp_h, b->d_fp_hgen};
    return 0;
}

static int batch_init(acvm_batch *b) {
    HIPCHK(hipGetDevice(&b->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, b->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(ACVM_E_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_dyn, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_heavy, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_heavy2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_heavy3, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->stream_digest, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&b->ev_start));
    HIPCHK(hipEventCreate(&b->ev_end));
    const Plan &p = b->plan;
    size_t w_bytes = (size_t)(b->reuse() ? p.n_slots : p.n_witnesses) * 2 * b->Bp * sizeof(uint4);
    HIPCHK(hipMalloc((void **)&b->d_W, w_bytes ? w_bytes : 16));
    if (int rc = upload(&b->d_gate_stream, p.gate_stream)) return rc;
    if (int rc = upload(&b->d_gate_offset, p.gate_offset)) return rc;
    std::vector<uint32_t> consts(p.constants.size() * 8);
    for (size_t i = 0; i < p.constants.size(); i++) {
        const FrH d = frh::to_device_form(p.constants[i]);
        memcpy(&consts[8 * i], d.l, 32);
    }
    if (int rc = upload(&b->d_consts, consts)) return rc;
    if (!p.scaled_ids.empty()) {
        std::vector<uint32_t> uc(p.unscale.size() * 8);
        for (size_t i = 0; i < p.unscale.size(); i++) {
            const FrH d = frh::to_device_form(p.unscale[i]);
            memcpy(&uc[8 * i], d.l, 32);
        }
        if (int rc = upload(&b->d_unscale_consts, uc)) return rc;
        for (size_t i = 0; i < p.unscale.size(); i++) {
            uint64_t c[4];
            frh::to_canonical(p.unscale[i], c);
            memcpy(&uc[8 * i], c, 32);
        }
        if (int rc = upload(&b->d_unscale_plain, uc)) return rc;
        if (int rc = upload(&b->d_unscale_index, p.unscale_index)) return rc;
        if (int rc = upload(&b->d_scaled_ids, p.scaled_ids)) return rc;
    }
    if (int rc = upload(&b->d_prog, p.prog)) return rc;
    if (int rc = upload(&b->d_prog_offset, p.prog_offset)) return rc;
    if (int rc = upload(&b->d_bytecode, p.bytecode)) return rc;
    if (int rc = upload(&b->d_init_ids, p.initial_ids)) return rc;
    {   // per-instance memory blocks (MemoryInit / MemoryOp), laid out like W
        size_t bytes = (size_t)p.mem_cells * 2 * b->Bp * sizeof(uint4);
        HIPCHK(hipMalloc((void **)&b->d_Mem, bytes ? bytes : 16));
    }
    // non-arithmetic record classes: per level, launch chunks whose per-instance scratch fits the class's scratch buffer
    const uint64_t scratch_cap_words = std::max<uint64_t>(1, (1ull << 30) / (std::max<uint64_t>(b->Bp, 64) * 4));  // 1 GiB per class (an empty batch is allowed)
    for (int k = 0; k < (int)N_CLS; k++) {
        const size_t n_levels = p.n_levels;
        b->cls_chunks[k].assign(n_levels, {});
        std::vector<uint32_t> scratch_off(2 * p.cls_offset[k].size(), 0);  // per record: (offset, words) of its per-lane scratch, in u32 words
        uint64_t need = 0;
        for (size_t L = 0; L < n_levels; L++) {
            uint32_t lo = p.cls_level_start[k][L], hi = p.cls_level_start[k][L + 1];
            if (k == CLS_HASH) {  // byte-message hashes first: their own kernel, no scratch (the records of a level are independent)
                auto is_coop = [&](uint32_t r) { return (b->plan.prog[b->plan.cls_offset[k][r] + 2] & PLAN_HASH_COOP_FLAG) != 0; };
                // The LDS message of a launch is sized by its longest record (a 1 024-byte message takes the whole 64 KiB a workgroup may have):
                // short messages (<= 256 bytes: 16 KiB per 64 instances) and long ones get launches of their own, so that one long message
                // somewhere in the circuit does not cost every 64-byte SHA record its occupancy.
                auto words_of = [&](uint32_t r) {  // message words of the record, or of the longest member of the chain it heads (plan.cpp hash chains)
                    uint32_t words = 0;
                    for (size_t at = b->plan.cls_offset[k][r];;) {
                        const std::vector<uint32_t> &pg = b->plan.prog;
                        const uint32_t n_in = pg[at + 3];
                        words = std::max(words, (n_in + 3u) / 4u);
                        if (!(pg[at + 2] & PLAN_HASH_CHAIN_FLAG)) break;
                        at = pg[pg[at + 6 + 2 * (size_t)n_in + 64 + ((pg[at + 2] & PLAN_HASH_RANGE_FLAG) ? 2 * (size_t)n_in : 0)]];
                    }
                    return words;
                };
                std::vector<std::pair<uint32_t, uint32_t>> recs;  // (offset, scratch)
                uint32_t n_pass[3] = {0, 0, 0}, words_pass[2] = {0, 0};
                for (int pass = 0; pass < 3; pass++)
                    for (uint32_t r = lo; r < hi; r++) {
                        const int cls = !is_coop(r) ? 2 : (words_of(r) <= 64u ? 0 : 1);
                        if (cls != pass) continue;
                        recs.push_back({b->plan.cls_offset[k][r], b->plan.cls_scratch[k][r]});
                        n_pass[pass]++;
                        if (pass < 2) words_pass[pass] = std::max(words_pass[pass], words_of(r));
                    }
                for (uint32_t r = lo; r < hi; r++) { b->plan.cls_offset[k][r] = recs[r - lo].first; b->plan.cls_scratch[k][r] = recs[r - lo].second; }
                for (int pass = 0; pass < 2; pass++) {
                    if (n_pass[pass]) b->cls_chunks[k][L].push_back({lo, n_pass[pass], true, words_pass[pass]});
                    lo += n_pass[pass];
                }
            }
            if (k == CLS_GRUMPKIN && hi - lo > 1) {
                // the longest records first (SchnorrVerify ~1.7 ms of one wave per SIMD, FixedBaseScalarMul 0.3): workgroups are placed in grid order,
                // and at ~240 registers a SIMD holds two of these waves -- a long wave that arrives last waits for a slot behind short ones elsewhere
                std::vector<std::pair<uint32_t, uint32_t>> recs;
                for (uint32_t r = lo; r < hi; r++) recs.push_back({b->plan.cls_offset[k][r], b->plan.cls_scratch[k][r]});
                std::stable_sort(recs.begin(), recs.end(), [&](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) {
                    auto rank = [&](uint32_t off) { const uint32_t kind = b->plan.prog[off]; return kind == PK_SCHNORR ? 0 : kind == PK_PEDERSEN ? 1 : 2; };
                    return rank(x.first) < rank(y.first);
                });
                for (uint32_t r = lo; r < hi; r++) { b->plan.cls_offset[k][r] = recs[r - lo].first; b->plan.cls_scratch[k][r] = recs[r - lo].second; }
            }
            if (k == CLS_LIGHT) {  // straight-line Brillig records last: they have a kernel of their own (kernels_ops.hip LightSlOp)
                auto is_sl = [&](uint32_t r) { return b->plan.prog[b->plan.cls_offset[k][r]] == PK_BRILLIG_SL; };
                std::vector<uint32_t> offs;
                uint32_t n_sl = 0;
                for (int pass = 0; pass < 2; pass++)
                    for (uint32_t r = lo; r < hi; r++)
                        if (is_sl(r) == (pass == 1)) { offs.push_back(b->plan.cls_offset[k][r]); n_sl += pass; }
                for (uint32_t r = lo; r < hi; r++) b->plan.cls_offset[k][r] = offs[r - lo];
                if (n_sl) {
                    if (hi - n_sl > lo) b->cls_chunks[k][L].push_back({lo, hi - n_sl - lo});
                    b->cls_chunks[k][L].push_back({hi - n_sl, n_sl, true});
                    continue;
                }
            }
            uint32_t first = lo;
            uint64_t used = 0;
            for (uint32_t r = lo; r < hi; r++) {
                uint64_t w = p.cls_scratch[k][r];
                if (r > first && used + w > scratch_cap_words) {
                    b->cls_chunks[k][L].push_back({first, r - first});
                    first = r;
                    used = 0;
                }
                scratch_off[2 * r] = (uint32_t)used;
                scratch_off[2 * r + 1] = (uint32_t)w;
                used += w;
                need = std::max(need, used);
            }
            if (hi > first) b->cls_chunks[k][L].push_back({first, hi - first});
        }
        // the exact kernels use slot 0 of the same buffer: it must hold the largest single record
        for (uint32_t oi = 0; oi < p.n_opcodes; oi++)
            if (p.prog_class[oi] == (uint32_t)k) {
                need = std::max<uint64_t>(need, p.prog_scratch[oi]);
                b->cls_exact_words[k] = std::max<uint64_t>(b->cls_exact_words[k], p.prog_scratch[oi]);
            }
        if (int rc = upload(&b->d_cls_offset[k], p.cls_offset[k])) return rc;
        if (int rc = upload(&b->d_cls_scratch_off[k], scratch_off)) return rc;
        if (need) HIPCHK(hipMalloc((void **)&b->d_cls_scratch[k], (size_t)need * b->Bp * 4));
    }
    b->dp.prog = b->d_prog;
    b->dp.prog_offset = b->d_prog_offset;
    b->dp.consts = b->d_consts;
    b->dp.bytecode = b->d_bytecode;
    b->dp.Mem = b->d_Mem;
    b->dp.grumpkin = GrumpkinTables{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    b->dp.ped_seed = nullptr;
    b->dp.ecdsa_g = nullptr;
    b->dp.fc_store = nullptr;
    b->dp.slot_of = nullptr;
    {   // limits of the Brillig VM for the level kernels and the first pass of the exact kernels (tuning.hpp)
        const Tuning &tn = p.tune;
        b->dp.brillig.steps = 1u << (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_steps_log2, 0), 31);
        b->dp.brillig.call_depth = (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_call_depth, 1), 1 << 20);
        b->dp.brillig.mem_cap = 0;
        b->dp.brillig.stride = 0;
        for (uint32_t oi = 0; oi < p.n_opcodes; oi++)
            if (p.prog[p.prog_offset[oi]] == PK_BRILLIG) b->br_max_regs = std::max(b->br_max_regs, p.prog[p.prog_offset[oi] + 7]);
    }
    if (p.n_digest_segments) {
        HIPCHK(hipMalloc((void **)&b->d_leaves, (size_t)p.n_digest_segments * 2 * b->Bp * sizeof(uint4)));
        if (int rc = ensure_digest_tables(b)) return rc;
    }
    if (!p.slot_of.empty()) {
        if (int rc = upload(&b->d_slot_of, p.slot_of)) return rc;
        b->dp.slot_of = b->d_slot_of;
        std::vector<uint32_t> rows(p.initial_ids.size());
        for (size_t i = 0; i < rows.size(); i++) rows[i] = p.slot_of[p.initial_ids[i]];
        if (int rc = upload(&b->d_init_rows, rows)) return rc;
    }
    if (!p.fc_slot_opcode.empty()) {
        b->fc_slots.resize(p.fc_slot_opcode.size());
        std::vector<FcStoreSlot> tab(b->fc_slots.size(), FcStoreSlot{nullptr, nullptr});
        if (int rc = upload(&b->d_fc_store, tab)) return rc;
        b->dp.fc_store = b->d_fc_store;
    }
    if (p.needs_ecdsa || p.needs_grumpkin) {  // the handle holds the device's table set until it is destroyed
        device_tables_retain(b->device);
        b->holds_tables = true;
    }
    if (p.needs_ecdsa) {
        b->dp.ecdsa_g = ecdsa_generator_tables();
        if (!b->dp.ecdsa_g) return set_err(ACVM_E_DEVICE, "could not build the ECDSA generator tables on the device");
    }
    if (p.needs_grumpkin) {
        // the level schedule's Pedersen kernel reads the 503 MB pair table (one mixed addition per 18 bits of input)
        const bool pairs = !p.cls_offset[CLS_PEDERSEN].empty();
        const bool windows = pairs && p.tune.pedersen_window_bits == (int64_t)GRUMPKIN_PEDW_BITS;  // 6.4 GB of 22-bit windows instead of the 503 MB of pairs
        bool windows_built = false, have = false;
        GrumpkinTables tabs;
        const GrumpkinTables *t = &tabs;
        if (windows) {
            have = windows_built = grumpkin_window_table(&tabs);
            if (!have) (void)hipGetLastError();  // (no room for 6.4 GB beside what the process holds: the pair table serves the same kernel)
        }
        if (!have) have = pairs ? grumpkin_pair_table(&tabs) : grumpkin_tables(&tabs);
        if (!have) return set_err(ACVM_E_DEVICE, "could not build the Grumpkin tables on the device");
        b->dp.grumpkin = *t;
        if (!windows_built) b->dp.grumpkin.pedw = nullptr;  // (another batch of the process may have built it: this one was planned for the pair table)
        if (!p.pedersen_seeds.empty()) {  // the instance-independent head of every Pedersen chain (kernels_grumpkin.hip)
            std::vector<uint32_t> keys;
            for (auto &k : p.pedersen_seeds) { keys.push_back(k.first); keys.push_back(k.second); }
            uint32_t *d_keys = nullptr;
            if (int rc = upload(&d_keys, keys)) return rc;
            HIPCHK(hipMalloc((void **)&b->d_ped_seed, p.pedersen_seeds.size() * 64));
            launch_pedersen_seeds(b->stream, *t, d_keys, (uint32_t)p.pedersen_seeds.size(), b->d_ped_seed);
            HIPCHK(hipStreamSynchronize(b->stream));
            hipFree(d_keys);
        }
        b->dp.ped_seed = b->d_ped_seed;
    }
    b->n_words = (p.n_witnesses + 31) / 32;
    if (int rc = upload(&b->d_producer, p.producer)) return rc;
    if (int rc = upload(&b->d_prog_class, p.prog_class)) return rc;
    if (int rc = upload(&b->d_dyn_offset, p.dyn_offset)) return rc;
    {
        size_t bytes = (size_t)p.n_inverse_slots * 2 * b->Bp * sizeof(uint4);
        HIPCHK(hipMalloc((void **)&b->d_inv, bytes ? bytes : 16));
    }
    HIPCHK(hipMalloc((void **)&b->d_event, ((size_t)b->B + 2) * 4));  // + the count of flagged instances and a ticket (kernels.hip event_count_kernel)
    HIPCHK(hipHostMalloc((void **)&b->h_flag_count, 64, hipHostMallocMapped));  // the same count where the host can read it after a synchronisation
    b->unscale = Unscale{b->d_unscale_index, b->d_unscale_consts, b->d_unscale_plain, b->d_scaled_ids, (uint32_t)p.scaled_ids.size(), b->d_event};
    b->h_event.assign(b->B, 0xFFFFFFFFu);
    b->slow_index.assign(b->B, -1);
    return 0;
}

acvm_batch_t *acvm_batch_new(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances,
                             const uint32_t *initial_ids, uint32_t n_initial) {
    return acvm_batch_new_ex(c, solver, n_instances, initial_ids, n_initial, 0, nullptr, 0);
}
acvm_batch_t *acvm_batch_new_ex(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances, const uint32_t *initial_ids,
                                uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids, uint32_t n_keep) try {
    if (!c || (n_initial && !initial_ids) || (n_keep && !keep_ids)) { set_err(ACVM_E_INVALID, "null argument"); return nullptr; }
    if (flags & ~(uint32_t)(ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) { set_err(ACVM_E_INVALID, "unknown batch flag"); return nullptr; }
    if ((flags & ACVM_BATCH_REUSE_SLOTS) && solver) { set_err(ACVM_E_UNSUPPORTED, "slot reuse with a caller-supplied BlackBoxFunctionSolver"); return nullptr; }
    if (solver && (!solver->schnorr_verify || !solver->pedersen || !solver->fixed_base_scalar_mul)) {
        set_err(ACVM_E_INVALID, "acvm_bb_solver_t with a null function pointer");
        return nullptr;
    }
    {
        std::vector<uint32_t> ids(initial_ids, initial_ids + n_initial);
        std::sort(ids.begin(), ids.end());
        if (std::adjacent_find(ids.begin(), ids.end()) != ids.end()) { set_err(ACVM_E_INVALID, "duplicate initial witness id"); return nullptr; }
    }
    auto b = std::make_unique<acvm_batch>();
    if (solver) { b->has_solver = true; b->solver = *solver; }
    b->opts.host_blackbox = solver != nullptr;
    b->opts.fold_digest = (flags & (ACVM_BATCH_FOLD_DIGEST | ACVM_BATCH_REUSE_SLOTS)) != 0;  // a recycled row must be hashed before it is reused
    b->opts.reuse_slots = (flags & ACVM_BATCH_REUSE_SLOTS) != 0;
    b->opts.keep.assign(keep_ids, keep_ids + n_keep);
    b->plan = build_plan(*c->c, initial_ids, n_initial, b->opts);
    if (!b->plan.unsupported.empty()) {
        set_err(ACVM_E_UNSUPPORTED, b->plan.unsupported);
        return nullptr;
    }
    b->B = b->capacity = n_instances;
    b->Bp = ((uint64_t)n_instances + 63) / 64 * 64;
    if (batch_init(b.get()) != 0) return nullptr;
    return b.release();
} ABI_CATCH_PTR
void acvm_batch_free(acvm_batch_t *b) { delete b; }

int batch_import_async(acvm_batch *b, const void *d_values_be32, hipEvent_t imported) {
    HIPCHK(hipSetDevice(b->device));
    // (acvm_batch_solve_then_import put exactly this import behind the previous solve, and it ran: the rows are there, in stream order)
    const bool already = b->next_imported && b->next_inputs == d_values_be32;
    b->next_imported = false;
    b->next_inputs = nullptr;
    if (!already)
        launch_import(b->stream, b->d_W, b->Bp, b->B, (const uint8_t *)d_values_be32, b->reuse() ? b->d_init_rows : b->d_init_ids,
                      (uint32_t)b->plan.initial_ids.size());
    HIPCHK(hipGetLastError());
    if (imported) HIPCHK(hipEventRecord(imported, b->stream));
    b->inputs_set = true;
    b->solved = false;
    b->stepping = false;
    clear_fc_store(b);
    return 0;
}
int acvm_batch_set_initial_witness_device(acvm_batch_t *b, const void *d_values_be32) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    const bool already = b->next_imported && b->next_inputs == d_values_be32;
    if (int rc = batch_import_async(b, d_values_be32, nullptr)) return rc;
    // the caller may reuse its buffer as soon as the call returns (an import that ran behind the previous solve left the buffer alone since)
    if (!already) HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
} ABI_CATCH

int acvm_batch_set_initial_witness(acvm_batch_t *b, const uint8_t *values_be32) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    size_t bytes = (size_t)b->B * b->plan.initial_ids.size() * 32;
    if (bytes && !values_be32) return set_err(ACVM_E_INVALID, "null values");
    HIPCHK(hipSetDevice(b->device));
    if (int rc = stage_reserve(b, bytes)) return rc;
    if (bytes) HIPCHK(hipMemcpyAsync(b->d_stage, values_be32, bytes, hipMemcpyHostToDevice, b->stream));
    return acvm_batch_set_initial_witness_device(b, b->d_stage);
} ABI_CATCH

int acvm_batch_set_force_slow_path(acvm_batch_t *b, int on) {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    if (on && b->reuse()) return set_err(ACVM_E_UNSUPPORTED, "the exact path for every instance needs the full witness table: not with ACVM_BATCH_REUSE_SLOTS");
    b->force_slow = on != 0;
    return 0;
}
int acvm_batch_set_profiling(acvm_batch_t *b, int on) {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    b->profiling = on != 0;
    return 0;
}
int acvm_batch_set_instances(acvm_batch_t *b, uint32_t n) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    return batch_set_live_count(b, n);
} ABI_CATCH
int acvm_batch_reset(acvm_batch_t *b) {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    b->solved = false;
    b->stepping = false;
    clear_fc_store(b);
    return 0;
}

static int ensure_slow_capacity(acvm_batch *b, uint32_t n) {
    if (n <= b->slow_cap) return 0;
    for (void *p : {(void *)b->d_slow_ids, (void *)b->d_assigned, (void *)b->d_slow_res, (void *)b->d_slow_start})
        if (p) hipFree(p);
    b->d_slow_ids = nullptr; b->d_assigned = nullptr; b->d_slow_res = nullptr; b->d_slow_start = nullptr;
    HIPCHK(hipMalloc((void **)&b->d_slow_ids, (size_t)n * 4));
    HIPCHK(hipMalloc((void **)&b->d_slow_start, (size_t)n * 4));
    HIPCHK(hipMalloc((void **)&b->d_assigned, (size_t)n * (b->n_words ? b->n_words : 1) * 4));
    HIPCHK(hipMalloc((void **)&b->d_slow_res, (size_t)n * sizeof(SlowResult)));
    b->slow_cap = n;
    return 0;
}

static ExactLanes exact_lanes(acvm_batch *b, uint32_t n_slow) {
    FcLanes fc{b->d_fc_pend_desc, b->fc_pend_desc_words, b->d_fc_pend_vals, b->fc_pend_vals_cap};
    return ExactLanes{b->xids(), n_slow, b->d_assigned, b->d_slow_start, b->d_slow_res, fc, b->br_retry_active ? b->d_br_lane : nullptr};
}

// Foreign-call round trip, device side: the buffers a pending call's inputs are written to (per exact lane, FcLanes::pend_*) and
// the store of the results the host resolved (per opcode slot and instance, FcStoreSlot): dirty slots are rebuilt and uploaded.
static int upload_fc_tables(acvm_batch *b, uint32_t n_slow) {
    const Plan &p = b->plan;
    if (!p.has_foreign_calls) return 0;
    b->fc_lane.resize(n_slow);
    const uint32_t pend_words = 1 + p.fc_max_inputs, pend_vals = (uint32_t)std::max<uint64_t>(1, p.fc_pending_vals);
    if (n_slow > b->fc_lanes_cap) {
        for (void *q : {(void *)b->d_fc_pend_desc, (void *)b->d_fc_pend_vals})
            if (q) hipFree(q);
        b->d_fc_pend_desc = nullptr;
        b->d_fc_pend_vals = nullptr;
        b->fc_lanes_cap = n_slow;
        b->fc_pend_desc_words = pend_words;
        b->fc_pend_vals_cap = pend_vals;
        HIPCHK(hipMalloc((void **)&b->d_fc_pend_desc, (size_t)pend_words * n_slow * 4));
        HIPCHK(hipMalloc((void **)&b->d_fc_pend_vals, (size_t)pend_vals * 2 * n_slow * sizeof(uint4)));
    }
    bool any = false;
    for (size_t si = 0; si < b->fc_slots.size(); si++) {
        auto &sl = b->fc_slots[si];
        if (!sl.dirty) continue;
        sl.dirty = false;
        any = true;
        uint32_t desc_words = 1, vals = 1;
        for (auto &kv : sl.inst) {
            uint32_t dw = 1, nv = 0;
            for (auto &res : kv.second) {
                dw += 1 + 2 * (uint32_t)res.size();
                for (auto &v : res) nv += (uint32_t)v.vals.size();
            }
            desc_words = std::max(desc_words, dw);
            vals = std::max(vals, nv);
        }
        if (desc_words > sl.desc_words || vals > sl.vals_cap) {
            for (void *q : {(void *)sl.d_desc, (void *)sl.d_vals})
                if (q) hipFree(q);
            sl.d_desc = nullptr;
            sl.d_vals = nullptr;
            sl.desc_words = desc_words + 8;
            sl.vals_cap = vals + 8;
            HIPCHK(hipMalloc((void **)&sl.d_desc, (size_t)sl.desc_words * b->Bp * 4));
            HIPCHK(hipMalloc((void **)&sl.d_vals, (size_t)sl.vals_cap * 2 * b->Bp * sizeof(uint4)));
        }
        // word w of instance j at desc[w * Bp + j]; value i: halves at (2 i) * Bp + j and (2 i + 1) * Bp + j, 4 words each
        std::vector<uint32_t> desc((size_t)sl.desc_words * b->Bp, 0), vbuf((size_t)sl.vals_cap * 2 * b->Bp * 4, 0);
        for (auto &kv : sl.inst) {
            const uint64_t j = kv.first;
            uint32_t w = 0, vi = 0;
            desc[(size_t)(w++) * b->Bp + j] = (uint32_t)kv.second.size();
            for (auto &res : kv.second) {
                desc[(size_t)(w++) * b->Bp + j] = (uint32_t)res.size();
                for (auto &v : res) {
                    desc[(size_t)(w++) * b->Bp + j] = v.is_array ? 1u : 0u;
                    desc[(size_t)(w++) * b->Bp + j] = (uint32_t)v.vals.size();
                    for (auto &xh : v.vals) {
                        const FrH x = frh::to_device_form(xh);
                        memcpy(&vbuf[((size_t)(2 * vi) * b->Bp + j) * 4], &x.l[0], 16);
                        memcpy(&vbuf[((size_t)(2 * vi + 1) * b->Bp + j) * 4], &x.l[2], 16);
                        vi++;
                    }
                }
            }
        }
        HIPCHK(hipMemcpy(sl.d_desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sl.d_vals, vbuf.data(), vbuf.size() * 4, hipMemcpyHostToDevice));
    }
    if (any && b->d_fc_store) {  // the kernels reach the tables through this array: the pointers may have moved
        std::vector<FcStoreSlot> tab(b->fc_slots.size());
        for (size_t si = 0; si < tab.size(); si++) tab[si] = FcStoreSlot{b->fc_slots[si].d_desc, b->fc_slots[si].d_vals};
        HIPCHK(hipMemcpy(b->d_fc_store, tab.data(), tab.size() * sizeof(FcStoreSlot), hipMemcpyHostToDevice));
    }
    return 0;
}
// One Pedersen / FixedBaseScalarMul / SchnorrVerify opcode through the caller's BlackBoxFunctionSolver callbacks
// (blackbox_solver/src/lib.rs:27-45) for the instances of the level schedule (exact == false, all B instances) or for the
// exact lanes. Inputs leave the device as canonical big-endian bytes, outputs come back the same way. All instances are
// gathered ONCE (one kernel, one copy), the callbacks run in one loop -- or in ONE call when the vtable has the *_batch
// member -- and the results are scattered once; a pass is capped at 2^18 instances only to bound the staging buffers.
static int run_host_blackbox(acvm_batch *b, uint32_t opcode, bool exact, uint32_t n_slow) {
    const Plan &p = b->plan;
    hipStream_t s = b->stream;
    const uint32_t *rec = &p.prog[p.prog_offset[opcode]];
    std::vector<uint32_t> sel, outs;
    uint32_t func = 0;
    switch (rec[0]) {
    case PK_FIXED_BASE: func = BB_FIXED_BASE_SCALAR_MUL; sel = {rec[2], rec[3]}; outs = {rec[4], rec[5], rec[6], rec[7]}; break;
    case PK_PEDERSEN: func = BB_PEDERSEN; sel.assign(rec + 8, rec + 8 + rec[3]); outs = {rec[4], rec[5], rec[6], rec[7]}; break;
    case PK_SCHNORR: func = BB_SCHNORR_VERIFY; sel = {rec[2], rec[3]}; sel.insert(sel.end(), rec + 8, rec + 8 + rec[4] + rec[5]); outs = {rec[6], rec[7]}; break;
    default: return set_err(ACVM_E_INVALID, "not a black box function of the solver trait");
    }
    const uint32_t n_sel = (uint32_t)sel.size(), n_out = (uint32_t)outs.size() / 2;
    const uint32_t n_total = exact ? n_slow : b->B;
    if (!n_total) return 0;
    const uint32_t chunk = std::min<uint32_t>(n_total, 1u << 18);
    const size_t in_row = (size_t)std::max<uint32_t>(n_sel, 1) * 32, out_row = (size_t)n_out * 32;
    // arena: sel | outs | active | in | rc | vals
    const size_t o_sel = 0, o_outs = o_sel + align256((size_t)std::max<uint32_t>(n_sel, 1) * 4), o_active = o_outs + align256(outs.size() * 4),
                 o_in = o_active + align256(n_total), o_rc = o_in + align256(chunk * in_row), o_vals = o_rc + align256(chunk);
    if (int rc = stage_reserve(b, o_vals + align256(chunk * out_row))) return rc;
    uint32_t *d_sel = (uint32_t *)(b->d_stage + o_sel), *d_outs = (uint32_t *)(b->d_stage + o_outs);
    uint8_t *d_active = b->d_stage + o_active, *d_in = b->d_stage + o_in, *d_rc = b->d_stage + o_rc, *d_vals = b->d_stage + o_vals;
    std::vector<uint8_t> active(n_total, 1);
    if (n_sel) HIPCHK(hipMemcpyAsync(d_sel, sel.data(), (size_t)n_sel * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_outs, outs.data(), outs.size() * 4, hipMemcpyHostToDevice, s));
    const ExactLanes L = exact_lanes(b, n_slow);
    if (exact) {
        launch_hostbb_precheck(s, L, opcode, d_sel, n_sel, d_active);
        HIPCHK(hipMemcpyAsync(active.data(), d_active, n_total, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    std::vector<uint8_t> in(chunk * in_row), rc(chunk), vals(chunk * out_row);
    static constexpr size_t ERR_STRIDE = 200;
    const acvm_bb_solver_t &sv = b->solver;
    for (uint32_t first = 0; first < n_total; first += chunk) {
        const uint32_t m = std::min(chunk, n_total - first);
        launch_hostbb_gather(s, b->d_W, b->Bp, exact ? b->d_slow_ids : nullptr, first, m, d_sel, n_sel, d_in);
        if (n_sel) HIPCHK(hipMemcpyAsync(in.data(), d_in, (size_t)m * n_sel * 32, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::fill(vals.begin(), vals.end(), 0);
        // the instances that really call the solver, packed
        std::vector<uint32_t> who;
        for (uint32_t i = 0; i < m; i++) {
            rc[i] = 255;
            if (active[first + i]) who.push_back(i);
        }
        const bool batched = (rec[0] == PK_FIXED_BASE && sv.fixed_base_scalar_mul_batch) || (rec[0] == PK_PEDERSEN && sv.pedersen_batch) ||
                             (rec[0] == PK_SCHNORR && sv.schnorr_verify_batch);
        auto instance_of = [&](uint32_t i) { return exact ? b->slow_ids[first + i] : first + i; };
        if (batched && !who.empty()) {
            const size_t n = who.size();
            std::vector<uint8_t> brc(n, 0), bout(n * 64, 0);
            std::vector<char> berr(n * ERR_STRIDE, 0);
            int r = 0;
            if (rec[0] == PK_FIXED_BASE) {
                std::vector<uint8_t> lh(n * 64);
                for (size_t q = 0; q < n; q++) memcpy(&lh[q * 64], &in[(size_t)who[q] * in_row], 64);
                r = sv.fixed_base_scalar_mul_batch(sv.ctx, n, lh.data(), bout.data(), brc.data(), berr.data(), ERR_STRIDE);
            } else if (rec[0] == PK_PEDERSEN) {
                const size_t k = rec[3];
                std::vector<uint8_t> pin(n * std::max<size_t>(k, 1) * 32);
                for (size_t q = 0; q < n; q++) memcpy(&pin[q * k * 32], &in[(size_t)who[q] * in_row], k * 32);
                r = sv.pedersen_batch(sv.ctx, n, pin.data(), k, rec[2], bout.data(), brc.data(), berr.data(), ERR_STRIDE);
            } else {  // to_u8_vec (signature/mod.rs:5-18): the last big-endian byte of each witness
                const uint32_t n_sig = rec[4], n_msg = rec[5];
                std::vector<uint8_t> pk(n * 64), sig(n * std::max<uint32_t>(n_sig, 1)), msg(n * std::max<uint32_t>(n_msg, 1)), ok(n, 0);
                for (size_t q = 0; q < n; q++) {
                    const uint8_t *a = &in[(size_t)who[q] * in_row];
                    memcpy(&pk[q * 64], a, 64);
                    for (uint32_t k = 0; k < n_sig; k++) sig[q * n_sig + k] = a[(size_t)(2 + k) * 32 + 31];
                    for (uint32_t k = 0; k < n_msg; k++) msg[q * n_msg + k] = a[(size_t)(2 + n_sig + k) * 32 + 31];
                }
                r = sv.schnorr_verify_batch(sv.ctx, n, pk.data(), sig.data(), n_sig, msg.data(), n_msg, ok.data(), brc.data(), berr.data(), ERR_STRIDE);
                for (size_t q = 0; q < n; q++) bout[q * 64 + 31] = ok[q] ? 1 : 0;
            }
            for (size_t q = 0; q < n; q++) {
                const uint32_t i = who[q];
                const int ri = r != 0 ? 3 : brc[q];  // a failing batch call fails every instance of it like a panic
                rc[i] = (uint8_t)(ri > 2 ? 3 : ri);
                memcpy(&vals[(size_t)i * out_row], &bout[q * 64], out_row);
                if (rc[i] != 0) {
                    berr[q * ERR_STRIDE + ERR_STRIDE - 1] = 0;
                    b->host_bb_msg[instance_of(i)] = r != 0 ? "batched BlackBoxFunctionSolver call failed" : &berr[q * ERR_STRIDE];
                }
            }
        } else {
            char err[ERR_STRIDE];
            for (uint32_t i : who) {
                const uint8_t *a = &in[(size_t)i * in_row];
                uint8_t *o = &vals[(size_t)i * out_row];
                err[0] = 0;
                int r = 0;
                if (rec[0] == PK_FIXED_BASE) r = sv.fixed_base_scalar_mul(sv.ctx, a, a + 32, o, o + 32, err, sizeof err);
                else if (rec[0] == PK_PEDERSEN) r = sv.pedersen(sv.ctx, a, rec[3], rec[2], o, o + 32, err, sizeof err);
                else {  // to_u8_vec (signature/mod.rs:5-18): the last big-endian byte of each witness
                    const uint32_t n_sig = rec[4], n_msg = rec[5];
                    std::vector<uint8_t> sig(n_sig + 1), msg(n_msg + 1);
                    for (uint32_t k = 0; k < n_sig; k++) sig[k] = a[(size_t)(2 + k) * 32 + 31];
                    for (uint32_t k = 0; k < n_msg; k++) msg[k] = a[(size_t)(2 + n_sig + k) * 32 + 31];
                    uint8_t ok = 0;
                    r = sv.schnorr_verify(sv.ctx, a, a + 32, sig.data(), n_sig, msg.data(), n_msg, &ok, err, sizeof err);
                    o[31] = ok ? 1 : 0;
                }
                rc[i] = (uint8_t)(r < 0 || r > 2 ? 3 : r);
                if (r != 0) b->host_bb_msg[instance_of(i)] = err;
            }
        }
        HIPCHK(hipMemcpyAsync(d_rc, rc.data(), m, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_vals, vals.data(), (size_t)m * out_row, hipMemcpyHostToDevice, s));
        if (exact) launch_hostbb_apply_exact(s, b->d_W, b->Bp, L, first, m, opcode, func, d_outs, n_out, d_active, d_rc, d_vals);
        else launch_hostbb_apply_level(s, b->d_W, b->Bp, first, m, opcode, func, d_outs, n_out, d_rc, d_vals, b->d_event);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    return 0;
}

// run the exact in-order kernels over the current lanes from opcode min_start on and fetch the outcomes
// (stepping: the lanes executed every earlier opcode themselves, nothing is replayed; only opcodes [min_start, end_opcode) run)
static int run_exact_segments(acvm_batch *b, uint32_t n_slow, uint32_t min_start, bool replay = true, uint32_t end_opcode = 0xFFFFFFFFu) {
    const Plan &p = b->plan;
    hipStream_t s = b->xstream();  // the batch's stream, or the side stream of an asynchronous job
    const ExactLanes L = exact_lanes(b, n_slow);
    const DeviceProgram xdp = b->xdp();
    // Memory side effects of the opcodes before the earliest event are replayed by the kernel, so the run starts at the first opcode
    // when the circuit has memory blocks, else at the earliest event. One launch covers every class (kernels_brillig.hip
    // exact_run_kernel); only the opcodes of a caller-supplied BlackBoxFunctionSolver split it (host callbacks in between).
    const bool has_mem = replay && p.mem_cells != 0;
    const ExactScratch sc{b->xscratch(CLS_HASH), b->xscratch(CLS_GRUMPKIN), b->br_retry_active ? b->d_br_scratch : b->xscratch(CLS_BRILLIG)};
    const uint32_t end = std::min(end_opcode, p.n_opcodes);
    uint32_t at = has_mem ? 0u : std::min(min_start, end);
    while (at < end) {
        uint32_t stop = at;
        while (stop < end && p.prog_class[stop] != CLS_HOSTBB) stop++;
        launch_exact_run(s, b->xW(), b->xBp(), xdp, L, at, stop, has_mem, b->d_prog_class, sc);
        if (stop < end) {
            if (stop >= min_start)  // (no lane stands before an opcode in front of the earliest start)
                if (int rc = run_host_blackbox(b, stop, true, n_slow)) return rc;
            stop++;
        }
        at = stop;
    }
    launch_exact_finish(s, L, b->stepping ? p.n_opcodes : 0u);
    HIPCHK(hipGetLastError());
    b->slow_res.resize(n_slow);
    // (a copy into pageable host memory blocks the caller until the stream has drained: an asynchronous job fetches its lanes' results
    // when it is collected, batch_finish_pending)
    if (!b->pending) HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
    b->pend_host_valid = false;
    return 0;
}

// The Brillig VM of the reference has no limits: memory grows on write (brillig_vm/src/memory.rs:27-39), a program may run any number
// of steps and nest calls to any depth (lib.rs:154-307). The kernels run with a memory capacity, a step limit and a call-stack depth
// (BrilligLimits); a lane that reaches one ends its pass with DE_PANIC and one of the three device-limit codes. Such lanes are
// RETRIED here: their opcode runs again (a failed VM run has no side effects: outputs are inserted after it finishes) with the
// limit that was hit raised -- memory to twice the cell the write wanted, steps and depth sixteen-fold -- in a scratch that holds
// only the retried lanes, until nothing hits a limit or a stated maximum of the library is reached (tuning.hpp: 2^26 steps, 2^16
// frames, 2^22 cells by default). Past a maximum THAT INSTANCE ends with status Failure / ACVM_ERR_DEVICE_LIMIT -- an outcome the
// reference does not have and the header says so: "this library could not finish the instance, run it with the reference" -- and every
// other instance of the batch keeps its result (round 3 failed the whole solve call and lost them: the reference's caller loop,
// acvm_js/src/execute.rs:60-119, loses one instance at most). Called with the lanes' results on the host (stream synchronised).
static bool is_device_limit(const SlowResult &r) {
    return r.status == ACVM_STATUS_FAILURE && r.err == ACVM_ERR_PANIC && (r.msg == 17u || r.msg == 18u || r.msg == 28u);
}
// the lane's final word: Failure / ACVM_ERR_DEVICE_LIMIT at its Brillig opcode, aux0 = the limit that was reached, aux1 = its value
static void give_up_lane(SlowResult &r, uint32_t kind, uint64_t limit, uint64_t wanted) {
    const uint32_t opcode = r.opcode_index;
    memset(&r, 0, sizeof r);
    r.status = ACVM_STATUS_FAILURE;
    r.err = ACVM_ERR_DEVICE_LIMIT;
    r.opcode_index = opcode;
    r.aux0 = kind;
    r.aux1 = (uint32_t)std::min<uint64_t>(limit, 0xFFFFFFFFu);
    r.msg = 29u;  // DM_DEVICE_LIMIT (ops_common.hpp): format_message words it
    r.x0 = (uint32_t)std::min<uint64_t>(wanted, 0xFFFFFFFFu);
}
static int retry_device_limits(acvm_batch *b, uint32_t n_slow, bool replay, uint32_t end_opcode) {
    const Plan &p = b->plan;
    const Tuning &tn = p.tune;
    hipStream_t s = b->xstream();
    const BrilligLimits base = b->dp.brillig;
    const uint64_t max_steps = 1ull << (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_steps_max_log2, 0), 31);
    const uint64_t max_depth = (uint64_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_call_depth_max, 1), 1 << 24);
    const uint64_t max_cells = 1ull << (uint32_t)std::min<int64_t>(std::max<int64_t>(tn.brillig_mem_max_log2, 0), 30);
    BrilligLimits lim = base;
    int rc = 0;
    for (;;) {
        // the memory a retried lane runs with so far (0 = the planner's estimate of its record)
        uint64_t record_cells = 0;
        for (uint32_t t = 0; t < n_slow; t++) {
            const SlowResult &r = b->slow_res[t];
            if (is_device_limit(r) && r.opcode_index < p.n_opcodes && p.prog[p.prog_offset[r.opcode_index]] == PK_BRILLIG)
                record_cells = std::max<uint64_t>(record_cells, p.prog[p.prog_offset[r.opcode_index] + 8]);
        }
        const uint64_t cur_cells = lim.mem_cap ? lim.mem_cap : record_cells;
        // lanes past a stated maximum are final; the rest is retried with the limits they reached raised
        std::vector<uint32_t> lanes;
        bool hit_steps = false, hit_depth = false, hit_mem = false;
        uint64_t want_cells = 0;
        for (uint32_t t = 0; t < n_slow; t++) {
            SlowResult &r = b->slow_res[t];
            if (!is_device_limit(r)) continue;
            if (r.msg == 18u && lim.steps >= max_steps) { give_up_lane(r, ACVM_LIMIT_BRILLIG_STEPS, max_steps, 0); continue; }
            if (r.msg == 28u && lim.call_depth >= max_depth) { give_up_lane(r, ACVM_LIMIT_BRILLIG_CALL_DEPTH, max_depth, 0); continue; }
            if (r.msg == 17u && ((uint64_t)r.x0 + 1 > max_cells || cur_cells >= max_cells)) { give_up_lane(r, ACVM_LIMIT_BRILLIG_MEMORY, max_cells, r.x0); continue; }
            lanes.push_back(t);
            hit_steps |= r.msg == 18u;
            hit_depth |= r.msg == 28u;
            if (r.msg == 17u) {
                hit_mem = true;
                want_cells = std::max<uint64_t>(want_cells, (uint64_t)r.x0 + 1);
            }
        }
        if (lanes.empty()) break;
        if (hit_steps) lim.steps = (uint32_t)std::min<uint64_t>((uint64_t)lim.steps * 16, max_steps);
        if (hit_depth) lim.call_depth = (uint32_t)std::min<uint64_t>((uint64_t)lim.call_depth * 16, max_depth);
        uint64_t cells = std::max<uint64_t>(cur_cells, 64);
        if (hit_mem) cells = std::min<uint64_t>(std::max<uint64_t>(2 * want_cells, 4 * cells), max_cells);
        lim.mem_cap = (uint32_t)cells;
        lim.stride = ((uint64_t)lanes.size() + 63) / 64 * 64;
        const uint64_t words = ((uint64_t)b->br_max_regs + cells) * 8 + lim.call_depth + cells / 4 + 16;
        const size_t bytes = (size_t)words * lim.stride * 4;
        if (bytes > b->br_scratch_bytes) {
            if (b->d_br_scratch) hipFree(b->d_br_scratch);
            b->d_br_scratch = nullptr;
            b->br_scratch_bytes = 0;
            if (hipMalloc((void **)&b->d_br_scratch, bytes) != hipSuccess) {
                (void)hipGetLastError();  // the device cannot hold the VM scratch of these lanes: they are final too
                for (uint32_t t : lanes) give_up_lane(b->slow_res[t], ACVM_LIMIT_DEVICE_MEMORY, bytes >> 20, 0);
                break;
            }
            b->br_scratch_bytes = bytes;
        }
        if (n_slow > b->br_lane_cap) {
            if (b->d_br_lane) hipFree(b->d_br_lane);
            b->d_br_lane = nullptr;
            HIPCHK(hipMalloc((void **)&b->d_br_lane, (size_t)n_slow * 4));
            b->br_lane_cap = n_slow;
        }
        std::vector<uint32_t> col(n_slow, 0xFFFFFFFFu);
        uint32_t min_start = 0xFFFFFFFFu;
        for (size_t i = 0; i < lanes.size(); i++) {
            const uint32_t t = lanes[i];
            col[t] = (uint32_t)i;
            const uint32_t at = b->slow_res[t].opcode_index;
            b->slow_start[t] = at;  // the opcode runs again
            min_start = std::min(min_start, at);
            memset(&b->slow_res[t], 0, sizeof(SlowResult));
            b->slow_res[t].status = ACVM_STATUS_IN_PROGRESS;
        }
        HIPCHK(hipMemcpyAsync(b->d_br_lane, col.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_slow_res, b->slow_res.data(), (size_t)n_slow * sizeof(SlowResult), hipMemcpyHostToDevice, s));
        // (the lanes that were given up stay Failure on the host and on the device: the exact kernels skip a lane that is not InProgress;
        // run_exact_segments fetches the device's records back, which is why the host copy of those lanes is restored below)
        std::vector<std::pair<uint32_t, SlowResult>> final_lanes;
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_res[t].err == ACVM_ERR_DEVICE_LIMIT) final_lanes.push_back({t, b->slow_res[t]});
        b->dp.brillig = lim;
        b->br_retry_active = true;
        rc = run_exact_segments(b, n_slow, min_start, replay, end_opcode);
        if (!rc && b->pending) {  // (an asynchronous job leaves its results on the device: fetch them for the next look at the limits)
            if (hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s) != hipSuccess) rc = set_err(ACVM_E_DEVICE, "hipMemcpyAsync failed in a Brillig retry pass");
        }
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_err(ACVM_E_DEVICE, "hipStreamSynchronize failed in a Brillig retry pass");
        for (auto &fl : final_lanes) b->slow_res[fl.first] = fl.second;
        b->dp.brillig = base;
        b->br_retry_active = false;
        b->n_brillig_retries++;
        if (rc) break;
    }
    return rc;
}

static int count_not_solved(acvm_batch *b) {
    int n = 0;
    for (auto &r : b->slow_res)
        if (r.status != ACVM_STATUS_SOLVED) n++;
    return n;
}

// continue the instances whose pending foreign call was resolved (ACVM::solve after resolve_pending_foreign_call)
static int solve_resume(acvm_batch *b) {
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    uint32_t min_start = 0xFFFFFFFFu;
    std::vector<uint32_t> resumed;
    for (uint32_t t = 0; t < n_slow; t++)
        if (b->slow_res[t].status == ACVM_STATUS_REQUIRES_FOREIGN_CALL && b->fc_lane[t].resolved_new) {
            resumed.push_back(t);
            b->fc_lane[t].resolved_new = false;
            b->slow_start[t] = b->slow_res[t].opcode_index;
            min_start = std::min(min_start, b->slow_start[t]);
            b->slow_res[t].status = ACVM_STATUS_IN_PROGRESS;
        }
    if (resumed.empty()) return count_not_solved(b);
    if (int rc = upload_fc_tables(b, n_slow)) return rc;
    hipStream_t s = b->stream;
    HIPCHK(hipEventRecord(b->ev_start, s));
    HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(b->d_slow_res, b->slow_res.data(), (size_t)n_slow * sizeof(SlowResult), hipMemcpyHostToDevice, s));
    if (int rc = run_exact_segments(b, n_slow, min_start)) return rc;
    HIPCHK(hipStreamSynchronize(s));
    if (int rc = retry_device_limits(b, n_slow, true, 0xFFFFFFFFu)) return rc;
    HIPCHK(hipEventRecord(b->ev_end, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev_start, b->ev_end));
    b->solve_device_ms = ms;
    b->slow_path_ms = ms;
    return count_not_solved(b);
}

// acvm_batch_solve_opcode (one == true: ACVM::solve_opcode, pwg/mod.rs:243-303) and acvm_batch_solve after some steps
// (one == false: the loop of ACVM::solve :236-241 over what is left). Every instance is an exact lane whose instruction
// pointer is slow_start[t]; see include/acvm_amd.h for the batch semantics.
static int solve_stepping(acvm_batch *b, bool one) {
    const Plan &p = b->plan;
    hipStream_t s = b->stream;
    const uint32_t n_slow = b->B;
    if (!b->stepping) {
        b->slow_ids.resize(n_slow);
        b->events_clean = false;
        for (uint32_t j = 0; j < n_slow; j++) { b->slow_ids[j] = j; b->slow_index[j] = (int32_t)j; }
        b->slow_start.assign(n_slow, 0);
        std::fill(b->h_event.begin(), b->h_event.end(), 0u);
        b->host_bb_msg.clear();
        b->stepping = true;
        b->solved = true;
        SlowResult fresh;
        memset(&fresh, 0, sizeof fresh);
        fresh.status = ACVM_STATUS_IN_PROGRESS;
        b->slow_res.assign(n_slow, fresh);
        if (!n_slow) return 0;
        if (int rc = ensure_slow_capacity(b, n_slow)) return rc;
        launch_fill_u32(s, b->d_event, 0u, b->B);  // no column is scaled: the exact kernels write plain values
        HIPCHK(hipMemcpyAsync(b->d_slow_ids, b->slow_ids.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        launch_init_assigned(s, b->d_assigned, n_slow, b->n_words, p.n_witnesses, b->d_producer, b->d_slow_start);
        b->fc_lane.assign(n_slow, acvm_batch::FcLaneState());
        if (int rc = upload_fc_tables(b, n_slow)) return rc;
        launch_exact_init(s, exact_lanes(b, n_slow));
    } else {
        bool any = false;
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_res[t].status == ACVM_STATUS_REQUIRES_FOREIGN_CALL && b->fc_lane[t].resolved_new) {
                b->fc_lane[t].resolved_new = false;
                b->slow_start[t] = b->slow_res[t].opcode_index;  // the opcode re-runs its VM (mod.rs:220-227)
                b->slow_res[t].status = ACVM_STATUS_IN_PROGRESS;
                any = true;
            }
        if (any) {
            if (int rc = upload_fc_tables(b, n_slow)) return rc;
            HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(b->d_slow_res, b->slow_res.data(), (size_t)n_slow * sizeof(SlowResult), hipMemcpyHostToDevice, s));
        }
    }
    if (!n_slow) return 0;
    uint32_t ip = 0xFFFFFFFFu;
    for (uint32_t t = 0; t < n_slow; t++)
        if (b->slow_res[t].status == ACVM_STATUS_IN_PROGRESS) ip = std::min(ip, b->slow_start[t]);
    if (ip == 0xFFFFFFFFu) return count_not_solved(b);
    const uint32_t end = one ? std::min(ip + 1, p.n_opcodes) : p.n_opcodes;
    if (one) {  // advance the instruction pointers on the host between the opcode and the Solved test
        if (ip < p.n_opcodes)
            if (int rc = run_exact_segments(b, n_slow, ip, false, end)) return rc;
        b->slow_res.resize(n_slow);
        HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (int rc = retry_device_limits(b, n_slow, false, end)) return rc;  // (a retried lane stands on `ip` again and re-runs only that opcode)
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_res[t].status == ACVM_STATUS_IN_PROGRESS && b->slow_start[t] <= ip) b->slow_start[t] = end;
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        launch_exact_finish(s, exact_lanes(b, n_slow), p.n_opcodes);
        HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
    } else {
        if (int rc = run_exact_segments(b, n_slow, ip, false)) return rc;  // ends with the Solved test of the lanes still running
        HIPCHK(hipStreamSynchronize(s));
        if (int rc = retry_device_limits(b, n_slow, false, 0xFFFFFFFFu)) return rc;
        for (uint32_t t = 0; t < n_slow; t++)
            if (b->slow_start[t] < p.n_opcodes) b->slow_start[t] = p.n_opcodes;
        // (run_exact_segments finishes only lanes whose pointer is at the end: publish the pointers first)
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        launch_exact_finish(s, exact_lanes(b, n_slow), p.n_opcodes);
        HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    b->pend_host_valid = false;
    for (uint32_t t = 0; t < n_slow; t++) {
        SlowResult &r = b->slow_res[t];
        if (r.status == ACVM_STATUS_IN_PROGRESS) r.opcode_index = b->slow_start[t];  // ACVM::instruction_pointer
        else if (r.status == ACVM_STATUS_REQUIRES_FOREIGN_CALL || r.status == ACVM_STATUS_FAILURE) b->slow_start[t] = r.opcode_index;
    }
    return count_not_solved(b);
}

int acvm_batch_solve_opcode(acvm_batch_t *b) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    if (!b->inputs_set && !b->plan.initial_ids.empty()) return set_err(ACVM_E_STATE, "initial witness not set");
    if (b->solved && !b->stepping) return set_err(ACVM_E_STATE, "acvm_batch_solve_opcode after acvm_batch_solve: reset the batch first");
    if (b->reuse()) return set_err(ACVM_E_UNSUPPORTED, "stepping needs the full witness table: not with ACVM_BATCH_REUSE_SLOTS");
    HIPCHK(hipSetDevice(b->device));
    return solve_stepping(b, true);
} ABI_CATCH

// per-launch HIP-event pairs of one solve (profiling on)
struct LaunchTimers {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> reg_pairs, dyn_pairs, cls_pairs[N_CLS];
    size_t ev_used = 0;
};

// The level schedule of one solve, enqueued on the batch's streams: the gate levels and the light records on the main stream,
// the inversion batches on a second one, the heavy record classes on the heavy lanes. (One hipGraph of the whole schedule was
// measured in round 2 at -2 % on the 250 k-opcode circuit and ROCm 7.2's hipStreamEndCapture recursed without bound on the
// five-stream schedule of larger ones: removed.)
static int enqueue_level_schedule(acvm_batch *b, LaunchTimers *tm) {
    const Plan &p = b->plan;
    hipStream_t s = b->stream;
    hipStream_t s2 = p.tune.overlap ? b->stream_dyn : b->stream;  // (overlap = 0, a measurement aid, serialises the two level kernels)
    auto next_event = [&]() -> hipEvent_t {
        if (tm->ev_used == b->ev_pool.size()) {
            hipEvent_t e;
            hipEventCreate(&e);
            b->ev_pool.push_back(e);
        }
        return b->ev_pool[tm->ev_used++];
    };
    const bool prof = tm != nullptr;
    launch_event_reset(s, b->d_event, b->B);
    // Per level the constant-coefficient gates and the other record classes (stream s) and the gates that need a
    // per-instance inversion (stream s2, ALU/latency-bound) are independent and run concurrently; level L+1 of
    // either stream waits for level L of both.
    const size_t n_levels = p.n_levels;
    while (b->ev_sync.size() < 2 * n_levels + 1) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        b->ev_sync.push_back(e);
    }
    // Record classes that are bound by the integer pipe or by latency run beside the HBM-bound gate levels on three lanes of their own
    // (plan.hpp heavy_lane: Pedersen | Brillig | hashes, Grumpkin, ECDSA), each a stream in order. The records of lane q at level L
    // start when level L-1 of the main stream is done and the levels of the OTHER lanes whose outputs they read are done
    // (plan.lane_needs_lane); a level of the main stream (or an inversion batch) waits for a lane only up to the level whose outputs
    // it reads (plan.level_needs_heavy[lane]): a level that reads a hash output does not wait for the Pedersen launch beside it.
    auto heavy_cls = [](int k) { return k == CLS_HASH || k == CLS_GRUMPKIN || k == CLS_BRILLIG || k == CLS_PEDERSEN || k == CLS_ECDSA || k == CLS_DIGEST; };
    // (measured in round 1, one MI355X, 2^16 instances: config 3 0.50 -> 0.42 ms, config-5 mix 29.0 -> 27.9 ms)
    bool any_heavy = false, any_main = !p.gate_offset.empty() || !p.cls_offset[CLS_LIGHT].empty();
    bool lane_any[N_HEAVY_LANES] = {false, false, false, false};  // a lane without records never joins the schedule (nor a capture)
    for (int k = 0; k < (int)N_CLS; k++)
        if (heavy_cls(k) && !p.cls_offset[k].empty()) { any_heavy = true; lane_any[heavy_lane(k)] = true; }
    // A circuit of heavy records only kept everything on one stream until round 4 (round 2 had measured config 4 at 2.62 ... 3.03 ms from run to run
    // with the lanes side by side against a steady 2.84-2.89 on one stream, and config 3 0.18 instead of 0.15 ms). But a record kernel of the
    // integer-bound classes that follows a large launch -- the import of its tile -- on the SAME stream runs 16-44 % longer than on a stream of its
    // own (profiles/r04_import_effect.txt: config 4 import + solve 3.27 -> 2.15 ms, ECDSA 3.83 -> 3.32 ms per 2^16): such circuits take the lanes'
    // streams too (tuning heavy_only_streams); a circuit of byte-message hashes alone stays on the main stream.
    const bool integer_bound = !p.cls_offset[CLS_GRUMPKIN].empty() || !p.cls_offset[CLS_PEDERSEN].empty() || !p.cls_offset[CLS_ECDSA].empty() || !p.cls_offset[CLS_BRILLIG].empty();
    const bool one_stream = !p.tune.overlap || !p.tune.heavy_streams || (!any_main && !(p.tune.heavy_only_streams && integer_bound));
    hipStream_t lane_stream[N_HEAVY_LANES] = {one_stream ? s : b->stream_heavy, one_stream ? s : b->stream_heavy2, one_stream ? s : b->stream_heavy3,
                                              one_stream ? s : b->stream_digest};
    bool any_dyn = !p.dyn_offset.empty();
    const bool any_async = any_dyn || any_heavy;
    if (any_async) {
        HIPCHK(hipEventRecord(b->ev_sync[2 * n_levels], s));
        if (any_dyn) HIPCHK(hipStreamWaitEvent(s2, b->ev_sync[2 * n_levels], 0));
        if (any_heavy && !one_stream)
            for (int q = 0; q < N_HEAVY_LANES; q++)
                if (lane_any[q]) HIPCHK(hipStreamWaitEvent(lane_stream[q], b->ev_sync[2 * n_levels], 0));
    }
    hipEvent_t last_reg = nullptr, last_dyn = nullptr, last_lane[N_HEAVY_LANES] = {nullptr, nullptr, nullptr, nullptr};
    bool main_dirty = false;  // the main stream has launches behind last_reg
    // A lane waits for the main stream only as far as its records read it (plan.lane_needs_main): a hash of initial witnesses and of other
    // hashes never waits for the range checks launched beside it (config 3: the Keccak level no longer starts behind the RANGE kernel).
    std::vector<std::pair<uint32_t, hipEvent_t>> main_marks;  // (L, event): "main levels < L are done", in order
    uint32_t lane_main_waited[N_HEAVY_LANES] = {0, 0, 0, 0};
    uint32_t waited_inverse_level = 0, waited_heavy[N_HEAVY_LANES] = {0, 0, 0, 0}, lane_waited[N_HEAVY_LANES][N_HEAVY_LANES] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (size_t L = 0; L < n_levels; L++) {
        uint32_t n = p.level_start[L + 1] - p.level_start[L];
        uint32_t nd = p.dyn_level_start[L + 1] - p.dyn_level_start[L];
        bool s_work = n != 0, h_work = false;
        bool lane_used[N_HEAVY_LANES] = {false, false, false, false};
        for (int k = 0; k < (int)N_CLS; k++) {
            (heavy_cls(k) ? h_work : s_work) |= !b->cls_chunks[k][L].empty();
            if (heavy_cls(k) && !b->cls_chunks[k][L].empty()) lane_used[heavy_lane(k)] = true;
        }
        // "levels < L of the main stream are done": recorded only where another stream is about to wait for it (an event
        // between two gate launches costs more than the launch gap itself)
        if ((nd || h_work) && main_dirty) {
            HIPCHK(hipEventRecord(b->ev_sync[2 * L], s));
            last_reg = b->ev_sync[2 * L];
            main_marks.push_back({(uint32_t)L, last_reg});
            main_dirty = false;
        }
        hipEvent_t prev_reg = last_reg;
        // the level waits for an inversion batch only if one of its gates reads that batch's rows (the planner put those
        // gates after the batch, usually several levels after): the batch runs beside all the levels in between
        const uint32_t need = p.level_needs_inverse[L + 1];  // 1-based inversion level, 0 = none
        if (s_work && need > waited_inverse_level) {
            HIPCHK(hipStreamWaitEvent(s, b->ev_sync[2 * (need - 1) + 1], 0));
            waited_inverse_level = need;
        }
        for (int q = 0; q < N_HEAVY_LANES; q++) {
            const uint32_t need_h = p.level_needs_heavy[q][L + 1];  // 1-based level of the lane's records, 0 = none
            if (s_work && need_h > waited_heavy[q]) {
                if (!one_stream) HIPCHK(hipStreamWaitEvent(s, b->ev_heavy[4 * (need_h - 1) + q], 0));
                waited_heavy[q] = need_h;
            }
        }
        // the level's light records (not the straight-line Brillig ones: a kernel of their own) ride in the gate launch when there is one
        const LaunchChunk *fused_light = nullptr;
        if (n && p.tune.light_fuse)
            for (const LaunchChunk &ch : b->cls_chunks[CLS_LIGHT][L])
                if (!ch.coop && (uint64_t)n + ch.count <= 65535u) { fused_light = &ch; break; }
        if (n) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (prof) { e0 = next_event(); hipEventRecord(e0, s); }
            if (fused_light)
                launch_arith_light_level(s, b->d_W, b->Bp, b->B, b->d_gate_stream, b->d_gate_offset + p.level_start[L], n, b->d_inv, b->dp,
                                         b->d_cls_offset[CLS_LIGHT] + fused_light->first, fused_light->count, b->d_event);
            else launch_arith_level(s, b->d_W, b->Bp, b->B, b->d_gate_stream, b->d_gate_offset + p.level_start[L], n, b->d_consts, b->d_event, b->d_inv);
            if (prof) { e1 = next_event(); hipEventRecord(e1, s); tm->reg_pairs.push_back({e0, e1}); }
            b->n_launches += (n + 65534) / 65535;
        }
        // what the lanes wait for: levels < L of the main stream, and the other lanes as far as they read them
        if (!one_stream)
            for (int q = 0; q < N_HEAVY_LANES; q++) {
                if (!lane_used[q]) continue;
                if (const uint32_t need_m = p.lane_needs_main[q][L + 1]; need_m > lane_main_waited[q]) {
                    // the earliest mark behind main level need_m (1-based): "levels < mark" with mark >= need_m
                    auto it = std::lower_bound(main_marks.begin(), main_marks.end(), need_m, [](const std::pair<uint32_t, hipEvent_t> &mk, uint32_t v) { return mk.first < v; });
                    if (it != main_marks.end()) {
                        HIPCHK(hipStreamWaitEvent(lane_stream[q], it->second, 0));
                        lane_main_waited[q] = it->first;
                    }
                }
                for (int q2 = 0; q2 < N_HEAVY_LANES; q2++) {
                    const uint32_t need_l = p.lane_needs_lane[q][q2][L + 1];
                    if (q2 != q && lane_stream[q2] != lane_stream[q] && need_l > lane_waited[q][q2]) {
                        HIPCHK(hipStreamWaitEvent(lane_stream[q], b->ev_heavy[4 * (need_l - 1) + q2], 0));
                        lane_waited[q][q2] = need_l;
                    }
                }
            }
        for (int k = 0; k < (int)N_CLS; k++)
            for (const LaunchChunk &ch : b->cls_chunks[k][L]) {
                if (&ch == fused_light) continue;  // went with the gates
                hipStream_t sk = heavy_cls(k) ? lane_stream[heavy_lane(k)] : s;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (prof) { e0 = next_event(); hipEventRecord(e0, sk); }
                const uint32_t *off = b->d_cls_offset[k] + ch.first, *soff = b->d_cls_scratch_off[k] + 2 * (size_t)ch.first;
                switch (k) {
                case CLS_LIGHT:
                    if (ch.coop) launch_light_sl_level(sk, b->d_W, b->Bp, b->B, b->dp, off, ch.count, b->d_event);  // (coop: the level's straight-line Brillig records)
                    else launch_light_level(sk, b->d_W, b->Bp, b->B, b->dp, off, ch.count, b->d_event);
                    break;
                case CLS_HASH:
                    if (ch.coop) launch_hash_coop_level(sk, b->d_W, b->Bp, b->B, b->dp, off, ch.count, b->d_event, ch.lds_words);
                    else launch_hash_level(sk, b->d_W, b->Bp, b->B, b->dp, off, soff, ch.count, b->d_event, b->d_cls_scratch[k]);
                    break;
                case CLS_GRUMPKIN: launch_grumpkin_level(sk, b->d_W, b->Bp, b->B, b->dp, off, soff, ch.count, b->d_event, b->d_cls_scratch[k]); break;
                case CLS_BRILLIG: launch_brillig_level(sk, b->d_W, b->Bp, b->B, b->dp, off, soff, ch.count, b->d_event, b->d_cls_scratch[k]); break;
                case CLS_PEDERSEN: launch_pedersen_level(sk, b->d_W, b->Bp, b->B, b->dp, off, ch.count, b->d_event); break;
                case CLS_ECDSA: launch_ecdsa_level(sk, b->d_W, b->Bp, b->B, b->dp, off, ch.count, b->d_event); break;
                case CLS_DIGEST: launch_digest_fold_level(sk, b->d_W, b->Bp, b->B, b->dp, off, ch.count, b->fp, b->d_leaves); break;
                case CLS_HOSTBB:  // host callbacks: everything launched so far on any stream must have finished
                    if (last_dyn) HIPCHK(hipStreamWaitEvent(s, last_dyn, 0));
                    for (int q = 0; q < N_HEAVY_LANES; q++)
                        if (last_lane[q]) HIPCHK(hipStreamWaitEvent(s, last_lane[q], 0));
                    for (uint32_t r = 0; r < ch.count; r++)
                        if (int rc = run_host_blackbox(b, p.prog[p.cls_offset[k][ch.first + r] + 1], false, 0)) return rc;
                    break;
                }
                if (prof) { e1 = next_event(); hipEventRecord(e1, sk); tm->cls_pairs[k].push_back({e0, e1}); }
                b->n_launches++;
            }
        if (!one_stream)
            for (int q = 0; q < N_HEAVY_LANES; q++)
                if (lane_used[q]) {
                    HIPCHK(hipEventRecord(b->ev_heavy[4 * L + q], lane_stream[q]));
                    last_lane[q] = b->ev_heavy[4 * L + q];
                }
        main_dirty |= s_work;
        if (nd) {
            if (prev_reg) HIPCHK(hipStreamWaitEvent(s2, prev_reg, 0));
            if (!one_stream)
                for (int q = 0; q < N_HEAVY_LANES; q++)
                    if (p.inv_needs_heavy[q][L + 1]) HIPCHK(hipStreamWaitEvent(s2, b->ev_heavy[4 * (p.inv_needs_heavy[q][L + 1] - 1) + q], 0));
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (prof) { e0 = next_event(); hipEventRecord(e0, s2); }
            launch_inverse_batch(s2, b->d_W, b->d_inv, b->Bp, b->B, b->d_gate_stream, b->d_dyn_offset + p.dyn_level_start[L], nd, b->d_event);
            if (prof) { e1 = next_event(); hipEventRecord(e1, s2); tm->dyn_pairs.push_back({e0, e1}); }
            b->n_launches++;
            HIPCHK(hipEventRecord(b->ev_sync[2 * L + 1], s2));
            last_dyn = b->ev_sync[2 * L + 1];
        }
    }
    for (int q = 0; q < N_HEAVY_LANES; q++)
        if (last_lane[q]) HIPCHK(hipStreamWaitEvent(s, last_lane[q], 0));
    if (last_dyn) HIPCHK(hipStreamWaitEvent(s, last_dyn, 0));
    if (p.truncated_at != 0xFFFFFFFFu) launch_min_u32(s, b->d_event, p.truncated_at, b->B);
    return 0;
}

// the cross-stream events of the schedule exist before anything is enqueued (nothing is created under stream capture)
static int ensure_level_events(acvm_batch *b) {
    const size_t n_levels = b->plan.n_levels;
    while (b->ev_sync.size() < 2 * n_levels + 1 || b->ev_heavy.size() < 4 * n_levels) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        (b->ev_sync.size() < 2 * n_levels + 1 ? b->ev_sync : b->ev_heavy).push_back(e);
    }
    return 0;
}
// The side table of the exact path (slot reuse, and the asynchronous jobs of the node driver): all witnesses x `n_lanes` flagged instances,
// padded to 64 lanes, the memory blocks beside it and -- for a job that runs beside the next tile's level kernels, which use the class
// buffers meanwhile -- per-class scratch of its own. Grow-only. 0 = ready, negative = an error (ACVM_E_DEVICE: no room).
static int ensure_side_table(acvm_batch *b, uint32_t n_lanes, bool own_scratch) {
    const Plan &p = b->plan;
    const uint64_t lanes = ((uint64_t)std::max<uint32_t>(n_lanes, 1) + 63) / 64 * 64;
    if (lanes > b->x_cap) {
        // a table of all witnesses per flagged instance: refuse when that is more than the level table itself
        const size_t need = (size_t)p.n_witnesses * 2 * lanes * sizeof(uint4), level_table = (size_t)(b->reuse() ? p.n_slots : p.n_witnesses) * 2 * b->Bp * sizeof(uint4);
        if (need > level_table && need > (8ull << 30))  // (a small batch pads to 64 lanes either way: below 8 GiB the table is simply allocated)
            return set_err(ACVM_E_UNSUPPORTED, "slot reuse: " + std::to_string(n_lanes) + " instances left the generic path; their own table would exceed "
                                               "the level table -- solve this tile without ACVM_BATCH_REUSE_SLOTS");
        for (void *q : {(void *)b->d_Wx, (void *)b->d_Memx, (void *)b->d_ids_x})
            if (q) hipFree(q);
        b->d_Wx = b->d_Memx = nullptr;
        b->d_ids_x = nullptr;
        b->x_cap = 0;
        HIPCHK(hipMalloc((void **)&b->d_Wx, std::max<size_t>(need, 16)));
        HIPCHK(hipMalloc((void **)&b->d_Memx, std::max<size_t>(16, (size_t)p.mem_cells * 2 * lanes * sizeof(uint4))));
        std::vector<uint32_t> ident(lanes);
        for (uint32_t t = 0; t < lanes; t++) ident[t] = t;
        if (int rc = upload(&b->d_ids_x, ident)) return rc;
        b->x_cap = lanes;
    }
    if (own_scratch && b->x_cap > b->x_scratch_lanes) {
        b->x_scratch_lanes = 0;
        for (int k = 0; k < (int)N_CLS; k++) {
            if (b->d_x_scratch[k]) hipFree(b->d_x_scratch[k]);
            b->d_x_scratch[k] = nullptr;
            if (b->cls_exact_words[k]) HIPCHK(hipMalloc((void **)&b->d_x_scratch[k], (size_t)b->cls_exact_words[k] * b->x_cap * 4));
        }
        b->x_scratch_lanes = b->x_cap;
    }
    return 0;
}

// ACVM::solve for the batch. next_inputs (acvm_batch_solve_then_import): the device buffer of the NEXT tile's initial witnesses, whose import is
// enqueued right behind this solve's event count, gated ON THE DEVICE by that count: it runs only if no instance left the generic path (the
// exact path still needs this tile's rows otherwise). The host then waits for the count alone -- not for the import -- so the next tile's
// level kernels are enqueued while the import runs, and the device does not idle across the tile boundary (0.38 ms of a 23.6 ms tile of the
// metric's workload in round 3: event count, read-back, the caller's loop, import, its synchronisation).
static int batch_solve_impl(acvm_batch *b, const void *next_inputs) {
    if (!b->inputs_set && !b->plan.initial_ids.empty()) return set_err(ACVM_E_STATE, "initial witness not set");
    HIPCHK(hipSetDevice(b->device));
    const Plan &p = b->plan;
    b->next_imported = false;
    b->next_inputs = nullptr;
    if (b->solved) {  // only resolved foreign calls can change anything
        if (b->stepping) return solve_stepping(b, false);
        // A few resumed instances continue on the exact in-order kernels from their Brillig opcode on. When a sizeable part of the
        // batch was answered (the usual case: every instance reaches the same oracle call), the whole LEVEL schedule runs again
        // instead: the answers are in the result store, the Brillig level kernel finds them, and the opcodes behind the call
        // run level-parallel for everybody (instances still waiting, or waiting at the next call, are flagged again there).
        uint32_t n_resolved = 0;
        for (uint32_t t = 0; t < b->slow_ids.size() && t < b->fc_lane.size(); t++)
            n_resolved += b->slow_res[t].status == ACVM_STATUS_REQUIRES_FOREIGN_CALL && b->fc_lane[t].resolved_new;
        const int64_t mode = p.tune.fc_relevel;
        const bool relevel = n_resolved && (mode >= 0 ? mode != 0 : (uint64_t)n_resolved * 16 >= b->B);
        if (!relevel) return solve_resume(b);
        b->solved = false;
    }
    hipStream_t s = b->stream;
    if (int rc = upload_fc_tables(b, 0)) return rc;  // the level kernels read the resolved results too
    b->n_launches = 0;
    b->n_brillig_retries = 0;
    b->host_bb_msg.clear();
    b->arith_kernel_ms = 0;
    b->dyn_kernel_ms = 0;
    b->slow_path_ms = 0;
    for (int k = 0; k < (int)N_CLS; k++) b->cls_kernel_ms[k] = 0;
    LaunchTimers tm;
    auto next_event = [&]() -> hipEvent_t {
        if (tm.ev_used == b->ev_pool.size()) {
            hipEvent_t e;
            hipEventCreate(&e);
            b->ev_pool.push_back(e);
        }
        return b->ev_pool[tm.ev_used++];
    };
    if (!b->force_slow)
        if (int rc = ensure_level_events(b)) return rc;
    HIPCHK(hipEventRecord(b->ev_start, s));
    if (b->force_slow) {
        launch_fill_u32(s, b->d_event, 0u, b->B);
    } else {
        if (int rc = enqueue_level_schedule(b, b->profiling ? &tm : nullptr)) return rc;
    }
    HIPCHK(hipGetLastError());
    // the exact job of the PREVIOUS solve (asynchronous mode) is collected here, while the device works on this solve's levels
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    // instances that left the generic path (or hit a failing opcode): exact in-order re-solve from their event on. Usually there is
    // none: only their count comes back (4 bytes instead of the B event words and a scan of them -- 30 us of a 0.25 ms solve of config 3)
    uint32_t n_flagged = b->B;
    if (!b->force_slow && b->B) {
        *b->h_flag_count = b->B;  // (stays "everything" if the kernel did not run)
        launch_event_count(s, b->d_event, b->B, b->h_flag_count);
        if (next_inputs) {
            if (!b->ev_counted) HIPCHK(hipEventCreate(&b->ev_counted));
            HIPCHK(hipEventRecord(b->ev_counted, s));
            launch_import(s, b->d_W, b->Bp, b->B, (const uint8_t *)next_inputs, b->reuse() ? b->d_init_rows : b->d_init_ids, (uint32_t)p.initial_ids.size(),
                          b->d_event + b->B);  // gate: the count of flagged instances the kernel above left there
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventSynchronize(b->ev_counted));
        } else HIPCHK(hipStreamSynchronize(s));
        n_flagged = *(volatile uint32_t *)b->h_flag_count;
    }
    const bool imported_next = next_inputs && !b->force_slow && b->B && n_flagged == 0;
    if (n_flagged || !b->events_clean) {
        if (n_flagged) {
            HIPCHK(hipMemcpyAsync(b->h_event.data(), b->d_event, (size_t)b->B * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        } else std::fill(b->h_event.begin(), b->h_event.end(), 0xFFFFFFFFu);
        b->slow_ids.clear();
        std::fill(b->slow_index.begin(), b->slow_index.end(), -1);
        for (uint32_t j = 0; j < b->B; j++)
            if (b->h_event[j] != 0xFFFFFFFFu) {
                b->slow_index[j] = (int32_t)b->slow_ids.size();
                b->slow_ids.push_back(j);
            }
        b->events_clean = n_flagged == 0;
    }
    uint32_t n_slow = (uint32_t)b->slow_ids.size();
    hipEvent_t slow0 = nullptr, slow1 = nullptr;
    // asynchronous exact path (node.cpp): a bounded number of flagged instances is re-solved in the side table on stream_x while the
    // caller goes on to the next tile; a batch full of them (a failing circuit, a truncated plan) keeps the synchronous path
    bool go_async = b->async_exact && n_slow && !b->force_slow && (uint64_t)n_slow * 8 <= std::max<uint64_t>(b->B, 512);
    b->side_job = go_async;
    if (n_slow) {
        if (int rc = ensure_slow_capacity(b, n_slow)) return rc;
        HIPCHK(hipMemcpyAsync(b->d_slow_ids, b->slow_ids.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        b->slow_start.resize(n_slow);
        uint32_t min_start = 0xFFFFFFFFu;
        for (uint32_t t = 0; t < n_slow; t++) {
            // slot reuse: the level table no longer holds what ran before the event: the lane starts over from its initial witnesses
            b->slow_start[t] = b->reuse() ? 0u : b->h_event[b->slow_ids[t]];
            min_start = std::min(min_start, b->slow_start[t]);
        }
        HIPCHK(hipMemcpyAsync(b->d_slow_start, b->slow_start.data(), (size_t)n_slow * 4, hipMemcpyHostToDevice, s));
        slow0 = next_event();
        slow1 = next_event();
        hipEventRecord(slow0, s);
        if (b->side()) {
            const int grown = ensure_side_table(b, n_slow, go_async);
            if (grown < 0) {
                // no room for the side table of this many lanes. A handle that does not recycle rows still has every column in the level table:
                // the job runs there, in place, before the caller's next import (the synchronous path); slot reuse has no such fallback
                if (b->reuse()) return grown;
                (void)hipGetLastError();
                go_async = false;
                b->side_job = false;
            }
        }
        if (b->side()) {
            // (on the batch's stream: the level table is read before the next tile's import overwrites it)
            if (b->reuse()) launch_gather_initial(s, b->d_Wx, b->x_cap, b->d_W, b->Bp, b->d_init_ids, b->d_init_rows, (uint32_t)p.initial_ids.size(), b->d_slow_ids, n_slow);
            else {  // the whole column of every flagged instance, plain values: the job resumes at the instance's event like the in-place path
                launch_gather_columns(s, b->d_Wx, b->x_cap, b->d_W, b->Bp, p.n_witnesses, b->d_slow_ids, n_slow, b->d_unscale_index, b->d_unscale_consts);
                launch_gather_columns(s, b->d_Memx, b->x_cap, b->d_Mem, b->Bp, p.mem_cells, b->d_slow_ids, n_slow, nullptr, nullptr);
            }
        } else
        launch_unscale_slow(s, b->d_W, b->Bp, b->d_slow_ids, n_slow, b->unscale);  // the exact kernels work on plain values
        if (go_async) {  // everything below runs on the side stream, behind the gather
            HIPCHK(hipEventRecord(b->ev_x_ready, s));
            HIPCHK(hipStreamWaitEvent(b->stream_x, b->ev_x_ready, 0));
            b->pending = true;
        }
        hipStream_t xs = b->xstream();
        launch_init_assigned(xs, b->d_assigned, n_slow, b->n_words, p.n_witnesses, b->d_producer, b->d_slow_start);
        b->fc_lane.assign(n_slow, acvm_batch::FcLaneState());
        if (int rc = upload_fc_tables(b, n_slow)) return rc;
        launch_exact_init(xs, exact_lanes(b, n_slow));
        if (int rc = run_exact_segments(b, n_slow, min_start)) return rc;
        if (!go_async) {
            HIPCHK(hipStreamSynchronize(s));
            if (int rc = retry_device_limits(b, n_slow, true, 0xFFFFFFFFu)) return rc;
        }
        hipEventRecord(slow1, s);
    }
    float ms = 0;
    if (imported_next && !n_slow) {  // the solve ended at its event count (waited for above); the next tile's import is still in flight
        HIPCHK(hipEventElapsedTime(&ms, b->ev_start, b->ev_counted));
        b->next_imported = true;
        b->next_inputs = next_inputs;
    } else {
        HIPCHK(hipEventRecord(b->ev_end, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipEventElapsedTime(&ms, b->ev_start, b->ev_end));
    }
    b->solve_device_ms = ms;
    auto sum_pairs = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
        double total = 0;
        for (auto &pr : v) {
            float t = 0;
            hipEventElapsedTime(&t, pr.first, pr.second);
            total += t;
        }
        return total;
    };
    b->arith_kernel_ms = sum_pairs(tm.reg_pairs);
    b->dyn_kernel_ms = sum_pairs(tm.dyn_pairs);
    for (int k = 0; k < (int)N_CLS; k++) b->cls_kernel_ms[k] = sum_pairs(tm.cls_pairs[k]);
    if (n_slow) {
        float t = 0;
        hipEventElapsedTime(&t, slow0, slow1);
        b->slow_path_ms = t;
    }
    b->solved = true;
    if (!n_slow) b->slow_res.clear();
    if (b->pending) return (int)n_slow;  // their outcome is not known yet
    return count_not_solved(b);
}
int acvm_batch_solve(acvm_batch_t *b) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    return batch_solve_impl(b, nullptr);
} ABI_CATCH
int acvm_batch_solve_then_import(acvm_batch_t *b, const void *d_next_values_be32) try {
    if (!b) return set_err(ACVM_E_INVALID, "null batch");
    // (resumed foreign calls, stepping and a caller-supplied solver keep the plain solve: nothing is imported behind them)
    const bool plain = !d_next_values_be32 || b->solved || b->stepping || b->has_solver || b->force_slow;
    return batch_solve_impl(b, plain ? nullptr : d_next_values_be32);
} ABI_CATCH

// ---- asynchronous exact path (batch.hpp)
// exact lanes whose side table a handle allocates up front: what a tile of a few diverging inputs needs, bounded by 1 GiB
static uint32_t async_exact_first_lanes(const Plan &p, uint32_t capacity) {
    const uint64_t by_bytes = (1ull << 30) / std::max<uint64_t>(64, (uint64_t)(p.n_witnesses + p.mem_cells) * 32);
    const uint64_t lanes = std::min<uint64_t>(std::min<uint64_t>(1024, std::max<uint64_t>(capacity / 8, 64)), std::max<uint64_t>(by_bytes, 64));
    return (uint32_t)(lanes / 64 * 64);
}
int batch_enable_async_exact(acvm_batch *b, const uint32_t *keep, uint32_t n_keep, bool digests) {
    const Plan &p = b->plan;
    if (b->has_solver || p.has_foreign_calls || p.truncated_at != 0xFFFFFFFFu || !p.tune.exact_async) return 0;
    if (!b->stream_x) HIPCHK(hipStreamCreateWithFlags(&b->stream_x, hipStreamNonBlocking));
    if (!b->ev_x_ready) HIPCHK(hipEventCreateWithFlags(&b->ev_x_ready, hipEventDisableTiming));
    b->async_exact = true;
    b->async_keep.assign(keep, keep + n_keep);
    for (uint32_t &w : b->async_keep)
        if (w >= p.n_witnesses) w = 0xFFFFFFFFu;  // (no row: exports as unassigned)
    b->async_digest = digests;
    // the side table of the first lanes now: a device without room for it says so at creation, not in the middle of a run (a job with more
    // lanes grows it, and falls back to the in-place path when it cannot)
    if (int rc = ensure_side_table(b, async_exact_first_lanes(p, b->capacity), true)) return rc;
    return 1;
}
size_t batch_device_bytes(const Plan &p, const PlanOpts &opts, uint64_t instances, bool async_exact) {
    const uint64_t Bp = (std::max<uint64_t>(instances, 1) + 63) / 64 * 64;
    const uint64_t rows = (opts.reuse_slots ? p.n_slots : p.n_witnesses) + (uint64_t)p.mem_cells + p.n_inverse_slots + p.n_digest_segments;
    size_t bytes = (size_t)rows * 2 * Bp * sizeof(uint4) + (size_t)Bp * 8;
    const uint64_t scratch_cap_words = std::max<uint64_t>(1, (1ull << 30) / (Bp * 4));
    for (int k = 0; k < (int)N_CLS; k++) {  // class scratch: the fattest level of the class (batch_init chunks a level at scratch_cap_words), or its fattest record
        uint64_t need = 0;
        for (size_t L = 0; L + 1 < p.cls_level_start[k].size(); L++) {
            uint64_t used = 0;
            for (uint32_t r = p.cls_level_start[k][L]; r < p.cls_level_start[k][L + 1]; r++) used += p.cls_scratch[k][r];
            need = std::max(need, std::min(used, std::max<uint64_t>(scratch_cap_words, 1)));
        }
        for (uint32_t oi = 0; oi < p.n_opcodes; oi++)
            if (p.prog_class[oi] == (uint32_t)k) need = std::max<uint64_t>(need, p.prog_scratch[oi]);
        bytes += (size_t)need * Bp * 4;
    }
    if (async_exact) bytes += (size_t)async_exact_first_lanes(p, (uint32_t)std::min<uint64_t>(instances, 0xFFFFFFFFu)) * (p.n_witnesses + (uint64_t)p.mem_cells) * 32;
    return bytes;
}

void batch_take_outcome(acvm_batch *b, ExactOutcome *out) {
    *out = std::move(b->last_outcome);
    b->last_outcome.clear();
}
const std::vector<uint32_t> *batch_exact_instances(const acvm_batch *b) { return &b->slow_ids; }
bool batch_exact_pending(const acvm_batch *b) { return b->pending; }
uint32_t batch_exact_unsolved(const acvm_batch *b, uint32_t n) {
    if (b->pending) return 0;  // (their outcome is not known yet)
    uint32_t bad = 0;
    for (size_t t = 0; t < b->slow_ids.size() && t < b->slow_res.size(); t++) bad += b->slow_ids[t] < n && b->slow_res[t].status != ACVM_STATUS_SOLVED;
    return bad;
}
bool batch_generic_assigned(const acvm_batch *b, uint32_t w) { return w < b->plan.n_witnesses && b->plan.producer[w] != 0xFFFFFFFFu; }
int batch_enqueue_kept(acvm_batch *b, uint32_t n, const uint32_t *d_keep, uint32_t n_keep, uint8_t *d_out, uint8_t *h_out, hipStream_t copy_stream,
                       hipEvent_t exported, hipEvent_t arrived) {
    HIPCHK(hipSetDevice(b->device));
    launch_export(b->stream, b->d_W, b->Bp, 0, n, d_keep, n_keep, d_out, b->unscale, b->d_slot_of);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(exported, b->stream));
    HIPCHK(hipStreamWaitEvent(copy_stream, exported, 0));
    HIPCHK(hipMemcpyAsync(h_out, d_out, (size_t)n * n_keep * 32, hipMemcpyDeviceToHost, copy_stream));
    HIPCHK(hipEventRecord(arrived, copy_stream));
    return 0;
}

// digests of lanes [first, first + n) of a witness table into host memory out32 ([n][32]), staged through the arena on stream s:
// arena = (slow_index) | partial sums | digests. The per-instance lane of `assigned` comes from a device array (d_slow_index) or from the
// batch's host vector (use_host_index: uploaded here); neither is needed when u.event is null (every lane read as an instance of the
// level kernels).
static int digest_range(acvm_batch *b, hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, const Unscale &u, const int32_t *d_slow_index,
                        bool use_host_index, uint32_t n_slow, uint8_t *out32) {
    const Plan &p = b->plan;
    if (!n) return 0;
    if (int rc = ensure_digest_tables(b)) return rc;
    const size_t idx_bytes = use_host_index ? align256((size_t)b->B * 4) : 0;
    const size_t part_bytes = align256((size_t)digest_chunks(p.n_witnesses) * n * 32);
    if (int rc = stage_reserve(b, idx_bytes + part_bytes + (size_t)n * 32)) return rc;
    if (use_host_index) {
        HIPCHK(hipMemcpyAsync(b->d_stage, b->slow_index.data(), (size_t)b->B * 4, hipMemcpyHostToDevice, s));
        d_slow_index = (const int32_t *)b->d_stage;
    }
    uint4 *d_part = (uint4 *)(b->d_stage + idx_bytes);
    uint8_t *d_out = b->d_stage + idx_bytes + part_bytes;
    launch_digest(s, W, Bp, first, n, p.n_witnesses, b->d_producer, u, b->fp, d_slow_index, b->d_assigned, n_slow, d_part, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out32, d_out, (size_t)n * 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// results, kept witnesses and digests of the lanes of the side table (all of them at once)
static int side_table_outcome(acvm_batch *b, ExactOutcome *out) {
    const Plan &p = b->plan;
    hipStream_t s = b->xstream();
    const uint32_t n_slow = (uint32_t)b->slow_ids.size(), n_keep = (uint32_t)b->async_keep.size();
    out->instance = b->slow_ids;
    out->results.resize(n_slow);
    for (uint32_t t = 0; t < n_slow; t++) {
        acvm_result_t &r = out->results[t];
        memset(&r, 0, sizeof r);
        const SlowResult &sr = b->slow_res[t];
        r.status = sr.status; r.err = sr.err; r.opcode_index = sr.opcode_index; r.aux0 = sr.aux0; r.aux1 = sr.aux1;
        r.n_call_stack = sr.n_call_stack > 16 ? 16 : sr.n_call_stack;
        for (uint32_t k = 0; k < r.n_call_stack; k++) r.call_stack[k] = sr.call_stack[k];
    }
    const size_t sel_bytes = align256((size_t)std::max<uint32_t>(n_keep, 1) * 4), val_bytes = align256((size_t)n_slow * std::max<uint32_t>(n_keep, 1) * 32);
    if (int rc = stage_reserve(b, sel_bytes + val_bytes)) return rc;
    Unscale plain = b->unscale;
    plain.event = b->d_slow_start;  // (opcode indices, never 0xFFFFFFFF = "solved by the level kernels": nothing in the side table is scaled)
    if (n_keep) {
        uint32_t *d_sel = (uint32_t *)b->d_stage;
        uint8_t *d_val = b->d_stage + sel_bytes;
        HIPCHK(hipMemcpyAsync(d_sel, b->async_keep.data(), (size_t)n_keep * 4, hipMemcpyHostToDevice, s));
        launch_export(s, b->d_Wx, b->x_cap, 0, n_slow, d_sel, n_keep, d_val, plain);
        out->kept_values.resize((size_t)n_slow * n_keep * 32);
        HIPCHK(hipMemcpyAsync(out->kept_values.data(), d_val, out->kept_values.size(), hipMemcpyDeviceToHost, s));
        std::vector<uint32_t> bitmap((size_t)n_slow * b->n_words);
        HIPCHK(hipMemcpyAsync(bitmap.data(), b->d_assigned, bitmap.size() * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        out->kept_assigned.resize((size_t)n_slow * n_keep);
        for (uint32_t t = 0; t < n_slow; t++)
            for (uint32_t k = 0; k < n_keep; k++) {
                const uint32_t w = b->async_keep[k];
                const bool a = w < p.n_witnesses && ((bitmap[(size_t)(w >> 5) * n_slow + t] >> (w & 31)) & 1u);
                out->kept_assigned[(size_t)t * n_keep + k] = a;
                if (!a) memset(&out->kept_values[((size_t)t * n_keep + k) * 32], 0, 32);
            }
    }
    if (b->async_digest) {
        out->digests.resize((size_t)n_slow * 32);
        if (int rc = digest_range(b, s, b->d_Wx, b->x_cap, 0, n_slow, plain, (const int32_t *)b->d_ids_x, false, n_slow, out->digests.data())) return rc;
    }
    return 0;
}

int batch_set_live_count(acvm_batch *b, uint32_t n) {
    if (!b || !n || n > b->capacity) return set_err(ACVM_E_INVALID, "live count out of range");
    if (n != b->B) {
        b->B = n;
        b->inputs_set = false;
        b->solved = false;
        b->stepping = false;
    }
    return 0;
}
int batch_finish_pending(acvm_batch *b, ExactOutcome *out) {
    if (out) out->clear();
    if (!b->pending) return 0;
    HIPCHK(hipSetDevice(b->device));
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    b->slow_res.resize(n_slow);
    HIPCHK(hipMemcpyAsync(b->slow_res.data(), b->d_slow_res, (size_t)n_slow * sizeof(SlowResult), hipMemcpyDeviceToHost, b->stream_x));
    HIPCHK(hipStreamSynchronize(b->stream_x));
    int rc = retry_device_limits(b, n_slow, true, 0xFFFFFFFFu);
    if (!rc && out) {
        rc = side_table_outcome(b, out);
        for (uint32_t t = 0; t < n_slow && !rc; t++)  // message texts
            if (b->slow_res[t].status == ACVM_STATUS_FAILURE && b->slow_res[t].msg) format_message(b, b->slow_ids[t], b->slow_res[t], out->results[t]);
    }
    b->pending = false;
    return rc;
}

int batch_export_tile(acvm_batch *b, uint32_t n, const uint32_t *keep, uint32_t n_keep, acvm_result_t *results, uint8_t *kept_values, uint8_t *kept_assigned,
                      uint8_t *digests) {
    const Plan &p = b->plan;
    if (!b->solved || n > b->B) return set_err(ACVM_E_STATE, "batch not solved");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = b->stream;
    const bool defer = b->pending;  // the instances of the exact path arrive with the job's outcome
    if (results)
        for (uint32_t j = 0; j < n; j++)
            if (!(defer && b->slow_index[j] >= 0)) fill_result(b, j, results[j]);
    if (n_keep && kept_values) {
        const size_t sel_bytes = align256((size_t)n_keep * 4);
        if (int rc = stage_reserve(b, sel_bytes + (size_t)n * n_keep * 32)) return rc;
        uint32_t *d_sel = (uint32_t *)b->d_stage;
        uint8_t *d_out = b->d_stage + sel_bytes;
        HIPCHK(hipMemcpyAsync(d_sel, keep, (size_t)n_keep * 4, hipMemcpyHostToDevice, s));
        launch_export(s, b->d_W, b->Bp, 0, n, d_sel, n_keep, d_out, b->unscale, b->d_slot_of);
        HIPCHK(hipMemcpyAsync(kept_values, d_out, (size_t)n * n_keep * 32, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (uint32_t k = 0; k < n_keep; k++) {
            const bool produced = keep[k] < p.n_witnesses && p.producer[keep[k]] != 0xFFFFFFFFu;
            for (uint32_t j = 0; j < n; j++)
                if (b->slow_index[j] < 0) {
                    if (kept_assigned) kept_assigned[(size_t)j * n_keep + k] = produced;
                    if (!produced) memset(kept_values + ((size_t)j * n_keep + k) * 32, 0, 32);
                }
        }
    }
    if (digests) {
        if (defer || b->slow_ids.empty()) {
            // (the table-wide kernels: flagged columns hold leftovers and are overwritten by the outcome)
            if (p.n_digest_segments && b->d_leaves) {
                if (int rc = stage_reserve(b, (size_t)n * 32)) return rc;
                launch_digest_final(s, b->d_leaves, p.n_digest_segments, b->Bp, 0, n, nullptr, b->fp, b->d_stage);
                HIPCHK(hipMemcpyAsync(digests, b->d_stage, (size_t)n * 32, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
            } else {
                // every lane is read as a generic instance here: an event word that is set would send the kernel to the assigned bitmap of a
                // job that is still running
                Unscale u = b->unscale;
                u.event = nullptr;
                if (int rc = digest_range(b, s, b->d_W, b->Bp, 0, n, u, nullptr, false, 0, digests)) return rc;
            }
        } else if (int rc = acvm_batch_digest(b, 0, n, digests)) return rc;
    }
    if (!defer && !b->slow_ids.empty() && n_keep && kept_values) {  // synchronous exact lanes: their values from where they live
        std::vector<uint8_t> one((size_t)b->B * 32), asg(b->B);
        for (uint32_t k = 0; k < n_keep; k++) {
            bool any = false;
            for (uint32_t j = 0; j < n; j++) any |= b->slow_index[j] >= 0;
            if (!any) break;
            if (int rc = acvm_batch_witness(b, keep[k], one.data(), asg.data())) return rc;
            for (uint32_t j = 0; j < n; j++)
                if (b->slow_index[j] >= 0) {
                    memcpy(kept_values + ((size_t)j * n_keep + k) * 32, &one[(size_t)j * 32], 32);
                    if (kept_assigned) kept_assigned[(size_t)j * n_keep + k] = asg[j];
                }
        }
    }
    return 0;
}

// ---- ACVM::get_pending_foreign_call / resolve_pending_foreign_call (pwg/mod.rs:203-228) per instance
static int fetch_pending(acvm_batch *b) {
    if (b->pend_host_valid) return 0;
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    b->h_pend_desc.assign((size_t)b->fc_pend_desc_words * n_slow, 0);
    b->h_pend_vals.assign((size_t)b->fc_pend_vals_cap * 2 * n_slow * 4, 0);
    if (n_slow && b->d_fc_pend_desc) {
        HIPCHK(hipMemcpy(b->h_pend_desc.data(), b->d_fc_pend_desc, b->h_pend_desc.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b->h_pend_vals.data(), b->d_fc_pend_vals, b->h_pend_vals.size() * 4, hipMemcpyDeviceToHost));
    }
    b->pend_host_valid = true;
    return 0;
}
static int waiting_lane(acvm_batch *b, uint32_t instance) {
    if (!b->solved || instance >= b->B) return -1;
    int32_t t = b->slow_index[instance];
    if (t < 0 || b->slow_res[t].status != ACVM_STATUS_REQUIRES_FOREIGN_CALL) return -1;
    return t;
}

int acvm_batch_pending_foreign_call(acvm_batch_t *b, uint32_t instance, acvm_foreign_call_info_t *info) try {
    if (!b || !info) return set_err(ACVM_E_INVALID, "null argument");
    memset(info, 0, sizeof *info);
    int t = waiting_lane(b, instance);
    if (t < 0) return 0;
    HIPCHK(hipSetDevice(b->device));
    if (int rc = fetch_pending(b)) return rc;
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    const SlowResult &sr = b->slow_res[t];
    info->opcode_index = sr.opcode_index;
    info->brillig_index = sr.x0;
    info->n_inputs = b->h_pend_desc[t];
    for (uint32_t i = 0; i < info->n_inputs; i++) info->n_values += b->h_pend_desc[(size_t)(1 + i) * n_slow + t];
    auto it = b->plan.fc_function.find(((uint64_t)sr.opcode_index << 32) | sr.x0);
    snprintf(info->function, sizeof info->function, "%s", it == b->plan.fc_function.end() ? "" : it->second.c_str());
    return 1;
} ABI_CATCH

int acvm_batch_pending_foreign_call_inputs(acvm_batch_t *b, uint32_t instance, uint32_t *lens, uint8_t *values_be32) try {
    if (!b || !lens || !values_be32) return set_err(ACVM_E_INVALID, "null argument");
    int t = waiting_lane(b, instance);
    if (t < 0) return set_err(ACVM_E_STATE, "instance is not waiting for a foreign call");
    HIPCHK(hipSetDevice(b->device));
    if (int rc = fetch_pending(b)) return rc;
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    const uint32_t n_in = b->h_pend_desc[t];
    uint32_t vi = 0;
    for (uint32_t i = 0; i < n_in; i++) {
        lens[i] = b->h_pend_desc[(size_t)(1 + i) * n_slow + t];
        for (uint32_t c = 0; c < lens[i]; c++, vi++) {
            FrH m;
            memcpy(&m.l[0], &b->h_pend_vals[((size_t)(2 * vi) * n_slow + t) * 4], 16);
            memcpy(&m.l[2], &b->h_pend_vals[((size_t)(2 * vi + 1) * n_slow + t) * 4], 16);
            uint64_t can[4];
            frh::to_canonical(frh::from_device_form(m), can);
            for (int k = 0; k < 32; k++) values_be32[(size_t)vi * 32 + 31 - k] = (uint8_t)(can[k / 8] >> (8 * (k % 8)));
        }
    }
    return 0;
} ABI_CATCH

int acvm_batch_resolve_foreign_call(acvm_batch_t *b, uint32_t instance, uint32_t n_values, const uint8_t *is_array, const uint32_t *lens,
                                    const uint8_t *values_be32) try {
    if (!b || (n_values && (!is_array || !lens || !values_be32))) return set_err(ACVM_E_INVALID, "null argument");
    int t = waiting_lane(b, instance);
    if (t < 0) return set_err(ACVM_E_STATE, "ACVM is not expecting a foreign call response as no call was made");  // mod.rs:215-217 panics
    auto &ls = b->fc_lane[t];
    const uint32_t opcode = b->slow_res[t].opcode_index;
    if (ls.resolved_new) return set_err(ACVM_E_STATE, "this instance's pending foreign call was already resolved; call acvm_batch_solve");
    const auto &slots = b->plan.fc_slot_opcode;
    const size_t si = (size_t)(std::find(slots.begin(), slots.end(), opcode) - slots.begin());
    if (si >= b->fc_slots.size()) return set_err(ACVM_E_STATE, "the instance does not wait at a Brillig opcode with a foreign call");
    std::vector<acvm_batch::FcValue> res(n_values);
    size_t off = 0;
    for (uint32_t i = 0; i < n_values; i++) {
        res[i].is_array = is_array[i] != 0;
        uint32_t n = res[i].is_array ? lens[i] : 1;
        for (uint32_t c = 0; c < n; c++, off++) res[i].vals.push_back(frh::from_be_bytes32_reduce(values_be32 + off * 32, 32));
    }
    // results accumulate per Brillig opcode and instance (brillig.foreign_call_results.push, mod.rs:223)
    b->fc_slots[si].inst[instance].push_back(std::move(res));
    b->fc_slots[si].dirty = true;
    ls.resolved_new = true;
    return 0;
} ABI_CATCH

// after acvm_batch_solve_then_import the rows of the INITIAL witnesses hold the next tile's values: whatever reads them back is refused
static int refuse_if_next_imported(const acvm_batch *b, const uint32_t *ws, uint32_t n, bool whole_map) {
    if (!b->next_imported) return 0;
    bool hit = whole_map;
    for (uint32_t k = 0; k < n && !hit; k++) hit = std::find(b->plan.initial_ids.begin(), b->plan.initial_ids.end(), ws[k]) != b->plan.initial_ids.end();
    if (!hit) return 0;
    return set_err(ACVM_E_STATE, "the initial witnesses of this solve are gone: acvm_batch_solve_then_import put the next tile's inputs into the table behind the solve "
                                 "(read results, non-initial witnesses and nothing else; or use acvm_batch_solve)");
}

// ---- slot reuse (ACVM_BATCH_REUSE_SLOTS): what can be read back
static bool reuse_kept(const acvm_batch *b, uint32_t w) {
    const Plan &p = b->plan;
    if (w >= p.n_witnesses) return false;
    if (std::find(p.initial_ids.begin(), p.initial_ids.end(), w) != p.initial_ids.end()) return true;
    return std::find(b->opts.keep.begin(), b->opts.keep.end(), w) != b->opts.keep.end();
}
static int reuse_check_kept(const acvm_batch *b, const uint32_t *ws, uint32_t n) {
    if (!b->reuse()) return 0;
    for (uint32_t k = 0; k < n; k++)
        if (ws[k] < b->plan.n_witnesses && !reuse_kept(b, ws[k]))
            return set_err(ACVM_E_STATE, "witness " + std::to_string(ws[k]) + " was not kept: the batch recycles witness rows (ACVM_BATCH_REUSE_SLOTS); "
                                         "only the initial witnesses and keep_ids can be read back");
    return 0;
}
// the instances of the exact path have their values in the table of their own: overwrite their rows of an export
// (values_be32 [n][n_sel][32] of instances [first, first + n), d_sel = the witness list already on the device)
static int reuse_patch_exact(acvm_batch *b, const uint32_t *d_sel, uint32_t n_sel, uint32_t first, uint32_t n, uint8_t *values_be32, uint8_t *d_tmp) {
    if (!b->side()) return 0;
    Unscale plain = b->unscale;
    plain.event = b->d_slow_start;  // opcode indices, never 0xFFFFFFFF: "not the generic instance", nothing is scaled in the exact table
    for (uint32_t i = 0; i < n; i++) {
        const int32_t t = b->slow_index[first + i];
        if (t < 0) continue;
        launch_export(b->stream, b->d_Wx, b->x_cap, (uint32_t)t, 1, d_sel, n_sel, d_tmp, plain);
        HIPCHK(hipMemcpyAsync(values_be32 + (size_t)i * n_sel * 32, d_tmp, (size_t)n_sel * 32, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    return 0;
}

// one witness of one instance as 32 canonical big-endian bytes (message texts only; rare)
static bool fetch_one(acvm_batch *b, uint32_t j, uint32_t w, uint8_t out[32]) {
    // While an exact job is pending its lanes live in the side table and the caller's NEXT tile may already be enqueued on the handle's
    // stream: the fetch goes through the job's stream (and a staging slot of its own), so that a failing instance's message does not wait
    // for a whole level schedule.
    const bool side_lane = b->side() && b->slow_index[j] >= 0;
    hipStream_t s = b->pending && side_lane ? b->stream_x : b->stream;
    if (!b->d_fetch && hipMalloc((void **)&b->d_fetch, 512) != hipSuccess) return false;
    uint32_t *d_sel = (uint32_t *)b->d_fetch;
    uint8_t *d_out = b->d_fetch + 256;
    if (hipMemcpyAsync(d_sel, &w, 4, hipMemcpyHostToDevice, s) != hipSuccess) return false;
    if (hipStreamSynchronize(s) != hipSuccess) return false;  // &w is a stack address
    if (side_lane) {
        Unscale plain = b->unscale;
        plain.event = b->d_slow_start;
        launch_export(s, b->d_Wx, b->x_cap, (uint32_t)b->slow_index[j], 1, d_sel, 1, d_out, plain);
    } else
    launch_export(s, b->d_W, b->Bp, j, 1, d_sel, 1, d_out, b->unscale, b->d_slot_of);
    return hipMemcpyAsync(out, d_out, 32, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
}

// message text of a failure, rebuilt from the device's DevMsg code (ops_common.hpp) + payload
static void format_message(acvm_batch *b, uint32_t j, const SlowResult &sr, acvm_result_t &r) {
    const Plan &p = b->plan;
    const uint32_t *rec = sr.opcode_index < p.n_opcodes ? &p.prog[p.prog_offset[sr.opcode_index]] : nullptr;
    char hx[65];
    switch (sr.msg) {
    case 1: snprintf(r.message, sizeof r.message, "Mul term in the arithmetic opcode must contain either zero or one term"); break;
    case 2: snprintf(r.message, sizeof r.message, "number of bits specified for each input must be the same"); break;
    case 3: snprintf(r.message, sizeof r.message, "fetch_nearest_bytes: range end index out of range"); break;
    case 4: snprintf(r.message, sizeof r.message, "Expected 32 outputs but encountered %u", sr.x0); break;
    case 5: {
        unsigned long long len = 0;
        if (rec && rec[0] == PK_HASH)
            for (uint32_t i = 0; i < rec[3]; i++) len += (rec[6 + 2 * i + 1] + 7) / 8;
        snprintf(r.message, sizeof r.message,
                 "the number of bytes to take from the message is more than the number of bytes in the message. %llu > %llu",
                 (unsigned long long)sr.x1 << 32 | sr.x0, len);
        break;
    }
    case 6: snprintf(r.message, sizeof r.message, "called `Option::unwrap()` on a `None` value (memory index)"); break;
    case 7: snprintf(r.message, sizeof r.message, "Memory must be read into a specified witness index, encountered an Expression"); break;
    case 8: snprintf(r.message, sizeof r.message, "The radix must be within 2...256"); break;
    case 9: case 10: case 11: {
        // the offending value: a witness of the FixedBaseScalarMul opcode, or the VM register value the device quoted
        const bool in_brillig = sr.err == ACVM_ERR_BRILLIG_FAILED;
        uint8_t val[32] = {0};
        if (in_brillig) {
            for (int i = 0; i < 32; i++) val[31 - i] = (uint8_t)(sr.val[i / 4] >> (8 * (i % 4)));
        } else if (rec && rec[0] == PK_FIXED_BASE) {
            if (sr.msg == 11) {
                uint8_t lo[32] = {0}, hi[32] = {0};
                fetch_one(b, j, rec[2], lo);
                fetch_one(b, j, rec[3], hi);
                memcpy(val, hi + 16, 16);
                memcpy(val + 16, lo + 16, 16);
            } else fetch_one(b, j, rec[sr.msg == 9 ? 2 : 3], val);
        }
        char reason[160];
        if (sr.msg == 11) {  // hex::encode(BigUint::to_bytes_be()) of high * 2^128 + low: minimal big-endian bytes
            int st = 0;
            while (st < 31 && val[st] == 0) st++;
            char hexs[65];
            for (int i = st; i < 32; i++) snprintf(hexs + 2 * (i - st), 3, "%02x", val[i]);
            snprintf(reason, sizeof reason, "Value %s is not a valid grumpkin scalar", hexs);
        } else {
            for (int i = 0; i < 32; i++) snprintf(hx + 2 * i, 3, "%02x", val[i]);
            snprintf(reason, sizeof reason, "Limb %s is not less than 2^128", hx);
        }
        if (in_brillig) snprintf(r.message, sizeof r.message, "failed to solve blackbox function: fixed_base_scalar_mul, reason: %s", reason);
        else snprintf(r.message, sizeof r.message, "%s", reason);
        break;
    }
    case 12: snprintf(r.message, sizeof r.message, "range end index 64 out of range for slice of length %u", sr.x0); break;
    case 13: snprintf(r.message, sizeof r.message, "Message overran wasm scratch space"); break;
    case 14: snprintf(r.message, sizeof r.message, "explicit trap hit in brillig"); break;
    case 15: snprintf(r.message, sizeof r.message, "return opcode hit, but callstack already empty"); break;
    case 16: {
        static const char *texts[17] = {"", "Reading register past maximum!", "Writing register past maximum!", "register does not fit into u64",
                                        "memory read out of range", "", "bit_size > 256 is not supported", "attempt to subtract with overflow",
                                        "attempt to divide by zero", "unsupported bit size for right shift",
                                        "called `Option::unwrap()` on a `None` value", "bad int op", "index out of bounds: bytecode",
                                        "bad brillig opcode", "", "index out of bounds: brillig memory", "bad black box op"};
        if (sr.x0 == 100) snprintf(r.message, sizeof r.message, "range end index 64 out of range for slice of length %u", sr.x1);
        else if (sr.x0 == 101) snprintf(r.message, sizeof r.message, "Message overran wasm scratch space");
        else if (sr.x0 == 102) snprintf(r.message, sizeof r.message, "Function result size does not match brillig bytecode (expected 1 result)");
        else if (sr.x0 == 103) snprintf(r.message, sizeof r.message, "Function result size does not match brillig bytecode size");
        else if (sr.x0 > 110 && sr.x0 < 117) {
            static const char *et[7] = {"", "ecdsa: signature scalars must be in [1, n-1] (Signature::try_from unwrap)",
                                        "ecdsa: public key x is not on the curve (PublicKey::from_encoded_point unwrap)",
                                        "ecdsa: hashed message must be 32 bytes (GenericArray::from_slice)",
                                        "ecdsa: hashed message is not below the group order (Scalar::from_repr unwrap)",
                                        "ecdsa: R is the identity (unreachable!)", "ecdsa: R.x is not below the group order (Scalar::from_repr unwrap)"};
            snprintf(r.message, sizeof r.message, "%s", et[sr.x0 - 110]);
        }
        else snprintf(r.message, sizeof r.message, "%s", sr.x0 < 17 ? texts[sr.x0] : "brillig vm panic");
        break;
    }
    // 17 / 18 / 28: device limits of the Brillig VM. retry_device_limits retries such lanes or ends them with ACVM_ERR_DEVICE_LIMIT (29); the texts are for debugging only
    case 17: snprintf(r.message, sizeof r.message, "brillig memory write at %u beyond the device capacity", sr.x0); break;
    case 18: snprintf(r.message, sizeof r.message, "brillig step limit reached on the device"); break;
    case 28: snprintf(r.message, sizeof r.message, "brillig call depth limit reached on the device"); break;
    case 19: {
        static const char *what[3] = {"Invalid public key x length", "Invalid public key y length", "Invalid signature length"};
        snprintf(r.message, sizeof r.message, "failed to solve blackbox function: %s, reason: %s", sr.x0 / 4 ? "ecdsa_secp256r1" : "ecdsa_secp256k1",
                 what[sr.x0 % 4 < 3 ? sr.x0 % 4 : 0]);
        break;
    }
    case 20: snprintf(r.message, sizeof r.message, "failed to solve blackbox function: pedersen, reason: Invalid signature length"); break;
    case 21: snprintf(r.message, sizeof r.message, "%u output values were provided as a foreign call result for %u destination slots", sr.x0, sr.val[0]); break;
    case 22: snprintf(r.message, sizeof r.message, "Function result size does not match brillig bytecode"); break;
    case 23: snprintf(r.message, sizeof r.message, "foreign call inputs exceed the device staging buffer"); break;
    case 25: {
        static const char *what[3] = {"pubkey_x", "pubkey_y", "signature"};
        snprintf(r.message, sizeof r.message, "expected %s size %u but received %u", what[sr.x0 < 3 ? sr.x0 : 0], sr.x0 == 2 ? 64u : 32u, sr.x1);
        break;
    }
    case 26: {
        static const char *texts[7] = {"", "ecdsa: signature scalars must be in [1, n-1] (Signature::try_from unwrap)",
                                       "ecdsa: public key x is not on the curve (PublicKey::from_encoded_point unwrap)",
                                       "ecdsa: hashed message must be 32 bytes (GenericArray::from_slice)",
                                       "ecdsa: hashed message is not below the group order (Scalar::from_repr unwrap)",
                                       "ecdsa: R is the identity (unreachable!)", "ecdsa: R.x is not below the group order (Scalar::from_repr unwrap)"};
        snprintf(r.message, sizeof r.message, "%s", sr.x0 < 7 ? texts[sr.x0] : "");
        break;
    }
    case 27: snprintf(r.message, sizeof r.message, "index out of bounds: the len is %u but the index is %u", sr.x0, sr.x1); break;
    case 29: {  // ACVM_ERR_DEVICE_LIMIT: not a reference outcome (include/acvm_amd.h)
        static const char *what[5] = {"", "VM steps", "nested calls", "cells of VM memory", "MiB of VM scratch on the device"};
        const uint32_t k = sr.aux0 < 5 ? sr.aux0 : 0;
        if (k == ACVM_LIMIT_BRILLIG_MEMORY)
            snprintf(r.message, sizeof r.message, "device limit: the Brillig program writes VM memory cell %u, beyond the %u cells this library runs it with; "
                                                  "the reference has no such limit: solve this instance with it", sr.x0, sr.aux1);
        else
            snprintf(r.message, sizeof r.message, "device limit: the Brillig program needs more than %u %s; the reference has no such limit: solve this "
                                                  "instance with it", sr.aux1, what[k]);
        break;
    }
    case 24: {
        auto it = b->host_bb_msg.find(j);
        snprintf(r.message, sizeof r.message, "%s", it == b->host_bb_msg.end() ? "" : it->second.c_str());
        break;
    }
    default: break;
    }
}

static void fill_result(acvm_batch *b, uint32_t j, acvm_result_t &r) {
    memset(&r, 0, sizeof r);
    if (b->plan.n_opcodes == 0 || b->slow_index[j] < 0) { r.status = ACVM_STATUS_SOLVED; return; }
    if (b->pending) { r.status = ACVM_STATUS_IN_PROGRESS; return; }  // its exact job is still running (batch_finish_pending)
    const SlowResult &sr = b->slow_res[b->slow_index[j]];
    r.status = sr.status; r.err = sr.err; r.opcode_index = sr.opcode_index; r.aux0 = sr.aux0; r.aux1 = sr.aux1;
    r.n_call_stack = sr.n_call_stack > 16 ? 16 : sr.n_call_stack;
    for (uint32_t k = 0; k < r.n_call_stack; k++) r.call_stack[k] = sr.call_stack[k];
    if (sr.status == ACVM_STATUS_FAILURE && sr.msg) format_message(b, j, sr, r);
}

int acvm_batch_results(acvm_batch_t *b, acvm_result_t *out) try {
    if (!b || !out) return set_err(ACVM_E_INVALID, "null argument");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    HIPCHK(hipSetDevice(b->device));
    for (uint32_t j = 0; j < b->B; j++) fill_result(b, j, out[j]);
    return 0;
} ABI_CATCH

// ---------------------------------------------------------------------------------------------- after solve
int acvm_circuit_assert_message(const acvm_circuit_t *c, uint32_t acir_index, uint32_t brillig_index, char *out, size_t cap) {
    if (!c) return set_err(ACVM_E_INVALID, "null argument");
    const bool want_brillig = brillig_index != ACVM_LOCATION_ACIR;
    for (const AssertMessage &m : c->c->assert_messages) {  // first match, like the reference's linear find
        if (m.is_brillig != want_brillig || m.acir_index != acir_index || (want_brillig && m.brillig_index != brillig_index)) continue;
        if (out && cap) snprintf(out, cap, "%s", m.message.c_str());
        return (int)m.message.size();
    }
    if (out && cap) out[0] = 0;
    return -1;
}

int acvm_circuit_witness_set(const acvm_circuit_t *c, int which, uint32_t *out, uint32_t cap) try {
    if (!c) return set_err(ACVM_E_INVALID, "null argument");
    const Circuit &k = *c->c;
    std::vector<uint32_t> v;
    switch (which) {
    case ACVM_SET_PRIVATE_PARAMETERS: v = k.private_parameters; break;
    case ACVM_SET_PUBLIC_PARAMETERS: v = k.public_parameters; break;
    case ACVM_SET_RETURN_VALUES: v = k.return_values; break;
    case ACVM_SET_PUBLIC_INPUTS: v = k.public_parameters; v.insert(v.end(), k.return_values.begin(), k.return_values.end()); break;
    case ACVM_SET_CIRCUIT_ARGUMENTS: v = k.private_parameters; v.insert(v.end(), k.public_parameters.begin(), k.public_parameters.end()); break;
    default: return set_err(ACVM_E_INVALID, "unknown witness set");
    }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (uint32_t i = 0; i < v.size() && i < cap && out; i++) out[i] = v[i];
    return (int)v.size();
} ABI_CATCH

// The expression OpcodeNotSolvable::ExpressionHasTooManyUnknowns carries for `instance` (pwg/mod.rs:72-78): the opcode partially evaluated
// on the instance's map for Opcode::Arithmetic (arithmetic.rs:31,38-42), the first input expression that does not reduce to a constant, as
// written, for Opcode::Brillig (brillig.rs:46-74, get_value pwg/mod.rs:321-332). Witnesses the instance has assigned are read back one by
// one (rare path: one failing instance). false: the opcode carries no such expression.
static bool too_many_unknowns_expr(acvm_batch *b, const Circuit &circ, uint32_t instance, uint32_t opcode_index, Expr &out) {
    if (opcode_index >= circ.opcodes.size()) return false;
    const int32_t lane = b->slow_index[instance];
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    auto known = [&](uint32_t w) -> bool {
        if (w >= b->plan.n_witnesses) return false;
        if (lane < 0) return b->plan.producer[w] != 0xFFFFFFFFu;
        uint32_t bitsw = 0;
        if (hipMemcpy(&bitsw, b->d_assigned + (size_t)(w >> 5) * n_slow + lane, 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        return (bitsw >> (w & 31)) & 1u;
    };
    auto value = [&](uint32_t w) {
        uint8_t be[32] = {0};
        fetch_one(b, instance, w, be);
        return frh::from_be_bytes32_reduce(be, 32);
    };
    // ArithmeticSolver::evaluate (arithmetic.rs:212-239)
    auto evaluate = [&](const Expr &e) {
        Expr r;
        for (const MulTerm &t : e.mul) {
            const bool kl = known(t.l), kr = known(t.r);
            if (kl && kr) r.qc = frh::add(r.qc, frh::mul(frh::mul(t.c, value(t.l)), value(t.r)));
            else if (!kl && !kr) { if (!t.c.is_zero()) r.mul.push_back(t); }
            else {
                const FrH v = frh::mul(t.c, value(kl ? t.l : t.r));
                if (!v.is_zero()) r.lin.push_back({v, kl ? t.r : t.l});
            }
        }
        for (const LinTerm &t : e.lin) {
            if (known(t.w)) r.qc = frh::add(r.qc, frh::mul(t.c, value(t.w)));
            else if (!t.c.is_zero()) r.lin.push_back(t);
        }
        r.qc = frh::add(r.qc, e.qc);
        return r;
    };
    const Opcode &o = circ.opcodes[opcode_index];
    if (o.kind == OP_ARITHMETIC) { out = evaluate(o.expr); return true; }
    if (o.kind == OP_BRILLIG) {  // the first input, in order, that does not reduce to a constant (get_value, pwg/mod.rs:321-332)
        auto stuck = [&](const Expr &e) { const Expr r = evaluate(e); return !r.mul.empty() || !r.lin.empty(); };
        for (const BrilligInput &in : o.brillig->inputs) {
            if (!in.is_array) { if (stuck(in.single)) { out = in.single; return true; } }
            else for (const Expr &e : in.arr) if (stuck(e)) { out = e; return true; }
        }
    }
    return false;
}
static std::string too_many_unknowns_expression(acvm_batch *b, const Circuit &circ, uint32_t instance, uint32_t opcode_index) {
    Expr e;
    return too_many_unknowns_expr(b, circ, instance, opcode_index, e) ? expression_display(e) : std::string();
}

int acvm_batch_error_expression(acvm_batch_t *b, const acvm_circuit_t *c, uint32_t instance, acvm_expression_t *head, uint8_t *mul_coef_be32,
                                uint32_t *mul_witnesses, uint32_t cap_mul, uint8_t *lin_coef_be32, uint32_t *lin_witnesses, uint32_t cap_lin) try {
    if (!b || !c || !head) return set_err(ACVM_E_INVALID, "null argument");
    memset(head, 0, sizeof *head);
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if (instance >= b->B) return set_err(ACVM_E_INVALID, "instance out of range");
    HIPCHK(hipSetDevice(b->device));
    acvm_result_t r;
    fill_result(b, instance, r);
    if (r.status != ACVM_STATUS_FAILURE || r.err != ACVM_ERR_TOO_MANY_UNKNOWNS) return 0;
    Expr e;
    if (!too_many_unknowns_expr(b, *c->c, instance, r.opcode_index, e)) return 0;
    auto put_be = [](uint8_t *dst, const FrH &x) {
        uint64_t can[4];
        frh::to_canonical(x, can);
        for (int k = 0; k < 32; k++) dst[31 - k] = (uint8_t)(can[k / 8] >> (8 * (k % 8)));
    };
    head->n_mul = (uint32_t)e.mul.size();
    head->n_lin = (uint32_t)e.lin.size();
    head->opcode_index = r.opcode_index;
    put_be(head->q_c, e.qc);
    for (uint32_t i = 0; i < head->n_mul && i < cap_mul; i++) {
        if (mul_coef_be32) put_be(mul_coef_be32 + 32 * (size_t)i, e.mul[i].c);
        if (mul_witnesses) { mul_witnesses[2 * i] = e.mul[i].l; mul_witnesses[2 * i + 1] = e.mul[i].r; }
    }
    for (uint32_t i = 0; i < head->n_lin && i < cap_lin; i++) {
        if (lin_coef_be32) put_be(lin_coef_be32 + 32 * (size_t)i, e.lin[i].c);
        if (lin_witnesses) lin_witnesses[i] = e.lin[i].w;
    }
    return 1;
} ABI_CATCH

int acvm_batch_error_string(acvm_batch_t *b, const acvm_circuit_t *c, uint32_t instance, char *out, size_t cap) try {
    if (!b || !out || !cap) return set_err(ACVM_E_INVALID, "null argument");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if (instance >= b->B) return set_err(ACVM_E_INVALID, "instance out of range");
    HIPCHK(hipSetDevice(b->device));
    acvm_result_t r;
    fill_result(b, instance, r);
    out[0] = 0;
    if (r.status != ACVM_STATUS_FAILURE) return 0;
    static const char *bb_name[BB_COUNT] = {"and", "xor", "range", "sha256", "blake2s", "schnorr_verify", "pedersen", "hash_to_field_128_security",
                                           "ecdsa_secp256k1", "ecdsa_secp256r1", "fixed_base_scalar_mul", "keccak256", "keccak256",
                                           "recursive_aggregation"};
    const char *func = r.aux0 < BB_COUNT ? bb_name[r.aux0] : "?";
    char msg[512];
    int have = -1;
    if (c) {
        if (r.err == ACVM_ERR_UNSATISFIED || r.err == ACVM_ERR_INDEX_OOB)
            have = acvm_circuit_assert_message(c, r.opcode_index, ACVM_LOCATION_ACIR, msg, sizeof msg);
        else if (r.err == ACVM_ERR_BRILLIG_FAILED && r.n_call_stack)
            have = acvm_circuit_assert_message(c, r.opcode_index, r.call_stack[r.n_call_stack - 1], msg, sizeof msg);
    }
    if (have >= 0) return snprintf(out, cap, "Assertion failed: %s", msg);
    switch (r.err) {
    case ACVM_ERR_MISSING_ASSIGNMENT: return snprintf(out, cap, "Cannot solve opcode: missing assignment for witness index %u", r.aux0);
    case ACVM_ERR_TOO_MANY_UNKNOWNS: {
        // OpcodeNotSolvable::ExpressionHasTooManyUnknowns(Expression) (pwg/mod.rs:72-78): the text carries the expression -- the opcode
        // partially evaluated on the instance's map for Opcode::Arithmetic (arithmetic.rs:31,38-42), the input expression as written for
        // Opcode::Brillig (brillig.rs:46-74)
        const std::string e = c ? too_many_unknowns_expression(b, *c->c, instance, r.opcode_index) : std::string();
        return snprintf(out, cap, "Cannot solve opcode: expression has too many unknowns %s", e.c_str());
    }
    case ACVM_ERR_UNSUPPORTED_BLACKBOX:
        return snprintf(out, cap, "Backend does not currently support the %s opcode. ACVM does not currently have a fallback for this opcode.", func);
    case ACVM_ERR_UNSATISFIED: return snprintf(out, cap, "Cannot satisfy constraint");
    case ACVM_ERR_INDEX_OOB: return snprintf(out, cap, "Index out of bounds, array has size %u, but index was %u", r.aux1, r.aux0);
    case ACVM_ERR_BLACKBOX_FAILED: return snprintf(out, cap, "Failed to solve blackbox function: %s, reason: %s", func, r.message);
    case ACVM_ERR_BRILLIG_FAILED: return snprintf(out, cap, "Failed to solve brillig function, reason: %s", r.message);
    case ACVM_ERR_PANIC: return snprintf(out, cap, "panicked: %s", r.message);
    case ACVM_ERR_DEVICE_LIMIT: return snprintf(out, cap, "Not solved by this library (%s)", r.message);
    default: return snprintf(out, cap, "unknown error %u", r.err);
    }
} ABI_CATCH

// assigned flags of instance j over all witnesses (host side bookkeeping + slow-path bitmap)
static int fetch_assigned(acvm_batch *b, uint32_t first, uint32_t n, uint8_t *assigned) {
    const Plan &p = b->plan;
    uint32_t nw = p.n_witnesses;
    std::vector<uint32_t> bitmap;
    uint32_t n_slow = (uint32_t)b->slow_ids.size();
    bool any_slow = false;
    for (uint32_t i = 0; i < n; i++) any_slow |= b->slow_index[first + i] >= 0;
    if (any_slow) {
        bitmap.resize((size_t)n_slow * b->n_words);
        HIPCHK(hipMemcpy(bitmap.data(), b->d_assigned, bitmap.size() * 4, hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < n; i++) {
        uint8_t *a = assigned + (size_t)i * nw;
        int32_t si = b->slow_index[first + i];
        if (si < 0) {
            for (uint32_t w = 0; w < nw; w++) a[w] = p.producer[w] != 0xFFFFFFFFu;
        } else {
            for (uint32_t w = 0; w < nw; w++) a[w] = (bitmap[(size_t)(w >> 5) * n_slow + si] >> (w & 31)) & 1u;
        }
    }
    return 0;
}

int acvm_batch_witness_map(acvm_batch_t *b, uint32_t first, uint32_t n, uint8_t *assigned, uint8_t *values_be32) try {
    if (!b || !assigned || !values_be32) return set_err(ACVM_E_INVALID, "null argument");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if ((uint64_t)first + n > b->B) return set_err(ACVM_E_INVALID, "instance range out of bounds");
    if (b->side()) return set_err(ACVM_E_STATE, "the batch recycles witness rows (ACVM_BATCH_REUSE_SLOTS) or solved its exact lanes in the side table: full maps are not kept; read the kept witnesses and the digest");
    if (int rc = refuse_if_next_imported(b, nullptr, 0, true)) return rc;
    HIPCHK(hipSetDevice(b->device));
    uint32_t nw = b->plan.n_witnesses;
    if (!n || !nw) return 0;
    if (int rc = fetch_assigned(b, first, n, assigned)) return rc;
    std::vector<uint32_t> sel(nw);
    for (uint32_t w = 0; w < nw; w++) sel[w] = w;
    // stage through a bounded slice of the arena
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)nw * 32));
    if (chunk > n) chunk = n;
    const size_t sel_bytes = align256((size_t)nw * 4);
    if (int rc = stage_reserve(b, sel_bytes + (size_t)chunk * nw * 32)) return rc;
    uint32_t *d_sel = (uint32_t *)b->d_stage;
    uint8_t *d_out = b->d_stage + sel_bytes;
    HIPCHK(hipMemcpyAsync(d_sel, sel.data(), (size_t)nw * 4, hipMemcpyHostToDevice, b->stream));
    for (uint32_t done = 0; done < n; done += chunk) {
        uint32_t m = std::min(chunk, n - done);
        launch_export(b->stream, b->d_W, b->Bp, first + done, m, d_sel, nw, d_out, b->unscale);
        HIPCHK(hipMemcpyAsync(values_be32 + (size_t)done * nw * 32, d_out, (size_t)m * nw * 32, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    for (size_t i = 0; i < (size_t)n * nw; i++)
        if (!assigned[i]) memset(values_be32 + i * 32, 0, 32);
    return 0;
} ABI_CATCH

// digests of the instances of the exact path listed in `flagged` (instance indices >= first), from their own witness maps, into
// out32[(instance - first) * 32]
static int digest_exact_instances(acvm_batch *b, const std::vector<uint32_t> &flagged, uint32_t first, uint8_t *out32) {
    const uint32_t n_slow = (uint32_t)b->slow_ids.size();
    if (b->side()) {  // all lanes of the side table at once (lane t = the t-th flagged instance), then scattered to their instances
        Unscale plain = b->unscale;
        plain.event = b->d_slow_start;  // opcode indices, never 0xFFFFFFFF: every lane of the side table is "an instance of the exact path"
        std::vector<uint8_t> lanes((size_t)n_slow * 32);
        if (int rc = digest_range(b, b->stream, b->d_Wx, b->x_cap, 0, n_slow, plain, (const int32_t *)b->d_ids_x, false, n_slow, lanes.data())) return rc;
        for (uint32_t j : flagged) memcpy(out32 + (size_t)(j - first) * 32, &lanes[(size_t)b->slow_index[j] * 32], 32);
        return 0;
    }
    // plain table: the instance's own column, one launch each (few by construction: acvm_batch_digest takes the table-wide kernel otherwise)
    for (uint32_t j : flagged)
        if (int rc = digest_range(b, b->stream, b->d_W, b->Bp, j, 1, b->unscale, nullptr, true, n_slow, out32 + (size_t)(j - first) * 32)) return rc;
    return 0;
}

// per-instance digest of the solved witness map (definition: kernels_hash.hip, include/acvm_amd.h)
int acvm_batch_digest(acvm_batch_t *b, uint32_t first, uint32_t n, uint8_t *out32) try {
    if (!b || (n && !out32)) return set_err(ACVM_E_INVALID, "null argument");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if ((uint64_t)first + n > b->B) return set_err(ACVM_E_INVALID, "instance range out of bounds");
    if (!n) return 0;
    if (int rc = refuse_if_next_imported(b, nullptr, 0, !(b->plan.n_digest_segments && b->d_leaves))) return rc;  // (a folded digest was summed during the solve)
    HIPCHK(hipSetDevice(b->device));
    const Plan &p = b->plan;
    if (int rc = ensure_digest_tables(b)) return rc;
    std::vector<uint32_t> flagged;
    for (uint32_t i = 0; i < n; i++)
        if (b->slow_index[first + i] >= 0) flagged.push_back(first + i);
    if (p.n_digest_segments && b->d_leaves && !b->force_slow && !b->stepping) {
        // folded into the solve: the partial sums of the generic instances are there; only their total is left (and the instances of the
        // exact path, whose sums come from their own maps below)
        if (int rc = stage_reserve(b, (size_t)n * 32)) return rc;
        launch_digest_final(b->stream, b->d_leaves, p.n_digest_segments, b->Bp, first, n, b->d_event, b->fp, b->d_stage);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out32, b->d_stage, (size_t)n * 32, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        if (flagged.empty()) return 0;
        if (b->side() || flagged.size() <= 64) return digest_exact_instances(b, flagged, first, out32);
        // (plain table with many instances of the exact path -- a whole batch waiting at a foreign call, a batch of failures: the
        // table-wide kernel below serves generic and exact lanes alike through slow_index)
    }
    if (b->side()) {  // the level table does not hold the maps of the exact path's instances
        Unscale u = b->unscale;
        u.event = nullptr;  // (their columns are read as leftovers and overwritten below)
        if (int rc = digest_range(b, b->stream, b->d_W, b->Bp, first, n, u, nullptr, false, 0, out32)) return rc;
        return flagged.empty() ? 0 : digest_exact_instances(b, flagged, first, out32);
    }
    return digest_range(b, b->stream, b->d_W, b->Bp, first, n, b->unscale, nullptr, true, (uint32_t)b->slow_ids.size(), out32);
} ABI_CATCH

int acvm_batch_extract_witnesses(acvm_batch_t *b, const uint32_t *witnesses, uint32_t n_witnesses, uint32_t first, uint32_t n,
                                 uint8_t *values_be32) try {
    if (!b || (n_witnesses && (!witnesses || !values_be32))) return set_err(ACVM_E_INVALID, "null argument");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if ((uint64_t)first + n > b->B) return set_err(ACVM_E_INVALID, "instance range out of bounds");
    if (!n || !n_witnesses) return 0;
    if (int rc = refuse_if_next_imported(b, witnesses, n_witnesses, false)) return rc;
    HIPCHK(hipSetDevice(b->device));
    const uint32_t nw = b->plan.n_witnesses;
    char text[160];
    for (uint32_t k = 0; k < n_witnesses; k++)
        if (witnesses[k] >= nw) {
            snprintf(text, sizeof text, "Failed to extract witness %u from witness map. Witness not found. (instance %u)", witnesses[k], first);
            return set_err(ACVM_E_STATE, text);
        }
    // assigned? An instance the level kernels solved has exactly the planner's set (producer[]); an instance of the exact path
    // has its bitmap. Only the listed witnesses are looked at: O(n + n_slow x n_witnesses), not O(n x all witnesses).
    {
        const uint32_t n_slow = (uint32_t)b->slow_ids.size();
        uint32_t first_fast = 0xFFFFFFFFu;  // first instance of the range that is not an exact lane
        std::vector<uint32_t> lanes;        // exact lanes of the range
        for (uint32_t i = 0; i < n; i++) {
            const int32_t si = b->slow_index[first + i];
            if (si >= 0) lanes.push_back((uint32_t)si);
            else if (first_fast == 0xFFFFFFFFu) first_fast = first + i;
        }
        uint32_t bad_w = 0, bad_j = 0xFFFFFFFFu;
        std::vector<uint32_t> word(n_slow);
        for (uint32_t k = 0; k < n_witnesses; k++) {
            const uint32_t w = witnesses[k];
            if (first_fast != 0xFFFFFFFFu && b->plan.producer[w] == 0xFFFFFFFFu && first_fast < bad_j) { bad_j = first_fast; bad_w = w; }
            if (!lanes.empty()) {
                HIPCHK(hipMemcpy(word.data(), b->d_assigned + (size_t)(w >> 5) * n_slow, (size_t)n_slow * 4, hipMemcpyDeviceToHost));
                for (uint32_t t : lanes)
                    if (!((word[t] >> (w & 31)) & 1u) && b->slow_ids[t] < bad_j) { bad_j = b->slow_ids[t]; bad_w = w; }
            }
        }
        if (bad_j != 0xFFFFFFFFu) {
            for (uint32_t k = 0; k < n_witnesses; k++) {  // the first missing witness of that instance, in the caller's order
                const uint32_t w = witnesses[k];
                const int32_t si = b->slow_index[bad_j];
                bool have = si < 0 ? b->plan.producer[w] != 0xFFFFFFFFu : true;
                if (si >= 0) {
                    uint32_t bits = 0;
                    HIPCHK(hipMemcpy(&bits, b->d_assigned + (size_t)(w >> 5) * n_slow + si, 4, hipMemcpyDeviceToHost));
                    have = (bits >> (w & 31)) & 1u;
                }
                if (!have) { bad_w = w; break; }
            }
            snprintf(text, sizeof text, "Failed to extract witness %u from witness map. Witness not found. (instance %u)", bad_w, bad_j);
            return set_err(ACVM_E_STATE, text);
        }
    }
    if (int rc = reuse_check_kept(b, witnesses, n_witnesses)) return rc;
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)n_witnesses * 32));
    if (chunk > n) chunk = n;
    const size_t sel_bytes = align256((size_t)n_witnesses * 4);
    if (int rc = stage_reserve(b, sel_bytes + (size_t)chunk * n_witnesses * 32)) return rc;
    uint32_t *d_sel = (uint32_t *)b->d_stage;
    uint8_t *d_out = b->d_stage + sel_bytes;
    HIPCHK(hipMemcpyAsync(d_sel, witnesses, (size_t)n_witnesses * 4, hipMemcpyHostToDevice, b->stream));
    for (uint32_t done = 0; done < n; done += chunk) {
        const uint32_t m = std::min(chunk, n - done);
        launch_export(b->stream, b->d_W, b->Bp, first + done, m, d_sel, n_witnesses, d_out, b->unscale, b->d_slot_of);
        HIPCHK(hipMemcpyAsync(values_be32 + (size_t)done * n_witnesses * 32, d_out, (size_t)m * n_witnesses * 32, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    return reuse_patch_exact(b, d_sel, n_witnesses, first, n, values_be32, d_out);
} ABI_CATCH

long long acvm_witness_map_decode(const uint8_t *bytes, size_t len, uint32_t *ids, uint8_t *values_be32, uint32_t cap) try {
    if (!bytes) return set_err(ACVM_E_INVALID, "null argument");
    std::vector<uint32_t> id;
    std::vector<uint8_t> val;
    std::string err;
    if (!witness_map_from_bytes(bytes, len, id, val, err)) return set_err(ACVM_E_MALFORMED, err.c_str());
    for (size_t i = 0; i < id.size() && i < cap; i++) {
        if (ids) ids[i] = id[i];
        if (values_be32) memcpy(values_be32 + 32 * i, val.data() + 32 * i, 32);
    }
    return (long long)id.size();
} ABI_CATCH

long long acvm_witness_map_encode(const uint32_t *ids, const uint8_t *values_be32, uint32_t n, uint8_t *out, size_t cap) try {
    if (n && (!ids || !values_be32)) return set_err(ACVM_E_INVALID, "null argument");
    std::vector<uint8_t> bytes;
    std::string err;
    if (!witness_map_to_bytes(ids, values_be32, n, bytes, err)) return set_err(ACVM_E_INVALID, err.c_str());
    if (out && bytes.size() <= cap) memcpy(out, bytes.data(), bytes.size());
    return (long long)bytes.size();
} ABI_CATCH

long long acvm_batch_witness_map_bytes(acvm_batch_t *b, uint32_t instance, uint8_t *out, size_t cap) try {
    if (!b) return set_err(ACVM_E_INVALID, "null argument");
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if (instance >= b->B) return set_err(ACVM_E_INVALID, "instance out of range");
    const uint32_t nw = b->plan.n_witnesses;
    std::vector<uint8_t> assigned(nw ? nw : 1), values((size_t)(nw ? nw : 1) * 32);
    if (int rc = acvm_batch_witness_map(b, instance, 1, assigned.data(), values.data())) return rc;
    std::vector<uint32_t> ids;
    std::vector<uint8_t> vals;
    for (uint32_t w = 0; w < nw; w++)
        if (assigned[w]) {
            ids.push_back(w);
            vals.insert(vals.end(), values.begin() + (size_t)w * 32, values.begin() + (size_t)w * 32 + 32);
        }
    return acvm_witness_map_encode(ids.data(), vals.data(), (uint32_t)ids.size(), out, cap);
} ABI_CATCH

int acvm_batch_witness(acvm_batch_t *b, uint32_t witness, uint8_t *out_be32, uint8_t *assigned) try {
    if (!b || !out_be32 || !assigned) return set_err(ACVM_E_INVALID, "null argument");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if (witness >= b->plan.n_witnesses) { memset(assigned, 0, b->B); memset(out_be32, 0, (size_t)b->B * 32); return 0; }
    if (int rc = refuse_if_next_imported(b, &witness, 1, false)) return rc;
    HIPCHK(hipSetDevice(b->device));
    if (!b->B) return 0;
    if (int rc = stage_reserve(b, 256 + (size_t)b->B * 32)) return rc;
    uint32_t *d_sel = (uint32_t *)b->d_stage;
    uint8_t *d_out = b->d_stage + 256;
    if (int rc = reuse_check_kept(b, &witness, 1)) return rc;
    HIPCHK(hipMemcpyAsync(d_sel, &witness, 4, hipMemcpyHostToDevice, b->stream));
    launch_export(b->stream, b->d_W, b->Bp, 0, b->B, d_sel, 1, d_out, b->unscale, b->d_slot_of);
    HIPCHK(hipMemcpyAsync(out_be32, d_out, (size_t)b->B * 32, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (int rc = reuse_patch_exact(b, d_sel, 1, 0, b->B, out_be32, d_out)) return rc;
    std::vector<uint32_t> bitmap;
    uint32_t n_slow = (uint32_t)b->slow_ids.size();
    if (n_slow) {
        bitmap.resize(n_slow);
        HIPCHK(hipMemcpy(bitmap.data(), b->d_assigned + (size_t)(witness >> 5) * n_slow, (size_t)n_slow * 4, hipMemcpyDeviceToHost));
    }
    for (uint32_t j = 0; j < b->B; j++) {
        int32_t si = b->slow_index[j];
        assigned[j] = si < 0 ? b->plan.producer[witness] != 0xFFFFFFFFu : (bitmap[si] >> (witness & 31)) & 1u;
        if (!assigned[j]) memset(out_be32 + (size_t)j * 32, 0, 32);
    }
    return 0;
} ABI_CATCH

int acvm_batch_stats(acvm_batch_t *b, acvm_stats_t *out) {
    if (!b || !out) return set_err(ACVM_E_INVALID, "null argument");
    const Plan &p = b->plan;
    plan_stats(p, out);
    out->n_kernel_launches = b->n_launches;
    out->n_slow_instances = (uint32_t)b->slow_ids.size();
    out->solve_device_ms = b->solve_device_ms;
    out->arith_kernel_ms = b->arith_kernel_ms;
    out->dyn_kernel_ms = b->dyn_kernel_ms;
    for (int k = 0; k < 4; k++) out->class_kernel_ms[k] = b->cls_kernel_ms[k];
    out->class_kernel_ms[CLS_GRUMPKIN] += b->cls_kernel_ms[CLS_PEDERSEN] + b->cls_kernel_ms[CLS_ECDSA] + b->cls_kernel_ms[CLS_HOSTBB];
    out->slow_path_ms = b->slow_path_ms;
    out->n_brillig_retries = b->n_brillig_retries;
    return 0;
}

}  // extern "C"
