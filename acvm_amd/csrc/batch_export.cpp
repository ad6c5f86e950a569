// batch_export.cpp -- what leaves the device after a solve: results and message texts (OpcodeResolutionError, acvm/src/pwg/mod.rs:100-114),
// witnesses and witness maps (ACVM::witness_map / finalize, :161,176-181), the per-instance map digest, the error string and expression of
// acvm_js/src/execute.rs:79-108, public-witness extraction (acvm_js/src/public_witness.rs), the WitnessMap wire format.
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
#include "display.hpp"

static const uint8_t DIGEST_G[32] = {0x23, 0x35, 0x53, 0x18, 0xdb, 0xff, 0xab, 0x2f, 0xb7, 0x72, 0x11, 0x7c, 0x57, 0x5c, 0x61, 0xb1,
                                     0x79, 0xf8, 0xc9, 0x83, 0x3c, 0x83, 0xba, 0x65, 0x59, 0x7e, 0x17, 0x3c, 0x35, 0xc4, 0xbb, 0xe3};
static const uint8_t DIGEST_H[32] = {0x28, 0x25, 0x78, 0x33, 0xe7, 0x23, 0x7f, 0xbd, 0x29, 0x7c, 0x55, 0x74, 0x6b, 0xe0, 0xa3, 0xa9,
                                     0x8a, 0x2a, 0x89, 0x8d, 0x8a, 0xb1, 0x0b, 0xe0, 0x05, 0xaa, 0x2f, 0xdf, 0x9c, 0x60, 0x11, 0xa4};
// device tables of the digest: g^(w+1), g^(w+1) / scale_w for the scaled witnesses, h^(w+1), and the h-sum of the planner's assigned set
int ensure_digest_tables(acvm_batch *b) {
    if (b->d_fp_g) return 0;
    const Plan &p = b->plan();
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


int digest_range(acvm_batch *b, hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, const Unscale &u, const int32_t *d_slow_index,
                        bool use_host_index, uint32_t n_slow, uint8_t *out32) {
    const Plan &p = b->plan();
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

int batch_export_tile(acvm_batch *b, uint32_t n, const uint32_t *keep, uint32_t n_keep, acvm_result_t *results, uint8_t *kept_values, uint8_t *kept_assigned,
                      uint8_t *digests) {
    const Plan &p = b->plan();
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


// after acvm_batch_solve_then_import the rows of the INITIAL witnesses hold the next tile's values: whatever reads them back is refused
int refuse_if_next_imported(const acvm_batch *b, const uint32_t *ws, uint32_t n, bool whole_map) {
    if (!b->next_imported) return 0;
    bool hit = whole_map;
    for (uint32_t k = 0; k < n && !hit; k++) hit = std::find(b->plan().initial_ids.begin(), b->plan().initial_ids.end(), ws[k]) != b->plan().initial_ids.end();
    if (!hit) return 0;
    return set_err(ACVM_E_STATE, "the initial witnesses of this solve are gone: acvm_batch_solve_then_import put the next tile's inputs into the table behind the solve "
                                 "(read results, non-initial witnesses and nothing else; or use acvm_batch_solve)");
}

// ---- slot reuse (ACVM_BATCH_REUSE_SLOTS): what can be read back
static bool reuse_kept(const acvm_batch *b, uint32_t w) {
    const Plan &p = b->plan();
    if (w >= p.n_witnesses) return false;
    if (std::find(p.initial_ids.begin(), p.initial_ids.end(), w) != p.initial_ids.end()) return true;
    return std::find(b->opts.keep.begin(), b->opts.keep.end(), w) != b->opts.keep.end();
}
static int reuse_check_kept(const acvm_batch *b, const uint32_t *ws, uint32_t n) {
    if (!b->reuse()) return 0;
    for (uint32_t k = 0; k < n; k++)
        if (ws[k] < b->plan().n_witnesses && !reuse_kept(b, ws[k]))
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
bool fetch_one(acvm_batch *b, uint32_t j, uint32_t w, uint8_t out[32]) {
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
void format_message(acvm_batch *b, uint32_t j, const SlowResult &sr, acvm_result_t &r) {
    const Plan &p = b->plan();
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
        if (it == b->host_bb_msg.end()) {  // (a callback made from inside a Brillig program: its text outlives the re-solve that reports it)
            it = b->fc_fail_msg.find(j);
            if (it == b->fc_fail_msg.end()) it = b->host_bb_msg.end();
        }
        snprintf(r.message, sizeof r.message, "%s", it == b->host_bb_msg.end() ? "" : it->second.c_str());
        break;
    }
    default: break;
    }
}

void fill_result(acvm_batch *b, uint32_t j, acvm_result_t &r) {
    memset(&r, 0, sizeof r);
    if (b->plan().n_opcodes == 0 || b->slow_index[j] < 0) { r.status = ACVM_STATUS_SOLVED; return; }
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
        if (w >= b->plan().n_witnesses) return false;
        if (lane < 0) return b->plan().producer[w] != 0xFFFFFFFFu;
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
    const Plan &p = b->plan();
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
    uint32_t nw = b->plan().n_witnesses;
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
    if (int rc = refuse_if_next_imported(b, nullptr, 0, !(b->plan().n_digest_segments && b->d_leaves))) return rc;  // (a folded digest was summed during the solve)
    HIPCHK(hipSetDevice(b->device));
    const Plan &p = b->plan();
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

// SURVEY 8d's digest as written -- Blake2s over the witness vector's bytes -- in tree form (definition: include/acvm_amd.h). The whole table must be
// there: not with recycled rows, not while an asynchronous exact job holds instances in its side table.
int acvm_batch_digest_blake2s(acvm_batch_t *b, uint32_t first, uint32_t n, uint8_t *out32) try {
    if (!b || (n && !out32)) return set_err(ACVM_E_INVALID, "null argument");
    if (b->pending)
        if (int rc = batch_finish_pending(b, &b->last_outcome)) return rc;
    if (!b->solved) return set_err(ACVM_E_STATE, "batch not solved");
    if ((uint64_t)first + n > b->B) return set_err(ACVM_E_INVALID, "instance range out of bounds");
    if (!n) return 0;
    if (b->side()) return set_err(ACVM_E_STATE, "the byte-wise digest hashes every witness row: not with ACVM_BATCH_REUSE_SLOTS (rows are recycled) nor while instances of "
                                                "the exact path live in a side table; acvm_batch_digest serves those");
    if (int rc = refuse_if_next_imported(b, nullptr, 0, true)) return rc;
    HIPCHK(hipSetDevice(b->device));
    const Plan &p = b->plan();
    hipStream_t s = b->stream;
    const size_t idx_bytes = align256((size_t)b->B * 4);
    const size_t leaf_bytes = align256((size_t)digest_b2s_leaves(p.n_witnesses) * 32 * n);
    if (int rc = stage_reserve(b, idx_bytes + leaf_bytes + (size_t)n * 32)) return rc;
    HIPCHK(hipMemcpyAsync(b->d_stage, b->slow_index.data(), (size_t)b->B * 4, hipMemcpyHostToDevice, s));
    uint32_t *d_leaves = (uint32_t *)(b->d_stage + idx_bytes);
    uint8_t *d_out = b->d_stage + idx_bytes + leaf_bytes;
    launch_digest_blake2s(s, b->d_W, b->Bp, first, n, p.n_witnesses, b->d_producer, b->unscale, (const int32_t *)b->d_stage, b->d_assigned, (uint32_t)b->slow_ids.size(), d_leaves, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out32, d_out, (size_t)n * 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
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
    const uint32_t nw = b->plan().n_witnesses;
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
            if (first_fast != 0xFFFFFFFFu && b->plan().producer[w] == 0xFFFFFFFFu && first_fast < bad_j) { bad_j = first_fast; bad_w = w; }
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
                bool have = si < 0 ? b->plan().producer[w] != 0xFFFFFFFFu : true;
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
    const uint32_t nw = b->plan().n_witnesses;
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
    if (witness >= b->plan().n_witnesses) { memset(assigned, 0, b->B); memset(out_be32, 0, (size_t)b->B * 32); return 0; }
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
        assigned[j] = si < 0 ? b->plan().producer[witness] != 0xFFFFFFFFu : (bitmap[si] >> (witness & 31)) & 1u;
        if (!assigned[j]) memset(out_be32 + (size_t)j * 32, 0, 32);
    }
    return 0;
} ABI_CATCH


