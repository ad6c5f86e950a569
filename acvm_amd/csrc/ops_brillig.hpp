// ops_brillig.hpp -- the Brillig register VM on the device, one lane per (instance, Opcode::Brillig):
//   BrilligSolver::solve / zero_out_brillig_outputs   acvm/src/pwg/brillig.rs:20-150
//   VM::process_opcode, registers, memory              brillig_vm/src/{lib.rs:61-390, registers.rs:4-43, memory.rs:4-45}
//   field / fixed-width integer ALU                    brillig_vm/src/arithmetic.rs:7-98 (BigUint semantics, bit_size <= 256)
//   black box ops                                      brillig_vm/src/black_box.rs:42-165
// Registers and memory live in per-lane device scratch laid out like the witness table ([slot][half][instance]), the call
// stack and the hash staging bytes behind them. Control flow is per lane (SIMT divergence does the masking). The reference's VM has
// no limits (memory grows on write, memory.rs:27-39; no step or call-depth bound, lib.rs:154-307); a kernel needs them: a memory
// capacity, a step limit and a call-stack depth arrive with the launch (BrilligLimits). An instance that reaches one leaves the level
// schedule, and the exact path RETRIES its opcode with the limits raised (batch.cpp retry_device_limits) up to the library's stated
// maxima; past those acvm_batch_solve fails as a whole with ACVM_E_UNSUPPORTED -- an instance never reports a failure the reference
// would not.
#pragma once
#include "kernels.hpp"
#include "ops_ecdsa.hpp"
#include "ops_grumpkin.hpp"
#include "ops_light.hpp"

namespace acvm {

enum BrOp : uint32_t {
    BRO_BINARY_FIELD_OP = 0, BRO_BINARY_INT_OP, BRO_JUMP_IF_NOT, BRO_JUMP_IF, BRO_JUMP, BRO_CALL, BRO_CONST, BRO_RETURN,
    BRO_FOREIGN_CALL, BRO_MOV, BRO_LOAD, BRO_STORE, BRO_BLACK_BOX, BRO_TRAP, BRO_STOP
};

struct BrVm {
    uint4 *slots;     // Fr slots: registers [0, n_regs), memory [n_regs, n_regs + mem_cap)
    uint32_t *words;  // call stack (cs_cap words) then hash staging, word w of the lane at words[w * Bp + j]
    uint64_t Bp, j;   // stride and column of this lane in the VM's scratch
    uint64_t iBp, ij; // stride and column of the INSTANCE in batch-wide tables (the foreign-call result store)
    uint32_t n_regs, mem_cap, cs_cap, n_mem, n_cs, pc;
    uint32_t status;  // 0 running, 1 finished, 2 failure (trap / return / black box), 4 panic, 5 device limit
    uint32_t code, x0;
    Fr val;

    __device__ __forceinline__ void panic(uint32_t c) { if (!status) { status = 4; code = c; } }
    __device__ __forceinline__ Fr reg_get(uint32_t r) {
        if (r >= 65536u) { panic(BP_REG_READ); return fr_zero(); }
        return fr_load(slots, r, Bp, j);
    }
    __device__ __forceinline__ void reg_set(uint32_t r, const Fr &v) {
        if (r >= 65536u) { panic(BP_REG_WRITE); return; }
        fr_store(slots, r, Bp, j, v);
    }
    // Value::to_usize (brillig/src/value.rs:46-49): panics above u64; device addresses are 32-bit, larger ones can only fail
    __device__ __forceinline__ bool to_usize(const Fr &v, uint64_t &out) {
        const Fr c = fr_to_canonical(v);
        if (c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7]) { panic(BP_U64); return false; }
        out = (uint64_t)c.v[1] << 32 | c.v[0];
        return true;
    }
    __device__ __forceinline__ bool mem_check_read(uint64_t ptr, uint64_t len) {
        if (ptr > n_mem || len > n_mem - ptr) { panic(BP_MEM_READ); return false; }
        return true;
    }
    __device__ __forceinline__ Fr mem_get(uint32_t cell) { return fr_load(slots, n_regs + cell, Bp, j); }
    // memory.rs:32-39 write: grows with zeros
    __device__ __forceinline__ bool mem_write(uint64_t ptr, const Fr &v) {
        if (ptr >= mem_cap) {
            if (!status) { status = 5; code = DM_BRILLIG_MEM_CAP; x0 = ptr > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ptr; }
            return false;
        }
        for (uint32_t c = n_mem; c < (uint32_t)ptr; c++) fr_store(slots, n_regs + c, Bp, j, fr_zero());
        fr_store(slots, n_regs + (uint32_t)ptr, Bp, j, v);
        if ((uint32_t)ptr + 1u > n_mem) n_mem = (uint32_t)ptr + 1u;
        return true;
    }
};

// evaluate_binary_bigint_op through the VM: a panic of the reference stops the lane (ops_light.hpp int_op_core)
static inline __device__ __noinline__ Fr brillig_int_op(BrVm &vm, uint32_t op, uint32_t bits, const Fr &fa, const Fr &fb, const Fr &pow2) {
    uint32_t panic = 0;
    const Fr r = int_op_core(op, bits, fa, fb, panic, &pow2);
    if (panic) vm.panic(panic);
    return r;
}

// black_box.rs:42-165. Operand words: see plan.cpp (HeapVector = pointer reg + size reg, HeapArray = pointer reg + literal)
static inline __device__ __noinline__ void brillig_black_box(BrVm &vm, uint32_t bbop, const uint32_t *__restrict__ w, const GrumpkinTables &T,
                                                             const uint32_t *__restrict__ ecdsa_g) {
    MsgBuf m{vm.words + (uint64_t)vm.cs_cap * vm.Bp, vm.Bp, vm.j, 0u, 0u};
    auto heap_vector = [&](uint32_t preg, uint32_t sreg, uint64_t &ptr, uint64_t &len) {
        const Fr pv = vm.reg_get(preg), sv = vm.reg_get(sreg);
        return !vm.status && vm.to_usize(pv, ptr) && vm.to_usize(sv, len);
    };
    auto mem_byte = [&](uint64_t ptr, uint32_t i) { return fr_to_canonical(vm.mem_get((uint32_t)ptr + i)).v[0] & 0xffu; };
    uint64_t ptr = 0, len = 0, optr = 0;
    switch (bbop) {
    case 0: case 1: case 2: case 3: {  // Sha256, Blake2s, Keccak256, HashToField128Security
        if (!heap_vector(w[0], w[1], ptr, len) || !vm.mem_check_read(ptr, len)) return;
        m.begin();
        for (uint32_t i = 0; i < (uint32_t)len; i++) m.put(mem_byte(ptr, i));
        m.end();
        const Digest d = bbop == 0u ? sha256_msg(m, (uint32_t)len) : (bbop == 2u ? keccak256_msg(m, (uint32_t)len) : blake2s_msg(m, (uint32_t)len));
        if (bbop == 3u) { vm.reg_set(w[2], digest_to_field(d)); return; }
        const Fr ov = vm.reg_get(w[2]);
        if (vm.status || !vm.to_usize(ov, optr)) return;
        for (uint32_t i = 0; i < 32u; i++)
            if (!vm.mem_write(optr + i, fr_from_byte(d.byte(i)))) return;
        return;
    }
    case 4: case 5: {  // EcdsaSecp256k1 / r1 (black_box.rs:74-131): hashed_msg (ptr, size), pkx / pky / signature (ptr, literal), result
        uint64_t ap[3] = {0, 0, 0};
        for (int g = 0; g < 3; g++) {
            const Fr pv = vm.reg_get(w[2 + 2 * g]);
            if (vm.status || !vm.to_usize(pv, ap[g]) || !vm.mem_check_read(ap[g], w[3 + 2 * g])) return;
            if (w[3 + 2 * g] != (g == 2 ? 64u : 32u)) { vm.status = 2; vm.code = DM_BRILLIG_BB_FAILED; vm.x0 = (bbop - 4u) * 4u + (uint32_t)g; return; }
        }
        if (!heap_vector(w[0], w[1], ptr, len) || !vm.mem_check_read(ptr, len)) return;
        uint32_t panic = 0;
        const uint32_t ok = ecdsa_verify(
            bbop - 4u, [&](uint32_t i) { return mem_byte(ap[0], i); }, [&](uint32_t i) { return mem_byte(ap[1], i); },
            [&](uint32_t i) { return mem_byte(ap[2], i); }, (uint32_t)len, [&](uint32_t i) { return mem_byte(ptr, i); }, ecdsa_g, &panic);
        if (panic) { vm.status = 4; vm.code = 110u + panic; return; }
        vm.reg_set(w[8], ok ? fr_one() : fr_zero());
        return;
    }
    case 6: {  // SchnorrVerify: pkx, pky, message (ptr, size), signature (ptr, size), result
        const Fr pkx = vm.reg_get(w[0]), pky = vm.reg_get(w[1]);
        uint64_t mp = 0, ml = 0, sp = 0, sl = 0;
        if (!heap_vector(w[2], w[3], mp, ml) || !vm.mem_check_read(mp, ml)) return;
        if (!heap_vector(w[4], w[5], sp, sl) || !vm.mem_check_read(sp, sl)) return;
        if (sl < 64u) { vm.status = 4; vm.code = 100u; vm.x0 = (uint32_t)sl; return; }      // lib.rs:50-52 slice panic
        if (128u + ml >= 1024u) { vm.status = 4; vm.code = 101u; return; }                  // wasm/schnorr.rs:79-82
        const bool ok = grumpkin_schnorr_verify(T, pkx, pky, [&](uint32_t i) { return mem_byte(sp, i); }, (uint32_t)ml,
                                                [&](uint32_t i) { return mem_byte(mp, i); }, m);
        vm.reg_set(w[6], ok ? fr_one() : fr_zero());
        return;
    }
    case 7: {  // Pedersen: inputs (ptr, size), domain separator, output (ptr, literal)
        if (!heap_vector(w[0], w[1], ptr, len) || !vm.mem_check_read(ptr, len)) return;
        const Fr ds = fr_to_canonical(vm.reg_get(w[2]));
        if (vm.status) return;
        if (ds.v[1] | ds.v[2] | ds.v[3]) { vm.status = 2; vm.code = DM_PEDERSEN_DOMAIN; return; }  // to_u128().try_into::<u32>() fails
        Fr x, y;
        grumpkin_pedersen(T, (uint32_t)len, ds.v[0], [&](uint32_t i) { return vm.mem_get((uint32_t)ptr + i); }, x, y);
        const Fr ov = vm.reg_get(w[3]);
        if (vm.status || !vm.to_usize(ov, optr)) return;
        if (vm.mem_write(optr, x)) vm.mem_write(optr + 1, y);
        return;
    }
    case 8: {  // FixedBaseScalarMul: low, high, result (ptr, literal)
        const Fr lo = vm.reg_get(w[0]), hi = vm.reg_get(w[1]);
        if (vm.status) return;
        Fr x, y;
        const OpResult e = grumpkin_fixed_base_values<FastPolicy>(lo, hi, T, x, y);
        if (e.err) {  // BlackBoxResolutionError::Failed -> the VM fails with its Display string
            vm.status = 2;
            vm.code = e.msg;
            vm.val = fr_to_canonical(e.msg == DM_LIMB_HIGH ? hi : lo);
            if (e.msg == DM_SCALAR) {
                const Fr cl = fr_to_canonical(lo), ch = fr_to_canonical(hi);
                vm.val = Fr{{cl.v[0], cl.v[1], cl.v[2], cl.v[3], ch.v[0], ch.v[1], ch.v[2], ch.v[3]}};
            }
            return;
        }
        const Fr ov = vm.reg_get(w[2]);
        if (vm.status || !vm.to_usize(ov, optr)) return;
        if (vm.mem_write(optr, x)) vm.mem_write(optr + 1, y);
        return;
    }
    default: vm.panic(BP_BAD_BB); return;
    }
}

// [K_BRILLIG, opcode, has_pred, n_inputs, n_outputs, bc_offset, n_bytecode, n_regs, mem_cap, fc_desc_off, fc_vals_off, fc_slot,
//  E(pred)?, inputs: (is_array, n, E x n)..., outputs: (is_array, n, (w, flag) x n)...]
// ForeignCall (brillig_vm/src/lib.rs:190-274). Results come first from the circuit (Brillig::foreign_call_results), then
// from what the host resolved for this instance and opcode (the batch-wide store, kernels.hpp FcStoreSlot: level kernels and
// exact kernels alike). Without a result the VM stops with status 3 and, on the exact path, hands the resolved inputs to the
// host (ForeignCallWaitInfo, pwg/brillig.rs:157-163).
static inline __device__ __noinline__ void brillig_foreign_call(BrVm &vm, const uint32_t *__restrict__ ex, const DeviceProgram &dp, uint32_t fc_desc_off,
                                                                uint32_t fc_vals_off, uint32_t &fc_counter, uint32_t n_bc, uint32_t fc_slot, const ExactLanes *L, uint32_t t) {
    const uint32_t n_dests = ex[0], n_in = ex[1];
    const uint32_t *dests = ex + 2, *ins = dests + 3 * n_dests;
    // locate result number fc_counter: static table first, then the instance's table
    const uint32_t *sdesc = dp.bytecode + fc_desc_off;
    const uint32_t n_static = sdesc[0];
    const bool is_static = fc_counter < n_static;
    const uint32_t n_slow = L ? L->n_slow : 0u;
    FcStoreSlot st{nullptr, nullptr};
    if (!is_static && dp.fc_store && fc_slot != K_NONE) st = dp.fc_store[fc_slot];
    auto desc = [&](uint32_t w) { return is_static ? sdesc[w] : st.desc[(uint64_t)w * vm.iBp + vm.ij]; };
    auto value = [&](uint32_t i) { return is_static ? fr_const(dp.consts, dp.bytecode[fc_vals_off + i]) : fr_load(st.vals, i, vm.iBp, vm.ij); };
    uint32_t k = fc_counter;
    bool found = is_static;
    if (!is_static) {
        k -= n_static;
        found = st.desc && k < st.desc[vm.ij];
    }
    if (!found) {
        vm.status = 3;
        if (!L) return;
        // ForeignCallWaitInfo inputs: registers as one value, heap arrays / vectors as their memory slice
        uint32_t nv = 0;
        if (1u + n_in > L->fc.pend_desc_words) { vm.status = 5; vm.code = DM_FC_PENDING_CAP; return; }
        L->fc.pend_desc[t] = n_in;
        for (uint32_t i = 0; i < n_in; i++) {
            const uint32_t kind = ins[3 * i], rg = ins[3 * i + 1], sz = ins[3 * i + 2];
            uint64_t start = 0, size = 1;
            if (kind != 0u) {
                const Fr pv = vm.reg_get(rg);
                if (vm.status == 4 || !vm.to_usize(pv, start)) return;
                size = sz;
                if (kind == 2u) {
                    const Fr sv = vm.reg_get(sz);
                    if (vm.status == 4 || !vm.to_usize(sv, size)) return;
                }
                if (!vm.mem_check_read(start, size)) return;
            }
            if (nv + size > L->fc.pend_vals_cap) { vm.status = 5; vm.code = DM_FC_PENDING_CAP; return; }
            L->fc.pend_desc[(uint64_t)(1u + i) * n_slow + t] = (uint32_t)size;
            for (uint32_t c = 0; c < (uint32_t)size; c++) {
                const Fr v = kind == 0u ? vm.reg_get(rg) : vm.mem_get((uint32_t)start + c);
                if (vm.status == 4) return;
                fr_store(L->fc.pend_vals, nv++, n_slow, t, v);
            }
        }
        return;
    }
    // walk to result k
    uint32_t pos = 1, vpos = 0;
    for (uint32_t rr = 0; rr < k; rr++) {
        const uint32_t nvals = desc(pos++);
        for (uint32_t v = 0; v < nvals; v++, pos += 2) vpos += desc(pos + 1);
    }
    const uint32_t n_res = desc(pos++);
    if (n_res >= 0xFFFFFFF0u) {
        // an INTERNAL call (the caller's BlackBoxFunctionSolver inside this program, plan.cpp) whose callback did not return Ok: the VM fails
        // here like the reference's does on evaluate_black_box's error (brillig_vm/src/lib.rs:298-307: self.fail(e.to_string())), a panicking
        // solver panics; the text is the host's (batch_exact.cpp resolve_internal_calls)
        vm.code = DM_HOST_MESSAGE;
        vm.status = n_res - 0xFFFFFFF0u >= 3u ? 5u : 2u;
        return;
    }
    bool invalid = false;
    const uint32_t nz = n_dests < n_res ? n_dests : n_res;
    for (uint32_t i = 0; i < nz; i++, pos += 2) {
        const uint32_t is_array = desc(pos), n = desc(pos + 1);
        const uint32_t kind = dests[3 * i], rg = dests[3 * i + 1], sz = dests[3 * i + 2];
        if (kind == 0u) {
            if (is_array) { vm.status = 4; vm.code = 102u; return; }  // "Function result size does not match brillig bytecode (expected 1 result)"
            vm.reg_set(rg, value(vpos));
        } else {
            if (!is_array) { vm.status = 4; vm.code = 103u; return; }  // "Function result size does not match brillig bytecode size"
            if (kind == 1u) {
                if (n != sz) { invalid = true; break; }
            } else vm.reg_set(sz, fr_from_u32(n));
            uint64_t dst = 0;
            const Fr pv = vm.reg_get(rg);
            if (vm.status == 4 || !vm.to_usize(pv, dst)) return;
            for (uint32_t c = 0; c < n; c++)
                if (!vm.mem_write(dst + c, value(vpos + c))) return;
        }
        if (vm.status == 4) return;
        vpos += n;
    }
    // lib.rs:262-270: both checks only record a failure; the program counter still advances
    bool failed = false;
    if (n_dests != n_res) { failed = true; vm.code = DM_FC_COUNT; vm.x0 = n_res; vm.val.v[0] = n_dests; }
    if (invalid) { failed = true; vm.code = DM_FC_SIZE; }
    fc_counter++;
    if (failed) vm.status = vm.pc + 1u >= n_bc ? 1u : 2u;  // running off the end turns the failure into Finished
}

template <class P>
__device__ __forceinline__ OpResult op_brillig(const P &p, const uint32_t *__restrict__ r, const DeviceProgram &dp, uint32_t *scratch, SlowResult *res,
                                               const ExactLanes *L, uint32_t t) {
    const uint32_t has_pred = r[2], n_inputs = r[3], n_outputs = r[4], n_bc = r[6];
    const uint32_t *__restrict__ bc = dp.bytecode + r[5];
    const uint32_t *q = r + 12;
    uint32_t fc_counter = 0;
    Fr pred = fr_one();
    if (has_pred) {  // brillig.rs:28-31: get_value error passes through (MissingAssignment)
        const OpResult e = expr_value(p, q, dp.consts, pred);
        if (e.err) return e;
        q += expr_len(q);
    }
    // locate the outputs behind the inputs
    const uint32_t *inputs = q;
    for (uint32_t i = 0; i < n_inputs; i++) {
        const uint32_t n = q[1];
        q += 2;
        for (uint32_t k = 0; k < n; k++) q += expr_len(q);
    }
    const uint32_t *outputs = q;
    if (fr_is_zero(pred)) {  // zero_out_brillig_outputs (brillig.rs:133-150)
        q = outputs;
        for (uint32_t i = 0; i < n_outputs; i++) {
            const uint32_t n = q[1];
            q += 2;
            for (uint32_t k = 0; k < n; k++, q += 2)
                if (!p.insert(q[0], fr_zero(), q[1])) return op_fail(DE_UNSATISFIED);
        }
        return op_ok();
    }
    // limits and scratch addressing of this launch: the level kernels and the first pass of the exact kernels use the record's memory
    // capacity and the lane's column of the class scratch; a retry pass of the exact path (L->br_lane) brings larger limits and a
    // scratch that holds only the lanes being retried
    const BrilligLimits &lim = dp.brillig;
    const bool compact = L && L->br_lane;
    if (compact && L->br_lane[t] == K_NONE) return op_ok();  // (not reached: a lane that is not retried is not InProgress)
    BrVm vm;
    vm.n_regs = r[7];
    vm.mem_cap = lim.mem_cap ? lim.mem_cap : r[8];
    vm.cs_cap = lim.call_depth;
    vm.slots = (uint4 *)scratch;
    vm.Bp = compact ? lim.stride : p.scratch_stride();
    vm.j = compact ? L->br_lane[t] : p.scratch_lane();
    vm.iBp = p.Bp;
    vm.ij = p.j;
    vm.words = scratch + (uint64_t)(vm.n_regs + vm.mem_cap) * 8u * vm.Bp;
    vm.n_mem = vm.n_cs = vm.pc = vm.status = vm.code = vm.x0 = 0u;
    vm.val = fr_zero();
    for (uint32_t i = 0; i < vm.n_regs; i++) fr_store(vm.slots, i, vm.Bp, vm.j, fr_zero());  // unset registers read 0 (registers.rs:25-33)
    // inputs (brillig.rs:46-74): an expression that does not reduce to a constant is ExpressionHasTooManyUnknowns
    q = inputs;
    for (uint32_t i = 0; i < n_inputs; i++) {
        const uint32_t is_array = q[0], n = q[1];
        q += 2;
        const uint32_t base = vm.n_mem;
        for (uint32_t k = 0; k < n; k++) {
            Fr v;
            const OpResult e = expr_value(p, q, dp.consts, v);
            if (e.err) return op_fail(DE_TOO_MANY_UNKNOWNS);
            q += expr_len(q);
            if (is_array) { if (!vm.mem_write(vm.n_mem, v)) return op_fail_msg(DE_PANIC, 0, DM_BRILLIG_MEM_CAP, vm.x0); }
            else vm.reg_set(i, v);
        }
        if (is_array) vm.reg_set(i, fr_from_u32(base));
    }
    // process_opcodes (lib.rs:136-142)
    if (n_bc == 0) vm.panic(BP_BYTECODE_OOB);
    uint32_t steps = 0;
    while (!vm.status) {
        if (++steps > lim.steps) { vm.status = 5; vm.code = DM_BRILLIG_STEP_LIMIT; break; }
        const uint32_t *__restrict__ ins = bc + 8u * vm.pc;
        const uint32_t op = ins[0], a = ins[1], b = ins[2], c = ins[3];
        uint32_t next = vm.pc + 1u;
        switch (op) {
        case BRO_BINARY_FIELD_OP: {  // arithmetic.rs:7-20
            const Fr x = vm.reg_get(b), y = vm.reg_get(c);
            Fr v;
            switch (ins[4] & 0xffu) {
            case 0: v = fr_add(x, y); break;
            case 1: v = fr_sub(x, y); break;
            case 2: v = fr_mul(x, y); break;
            case 3: v = fr_mul(x, fr_inv(y)); break;
            default: v = fr_eq(x, y) ? fr_one() : fr_zero(); break;
            }
            vm.reg_set(a, v);
            break;
        }
        case BRO_BINARY_INT_OP: {
            const Fr x = vm.reg_get(b), y = vm.reg_get(c);
            if (vm.status) break;
            // (word 6: the constant 2^bit_size mod p of a Sub wider than 256 bits, else unused)
            const uint32_t bits = ins[4] >> 8;
            const Fr v = brillig_int_op(vm, ins[4] & 0xffu, bits, x, y, bits > 256u && (ins[4] & 0xffu) == 1u ? fr_const(dp.consts, ins[6]) : fr_zero());
            if (!vm.status) vm.reg_set(a, v);
            break;
        }
        case BRO_JUMP: next = ins[5]; break;
        case BRO_JUMP_IF: if (!fr_is_zero(vm.reg_get(a))) next = ins[5]; break;
        case BRO_JUMP_IF_NOT: if (fr_is_zero(vm.reg_get(a))) next = ins[5]; break;
        case BRO_RETURN:
            if (vm.n_cs) next = vm.words[(uint64_t)(--vm.n_cs) * vm.Bp + vm.j] + 1u;
            else { vm.status = 2; vm.code = DM_BRILLIG_RETURN; }
            break;
        case BRO_CALL:
            if (vm.n_cs >= vm.cs_cap) { vm.status = 5; vm.code = DM_BRILLIG_CALL_DEPTH; break; }
            vm.words[(uint64_t)(vm.n_cs++) * vm.Bp + vm.j] = vm.pc;
            next = ins[5];
            break;
        case BRO_CONST: vm.reg_set(a, fr_const(dp.consts, ins[6])); break;
        case BRO_MOV: vm.reg_set(a, vm.reg_get(b)); break;
        case BRO_LOAD: {  // a = destination, b = source pointer
            uint64_t u;
            const Fr pv = vm.reg_get(b);
            if (vm.status || !vm.to_usize(pv, u) || !vm.mem_check_read(u, 1)) break;
            vm.reg_set(a, vm.mem_get((uint32_t)u));
            break;
        }
        case BRO_STORE: {  // a = destination pointer, b = source
            uint64_t u;
            const Fr pv = vm.reg_get(a);
            if (vm.status || !vm.to_usize(pv, u)) break;
            const Fr v = vm.reg_get(b);
            if (!vm.status) vm.mem_write(u, v);
            break;
        }
        case BRO_TRAP: vm.status = 2; vm.code = DM_BRILLIG_TRAP; break;
        case BRO_STOP: vm.status = 1; break;
        case BRO_BLACK_BOX: brillig_black_box(vm, ins[4], dp.bytecode + ins[7], dp.grumpkin, dp.ecdsa_g); break;
        case BRO_FOREIGN_CALL: brillig_foreign_call(vm, dp.bytecode + ins[7], dp, r[9], r[10], fc_counter, n_bc, r[11], L, t); break;
        default: vm.panic(BP_BAD_OPCODE); break;
        }
        if (vm.status) break;
        vm.pc = next;  // set_program_counter (lib.rs:322-329)
        if (vm.pc >= n_bc) vm.status = 1;
    }
    if (vm.status == 1) {  // Finished (brillig.rs:95-111): register i -> output i
        q = outputs;
        for (uint32_t i = 0; i < n_outputs; i++) {
            const uint32_t is_array = q[0], n = q[1];
            q += 2;
            const Fr rv = vm.reg_get(i);
            if (!is_array) {
                if (!p.insert(q[0], rv, q[1])) return op_fail(DE_UNSATISFIED);
                q += 2;
                continue;
            }
            const Fr cb = fr_to_canonical(rv);
            if (cb.v[2] | cb.v[3] | cb.v[4] | cb.v[5] | cb.v[6] | cb.v[7]) return op_fail_msg(DE_PANIC, 0, DM_BRILLIG_PANIC, BP_U64);
            const uint64_t base = (uint64_t)cb.v[1] << 32 | cb.v[0];
            for (uint32_t k = 0; k < n; k++, q += 2) {
                if (base + k >= vm.n_mem) return op_fail_msg(DE_PANIC, 0, DM_BRILLIG_PANIC, BP_OUT_MEM_OOB);
                if (!p.insert(q[0], vm.mem_get((uint32_t)(base + k)), q[1])) return op_fail(DE_UNSATISFIED);
            }
        }
        return op_ok();
    }
    if (vm.status == 2) {  // BrilligFunctionFailed { message, call_stack } (brillig.rs:113-125): call stack + failing pc
        if (res) {
            uint32_t n = 0;
            for (uint32_t k = 0; k < vm.n_cs && n < 15u; k++) res->call_stack[n++] = vm.words[(uint64_t)k * vm.Bp + vm.j];
            res->call_stack[n++] = vm.pc;
            res->n_call_stack = n;
#pragma unroll
            for (int k = 0; k < 8; k++) res->val[k] = vm.val.v[k];
        }
        return op_fail_msg(DE_BRILLIG_FAILED, 0, vm.code, vm.x0, vm.val.v[0]);
    }
    if (vm.status == 3) return op_fail_msg(DE_WAIT_FOREIGN_CALL, 0, 0, vm.pc);
    if (vm.status == 5) return op_fail_msg(DE_PANIC, 0, vm.code, vm.x0);
    return op_fail_msg(DE_PANIC, 0, DM_BRILLIG_PANIC, vm.code, vm.x0);
}

}  // namespace acvm
