"""Host logic without a GPU: the static planner (levelisation, gate classification, algorithmic-byte accounting of SURVEY 8d,
refusal of opcodes without a kernel) through the host-only C-ABI entry acvm_circuit_plan_stats."""
import pytest

import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import P, BlackBoxFuncCall as BB, Brillig, Circuit, Expression as E, FunctionInput as FI, MemoryInit, MemoryOp, PermutationSort


def stats(circ, ids):
    return acvm_amd.Circuit(circ.to_bytes()).plan_stats(ids)


def test_arithmetic_mix_accounting():
    circ, ids = synth.arithmetic_circuit(2000, seed=0xAC1D0002)
    st = stats(circ, ids)
    assert st["n_opcodes"] == 2000 and st["n_fast_gates"] + st["n_dyn_gates"] == 2000
    assert 60 <= st["n_dyn_gates"] <= 140          # 5 % unknown-in-mul gates
    assert st["truncated_at"] == 0xFFFFFFFF and st["n_other_records"] == 0
    # 32 B x (distinct known operands + written witness): between 64 and 128 B per gate, about 102 B on this mix
    per_gate = st["algorithmic_bytes_per_instance"] / 2000
    assert 96 <= per_gate <= 108
    assert st["algorithmic_bytes_per_instance"] == st["arith_algorithmic_bytes_per_instance"] + st["dyn_algorithmic_bytes_per_instance"]


def test_gate_pairs_and_inverse_slots():
    """Config-2 shape: about half of the gates run behind their producer in the same wave (up to five records behind a SOLVE
    host, chains host -> tail -> tail's consumer included; no inversion gates), and the inverse table needs far fewer rows than
    there are inversion gates (rows are reused)."""
    circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
    st = stats(circ, ids)
    assert 4500 <= st["n_gate_pairs"] <= 6000
    assert 0 < st["n_inverse_slots"] < st["n_dyn_gates"] // 2
    # a chain: every gate's only fresh operand is its predecessor: runs of up to six gates share a wave
    chain, cids = synth.arithmetic_circuit(200, seed=7, chain=True, mix=(47, 32, 21, 0))
    assert stats(chain, cids)["n_gate_pairs"] >= 60


def test_chain_has_one_level_per_gate():
    circ, ids = synth.arithmetic_circuit(200, seed=7, chain=True)
    assert stats(circ, ids)["n_levels"] >= 200


def test_exact_bytes_of_small_circuit():
    # w3 = w1*w2 (reads 2, writes 1 = 96 B); assert w3 - w1*w2 == 0 (reads 3 = 96 B); RANGE(w3) 32 B; sha256 of 2 bytes: (2 + 32) x 32 B
    ops = [E([(1, 1, 2)], [(P - 1, 3)], 0), E([(1, 1, 2)], [(P - 1, 3)], 0), BB("RANGE", {"input": FI(3, 200)}),
           BB("SHA256", {"inputs": [FI(1, 8), FI(2, 8)], "outputs": list(range(4, 36))})]
    st = stats(Circuit(35, ops), [1, 2])
    assert st["algorithmic_bytes_per_instance"] == 96 + 96 + 32 + 34 * 32
    assert st["class_algorithmic_bytes_per_instance"] == [32, 34 * 32, 0, 0]
    assert st["n_levels"] == 2 and st["n_fast_gates"] == 2 and st["n_other_records"] == 2
    # the two message bytes are initial witnesses: each gets a byte plane (a 4-byte copy for the hash kernel; the byte figures keep the reference's unit)
    assert st["n_byte_planes"] == 2 and st["n_byte_plane_reads"] == 2
    with acvm_amd.tuning(byte_plane=0):
        assert stats(Circuit(35, ops), [1, 2])["n_byte_planes"] == 0


def test_memory_blocks_are_chained_in_program_order():
    ops = [MemoryInit(0, [1, 2]), MemoryOp(0, E.constant(1), E.constant(0), E.from_witness(1)), MemoryOp(0, E.constant(0), E.constant(0), E.from_witness(3)),
           MemoryOp(0, E.constant(1), E.constant(1), E.from_witness(3)), MemoryOp(0, E.constant(0), E.constant(1), E.from_witness(4))]
    st = stats(Circuit(4, ops), [1, 2])
    assert st["n_levels"] == 5 and st["truncated_at"] == 0xFFFFFFFF


def test_generic_instance_failure_truncates_the_level_plan():
    # opcode 1 needs w9, which nothing assigns: every instance is handed to the exact kernels from opcode 1 on
    st = stats(Circuit(9, [E([], [(1, 1), (P - 1, 2)], 0), BB("RANGE", {"input": FI(9, 8)}), E([], [(1, 2), (P - 1, 3)], 0)]), [1])
    assert st["truncated_at"] == 1 and st["n_fast_gates"] == 1
    # two unknowns in one expression
    st = stats(Circuit(4, [E([], [(1, 1), (1, 2), (1, 3)], 0)]), [1])
    assert st["truncated_at"] == 0


def test_every_opcode_kind_is_planned():
    st = stats(Circuit(3, [PermutationSort([[E.from_witness(1)], [E.from_witness(1)], [E.from_witness(1)]], 1, [2, 3, 4], [0])]), [1])
    assert st["n_other_records"] == 1 and st["truncated_at"] == 0xFFFFFFFF


def test_ecdsa_is_planned():
    op = BB("EcdsaSecp256k1", {"public_key_x": [FI(1, 8)] * 32, "public_key_y": [FI(1, 8)] * 32, "signature": [FI(1, 8)] * 64,
                               "hashed_message": [FI(1, 8)] * 32, "output": 2})
    st = stats(Circuit(3, [op]), [1])
    assert st["n_other_records"] == 1 and st["truncated_at"] == 0xFFFFFFFF


def test_brillig_and_foreign_calls_are_planned():
    br = Brillig(inputs=[E.from_witness(1)], outputs=[2], bytecode=[("ForeignCall", "f", [("Register", 0)], [("Register", 0)]), ("Stop",)])
    st = stats(Circuit(2, [br]), [1])
    assert st["n_other_records"] == 1 and st["class_algorithmic_bytes_per_instance"][3] == 64


def test_projective_witnesses_only_where_arithmetic_gates_are_the_only_users():
    """plan.cpp keeps a witness as scale x value when that makes its gate cheaper -- but never an initial witness and never one a
    non-Arithmetic opcode mentions (those kernels read plain values)."""
    circ, ids = synth.arithmetic_circuit(2000, seed=0xAC1D0002)
    with acvm_amd.tuning(relax=0):
        st = stats(circ, ids)
        assert 1500 <= st["n_scaled_witnesses"] <= 2000          # random coefficients: almost every gate has one to remove
        # all coefficients +-1: nothing to gain, nothing is scaled
        ops = [E([(1, 1, 2)], [(P - 1, 3)], 0), E([], [(1, 3), (1, 1), (P - 1, 4)], 0), E([(P - 1, 3, 4)], [(1, 5)], 7)]
        assert stats(Circuit(5, ops), [1, 2])["n_scaled_witnesses"] == 0
        # w3 = 5 w1 w2 would be scaled, but RANGE reads it; w4 = 3 w3 + 7 w1 has no other user and is scaled
        ops2 = [E([(5, 1, 2)], [(P - 1, 3)], 0), BB("RANGE", {"input": FI(3, 200)}), E([], [(3, 3), (7, 1), (P - 1, 4)], 0)]
        assert stats(Circuit(4, ops2), [1, 2])["n_scaled_witnesses"] == 1
    # relaxed rows (the default): every witness that only Arithmetic gates read is stored as some representative below 2^256 and takes the
    # same road out (a scale, 1 if nothing was gained); the RANGE operand and the initial witnesses stay canonical
    st = stats(Circuit(5, ops), [1, 2])
    assert st["n_scaled_witnesses"] == 3 and st["n_gate_out_canon"] == 0 and st["n_gate_out_asis"] == 3
    st = stats(Circuit(4, ops2), [1, 2])
    assert st["n_scaled_witnesses"] == 1 and st["n_gate_out_canon"] == 1 and st["n_gate_out_asis"] == 1


def test_relaxed_rows_bounds_and_modes():
    """The planner's bound bookkeeping (gate_record.hpp units of p / 256): a chain of +-1 additions grows by the operand's bound per gate and
    must ask for a reduction before 2^256; the denominators of SOLVE_DYN gates are stored canonical for the inversion kernel's zero test."""
    circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
    st = stats(circ, ids)
    assert st["n_gate_out_asis"] + st["n_gate_out_weak"] + st["n_gate_out_canon"] == 10000
    assert st["n_gate_out_asis"] > 9000 and 0 < st["n_gate_out_weak"] < 500
    assert 300 < st["n_gate_out_canon"] <= st["n_dyn_gates"]     # one canonical producer per distinct denominator
    assert st["max_gate_bound"] < 169 * 256                      # fr29_weak takes anything below 2^261 = 169 p
    chain, cids = synth.arithmetic_circuit(400, seed=5, chain=True, mix=(0, 100, 0, 0))
    st = stats(chain, cids)
    assert st["n_gate_out_weak"] >= 50
    with acvm_amd.tuning(relax=0):
        st = stats(circ, ids)
        assert st["n_gate_out_asis"] == 0 and st["n_gate_out_weak"] == 0 and st["n_gate_out_canon"] == 10000


# ---------------------------------------------------------------------------------------------------------------- the planner, pass by pass
# plan.cpp is a sequence of passes over an explicit intermediate state (struct Planner: init -> in-order program -> pins -> replay -> gate pairs
# -> inverse slots -> digest leaves -> range fusing -> hash chains -> range merging -> order + dependencies -> rows -> layout). With tuning
# plan_validate = 1 every pass checks what it promises before the next one runs; a violation is an error of the call.
def _corpus():
    import circuit_corpus as cc
    return cc


def test_every_pass_keeps_its_promise_on_the_whole_corpus():
    cc = _corpus()
    n = 0
    for name, data, ids in cc.corpus(big=True):
        gc = acvm_amd.Circuit(data)
        if ids is None:
            ids = gc.witness_set("circuit_arguments")
        keep = gc.witness_set("return_values")
        for mode in cc.PLANNER_MODES:
            with acvm_amd.tuning(plan_validate=1, **mode):
                for kw in ({}, {"fold_digest": True}, {"reuse_slots": True, "keep": keep}):
                    try:
                        gc.plan_stats(ids, **kw)
                    except acvm_amd.AcvmError as e:
                        assert "slot reuse needs" in str(e), (name, mode, kw, str(e))  # the only refusal there is; never "plan invariant violated"
                    n += 1
    assert n >= 2500


@pytest.mark.parametrize("k,pass_name", [(1, "fuse_gate_pairs"), (2, "assign_inverse_slots"), (3, "lay_out")])
def test_a_broken_invariant_is_caught_by_its_pass(k, pass_name):
    """plan_validate = 100 + k breaks invariant k on purpose (a gate marked fused at its own level; two live inverses in one row of the table; a wave
    program missing from the level lists): the pass's check names itself"""
    circ, ids = synth.arithmetic_circuit(400, seed=3, mix=(30, 30, 20, 20))
    gc = acvm_amd.Circuit(circ.to_bytes())
    with acvm_amd.tuning(plan_validate=100 + k):
        with pytest.raises(acvm_amd.AcvmError, match="plan invariant violated after pass " + pass_name):
            gc.plan_stats(ids)
    with acvm_amd.tuning(plan_validate=1):
        assert gc.plan_stats(ids)["n_opcodes"] == 400


def test_plan_is_built_once_per_option_set_and_shared():
    """acvm_circuit_t keeps its plans: handles (and the host-only entry points) that ask for the same initial ids, options and tuning get the same
    immutable plan; another option set, or another tuning, is another plan (the reference's callers build one opcode list per circuit:
    acvm_js/src/execute.rs:60-119)"""
    circ, ids = synth.mixed_circuit(600, seed=11, blocks=4, cells=16)
    gc = acvm_amd.Circuit(circ.to_bytes())
    keep = gc.witness_set("return_values")
    assert gc.plans_built() == 0
    a = gc.plan_stats(ids)
    assert gc.plans_built() == 1
    assert gc.plan_stats(ids) == a and gc.check_schedule(ids)["ok"] and gc.plans_built() == 1
    gc.plan_stats(ids, reuse_slots=True, keep=keep)
    assert gc.plans_built() == 2
    gc.plan_stats(ids, reuse_slots=True, keep=keep)
    gc.plan_stats(list(reversed(ids)))  # (another order of the same ids is another key: the import layout follows it)
    assert gc.plans_built() == 3
    with acvm_amd.tuning(inv_epoch=2):
        gc.plan_stats(ids)
    assert gc.plans_built() == 4
    gc.plan_stats(ids)
    assert gc.plans_built() == 4


def test_planner_refactoring_moved_no_word():
    """fingerprints of everything the planner emits (acvm_debug_plan_fingerprint), pinned for three circuits: tools/plan_fingerprint.py compares
    the whole corpus x modes before and after a change of plan.cpp; these pins catch an accidental change in the suite"""
    import hashlib
    circ, ids = synth.arithmetic_circuit(2000, seed=0xAC1D0002)
    fp = acvm_amd.Circuit(circ.to_bytes()).plan_fingerprint(ids)
    assert len(fp) >= 60
    h = hashlib.sha256(b"".join(int(x).to_bytes(8, "little") for x in fp)).hexdigest()[:16]
    circ2, ids2 = synth.mixed_circuit(900, seed=0xAC1D0005, heavy=True, blocks=4, cells=16)
    gc2 = acvm_amd.Circuit(circ2.to_bytes())
    h2 = hashlib.sha256(b"".join(int(x).to_bytes(8, "little") for x in gc2.plan_fingerprint(ids2, reuse_slots=True, keep=gc2.witness_set("return_values")))).hexdigest()[:16]
    assert (h, h2) == PLAN_PINS, (h, h2)


PLAN_PINS = ("dcf4da57cf56e6e0", "ff690a45bee30aee")
