"""The level schedule proved free of hazards WITHOUT a GPU (acvm_amd/csrc/schedule_check.cpp through acvm_circuit_check_schedule).

One solve is enqueued on up to six streams (gate levels + light records, inversion batches, three heavy lanes, the digest lane) with
partial waits, witness rows recycled under slot reuse, inverse rows reused and rows whose representation depends on the consumer. The
reference executes an instance's opcodes strictly in order (acvm/src/pwg/mod.rs:236-303), so a missing cross-stream edge is a
timing-dependent wrong witness that a parity test on one box need not see. The checker derives every launch's reads and writes from the
record words (not from the planner's dependency tables), builds happens-before from stream order + event edges over the SAME step
list that batch_schedule.cpp enqueues, and asserts order, value identity, canonical rows for readers outside the gate kernels and the
relaxed-row bounds. Covered here: every circuit shape of the GPU suite (tests/circuit_corpus.py), tools/t_fuzz.py's wild circuits, the
reference's own serialised circuits, x every planner mode x {plain, folded digest, slot reuse, caller-supplied solver} x two tile
sizes; the 10^6-opcode config-5 circuit; and mutations -- a dropped wait must be named with the resource and the two launches."""
import re

import pytest

import acvm_amd
import circuit_corpus as cc
from acvm_amd.acir import P, BlackBoxFuncCall as BB, Circuit, Expression as E, FunctionInput as FI

VARIANTS = [("plain", {}), ("fold", {"fold_digest": True}), ("reuse", {"reuse_slots": True}), ("solver", {"host_solver": True})]
CORPUS = cc.corpus()
BIG = [c for c in cc.corpus(big=True) if c[0] in ("config2_10k", "north_star_10k_8", "mixed_20k")]


def _check_all(gc, ids, sizes=(64, 1 << 17), variants=VARIANTS):
    keep = gc.witness_set("return_values")
    n = 0
    for _, kw in variants:
        kw = dict(kw)
        if kw.get("reuse_slots"):
            kw["keep"] = keep
        for B in sizes:
            try:
                r = gc.check_schedule(ids, n_instances=B, **kw)
            except acvm_amd.AcvmError as e:  # slot reuse refuses truncated plans and foreign calls; nothing else may be refused
                assert kw.get("reuse_slots") and "slot reuse" in str(e), str(e)
                continue
            assert r["ok"], r["report"]
            assert r["n_launches"] >= 2 and r["n_findings"] == 0
            n += 1
    return n


@pytest.mark.parametrize("name,data,ids", CORPUS, ids=[c[0] for c in CORPUS])
def test_schedule_is_hazard_free_in_every_mode(name, data, ids):
    gc = acvm_amd.Circuit(data)
    if ids is None:
        ids = gc.witness_set("circuit_arguments")
    for mode in cc.PLANNER_MODES:
        with acvm_amd.tuning(**mode):
            assert _check_all(gc, ids) >= 6, (name, mode)


@pytest.mark.parametrize("name,data,ids", BIG, ids=[c[0] for c in BIG])
def test_full_size_circuits(name, data, ids):
    """config 2 (10^4 gates), the north-star shape (10^4 gates + 8 Pedersen) and a 20k-opcode config-5 mix, default mode and the modes that move the most"""
    gc = acvm_amd.Circuit(data)
    for mode in ({}, {"inv_epoch": 1}, {"heavy_epoch": 4, "heavy_latency": 4}, {"digest_epoch": 1}, {"heavy_streams": 0}, {"pairs": 0, "scale": 0}):
        with acvm_amd.tuning(**mode):
            _check_all(gc, ids)


def test_config5_circuit_at_circuit_size():
    """BASELINE config 5: the 10^6-opcode circuit, plain and with slot reuse + folded digest, at tiles of 4 096 and of 8 192 (the bench leg's) --
    millions of accesses over ~10^6 rows, every one ordered"""
    circ, ids = cc.config5_circuit()
    gc = acvm_amd.Circuit(circ.to_bytes())
    keep = gc.witness_set("return_values")
    for kw in ({}, {"reuse_slots": True, "keep": keep}):  # (slot reuse folds the digest: the digest lane and its leaves are in; the plain plan has neither)
        r = gc.check_schedule(ids, n_instances=4096, **kw)
        assert r["ok"], r["report"]
        assert r["n_records"] >= 900_000 and r["n_accesses"] >= 3_000_000
    assert gc.plans_built() == 2  # one plan per option set, shared by whoever asks again
    assert gc.check_schedule(ids, n_instances=8192, reuse_slots=True, keep=keep)["ok"] and gc.plans_built() == 2


# ---------------------------------------------------------------------------------------------------------------- mutations
def _mutations(gc, ids, **kw):
    base = gc.check_schedule(ids, **kw)
    assert base["ok"], base["report"]
    out = []
    for k in range(base["n_waits"]):
        out.append(gc.check_schedule(ids, drop_wait=k, **kw))
    return base, out


LAUNCH = r"launch #\d+ \[[a-z_+ ]+, level \d+, stream \w+, step \d+\]"


def test_dropped_wait_names_the_row_and_the_two_launches():
    """hash -> gate: the gate level reads a digest byte written on the hash lane. The schedule has three waits (the lane behind the start of the
    solve, the main stream behind the lane's level, the join at the end); without the second one the checker must name the witness row, the
    gate launch and the hash launch."""
    ops = [BB("SHA256", {"inputs": [FI(1, 8), FI(2, 8)], "outputs": list(range(3, 35))}), E([(1, 3, 4)], [(P - 1, 35)], 0), E([(1, 35, 1)], [(P - 1, 36)], 0)]
    gc = acvm_amd.Circuit(Circuit(36, ops).to_bytes())
    # (a circuit of byte hashes and gates: the hash runs on its lane because the circuit has main-stream work)
    base, muts = _mutations(gc, [1, 2])
    assert base["n_waits"] == 3
    # wait 0, the lane behind the start of the solve: the hash could flag an instance before the event words are reset, and read the byte planes
    # of its two inputs before the import has written them
    rep = muts[0]["report"]
    assert not muts[0]["ok"] and "EVENT WORDS: launch #2 [hash_coop, level 1, stream lane0" in rep, rep
    assert re.search(r"RAW: record at prog\[0\] \(kind 3, opcode 0\) reads byte plane 0 in " + LAUNCH + r" but its writer launch #0 \[import of the initial witnesses", rep), rep
    # wait 1, the gate level behind the hash lane: the row, the reader and the writer by name
    rep = muts[1]["report"]
    m = re.search(r"RAW: gate of opcode 1 reads witness row (\d+) in (" + LAUNCH + r") but its writer (" + LAUNCH + r") is not ordered before it", rep)
    assert not muts[1]["ok"] and m, rep
    assert int(m.group(1)) == 3 and "[gates, level 2, stream main" in m.group(2) and "[hash_coop, level 1, stream lane0" in m.group(3)
    assert muts[1]["n_findings"] == 2  # rows 3 and 4, nothing else
    # wait 2, the join at the end: the main stream already waited for that lane's only level
    assert muts[2]["ok"] and "redundant wait" in muts[2]["report"]


def test_dropped_wait_under_slot_reuse_is_a_war_on_the_recycled_row():
    """Slot reuse: a row is recycled while its old owner's readers sit on other streams (digest lane, hash lane, inversion stream). Dropping the
    wait that orders such a reader before the row's next writer must come out as WAR on that row."""
    circ, ids = cc.synth.mixed_circuit(900, seed=0xAC1D0005, heavy=True, blocks=4, cells=16)
    gc = acvm_amd.Circuit(circ.to_bytes())
    keep = gc.witness_set("return_values")
    base, muts = _mutations(gc, ids, reuse_slots=True, keep=keep)
    flagged = [m for m in muts if not m["ok"]]
    kinds = {}
    for m in flagged:
        first = m["report"].split("\n")[1]
        kinds[first.split(":")[0]] = kinds.get(first.split(":")[0], 0) + 1
        assert re.search(LAUNCH, first), first
    assert kinds.get("WAR", 0) >= 3 and kinds.get("RAW", 0) >= 10, kinds
    war = next(m["report"] for m in flagged if m["report"].split("\n")[1].startswith("WAR"))
    assert re.search(r"WAR: .* overwrites witness row \d+ in " + LAUNCH + r" but its reader " + LAUNCH + r" is not ordered before it", war), war
    # (a dropped wait that changes nothing is one whose event was already behind the stream when it was enqueued -- the report says so -- or one that
    # another wait covers: a lane's wait for the start of the solve when a later wait of that lane follows before its first launch, a join at the end
    # of the solve when another lane that is joined had itself waited for that lane's last level)
    assert len(flagged) * 10 >= len(muts) * 6


def test_mutations_across_shapes_and_modes():
    """every wait of several shapes x modes: each one is either needed (and then named) or provably redundant"""
    shapes = [c for c in CORPUS if c[0] in ("north_star_1000_4", "config4_grumpkin", "inverse_behind_pedersen", "hash_gate_hash_gate", "mixed_900_300d0007", "arith_inversions_40pct")]
    total = needed = 0
    for name, data, ids in shapes:
        gc = acvm_amd.Circuit(data)
        for mode in ({}, {"inv_epoch": 1, "inv_latency": 0}, {"heavy_epoch": 4, "heavy_latency": 4}, {"digest_epoch": 1}):
            with acvm_amd.tuning(**mode):
                for kw in ({}, {"fold_digest": True}):
                    base, muts = _mutations(gc, ids, **kw)
                    for m in muts:
                        total += 1
                        if not m["ok"]:
                            needed += 1
                            line = m["report"].split("\n")[1]
                            assert line.split(":")[0] in ("RAW", "WAR", "WAW", "JOIN", "EVENT WORDS", "VALUE"), line
    assert total >= 200 and needed * 2 >= total, (total, needed)


def test_heavy_records_on_the_main_stream_order_the_inversions():
    """Regression (round 6, found by this checker): with tuning heavy_streams = 0 the heavy records share the main stream, and an inversion batch
    whose denominator a Pedersen record had written there waited for nothing -- the 'main levels < L are done' mark was only recorded behind
    gate / light launches."""
    name, data, ids = next(c for c in CORPUS if c[0] == "inverse_behind_pedersen")
    gc = acvm_amd.Circuit(data)
    with acvm_amd.tuning(heavy_streams=0):
        for kw in ({}, {"fold_digest": True}, {"reuse_slots": True}):
            r = gc.check_schedule(ids, **kw)
            assert r["ok"], r["report"]
