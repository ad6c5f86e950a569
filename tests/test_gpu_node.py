"""The node-level driver (acvm_node_*, node.cpp): one call = split over devices + tiles + pinned double-buffered uploads + the exact path
beside the next tile. The test box has ONE GPU: "two devices" is device 0 listed twice -- two handles driven by two host threads on one
device, which is also the header's threading contract (include/acvm_amd.h). Everything is compared with one plain batch of the same
instances and, through results / kept witnesses / digests, with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def plain_batch(data, ids, values, B, keep):
    import acvm_amd
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    batch.set_initial_witness(values)
    batch.solve()
    res = [r.as_tuple() for r in batch.results()]
    dig = batch.digest()
    kept = np.zeros((B, len(keep), 32), dtype=np.uint8)
    asg = np.zeros((B, len(keep)), dtype=np.uint8)
    for k, w in enumerate(keep):
        v, a = batch.witness(w)
        kept[:, k] = v
        asg[:, k] = a
    batch.free()
    return res, kept, asg, dig


@pytest.mark.parametrize("devices,tile,B", [([0], 256, 1000), ([0, 0], 192, 1000), ([0, 0], 64, 130), ([0, 0, 0], 512, 700), ([0], 512, 3),
                                            ([0] * 8, 128, 2500)])  # (eight lanes: the shape acvm_node_new has on an 8-GPU node, rehearsed on one device)
def test_node_equals_one_batch_and_oracle(oracle, devices, tile, B):
    """mixed circuit with edge-case instances: the flagged instances of every tile take the asynchronous exact path; partial last tiles"""
    import acvm_amd
    from acvm_amd import synth
    circ, ids = synth.mixed_circuit(500, seed=0x40DE0001)
    values = synth.witness_batch(B, seed=0x40DE0001, edge_cases=True)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    keep = gc.witness_set("return_values") + [ids[0], 7]
    want = plain_batch(data, ids, values, B, keep)
    node = acvm_amd.Node(gc, ids, keep=keep, devices=devices, tile=tile)
    for _ in range(2):  # the handle is reusable
        not_solved, res, kept, asg, dig = node.solve(values, B)
        assert [r.as_tuple() for r in res] == want[0]
        assert not_solved == sum(1 for r in want[0] if r[0] != 0)
        assert np.array_equal(asg, want[2]) and np.array_equal(kept, want[1])
        assert np.array_equal(dig, want[3])
    st = node.stats()
    assert st["n_devices"] == len(devices) and sum(st["tiles"]) >= (B + tile - 1) // tile
    assert all(st["async_exact"])
    # the circuit is levelised ONCE per node, whatever the number of lanes (the plan is immutable and shared: round 6; until then 1 + lanes times)
    assert st["plans_built"] == 1 and gc.plans_built() == 1 and st["plan_ms"] > 0 and st["create_ms"] >= st["plan_ms"] and st["host_rss_bytes"] > 0
    # (a partial last tile solves its own instances only: the lanes behind them are dead, not copies of an instance)
    assert 0 < sum(st["exact_instances"]) <= B
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, values, B)
    for j in range(0, B, 7):
        assert res[j].as_tuple() == ores[j].as_tuple()
        assert bytes(dig[j]) == oracle.witness_map_digest(oasg[j], ovals[j])
        if res[j].message or ores[j].message:
            assert res[j].message == ores[j].message
    node.free()
    again = acvm_amd.Node(gc, ids, keep=keep, devices=devices, tile=tile)  # the circuit handle still holds the plan: a second node plans nothing
    assert again.stats()["plans_built"] == 0 and gc.plans_built() == 1
    again.free()


def test_node_with_slot_reuse_and_null_outputs(oracle):
    import acvm_amd
    from acvm_amd import synth
    circ, ids = synth.mixed_circuit(400, seed=0x40DE0002)
    B = 300
    values = synth.witness_batch(B, seed=0x40DE0002, edge_cases=True)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    keep = gc.witness_set("return_values")
    want = plain_batch(data, ids, values, B, keep)
    node = acvm_amd.Node(gc, ids, keep=keep, devices=[0, 0], tile=128, reuse_slots=True)
    not_solved, res, kept, asg, dig = node.solve(values, B)
    assert [r.as_tuple() for r in res] == want[0] and np.array_equal(dig, want[3]) and np.array_equal(kept, want[1]) and np.array_equal(asg, want[2])
    n2, r2, k2, a2, d2 = node.solve(values, B, results=False, kept=False, digests=False)  # only the count
    assert n2 == not_solved and r2 is None and k2 is None and d2 is None
    node.free()


def test_node_foreign_calls_stay_synchronous(oracle):
    """a circuit with a pending foreign call cannot defer its exact lanes (the caller must answer): the handle stays synchronous and the
    instances report RequiresForeignCall"""
    import acvm_amd
    from acvm_amd.acir import Brillig, Circuit, Expression as E
    from acvm_amd.synth import values_from_rows
    circ = Circuit(3, [Brillig(inputs=[E.from_witness(1)], outputs=[2], bytecode=[("ForeignCall", "f", [("Register", 0)], [("Register", 0)]), ("Stop",)])])
    node = acvm_amd.Node(acvm_amd.Circuit(circ.to_bytes()), [1], keep=[2], devices=[0], tile=64)
    not_solved, res, kept, asg, dig = node.solve(values_from_rows([[5], [6], [7]]), 3)
    assert not_solved == 3 and all(r.status == acvm_amd.STATUS_REQUIRES_FOREIGN_CALL for r in res) and not any(node.stats()["async_exact"])
    node.free()


def test_node_lookup_tables_are_shared_refcounted_and_releasable():
    """eight handles of a circuit with Pedersen opcodes on one device build the device's lookup tables once (under the device's lock, on a build
    stream), hold them while they live, and acvm_device_release_tables gives the memory back afterwards -- and only afterwards"""
    import gc
    import acvm_amd
    from acvm_amd import synth
    gc.collect()  # (handles of earlier tests that are only waiting for the collector hold the tables too)
    circ, ids = synth.arith_pedersen_circuit(200, 3)
    gc = acvm_amd.Circuit(circ.to_bytes())
    values = synth.witness_batch(600, seed=0x40DE0011, edge_cases=True)
    node = acvm_amd.Node(gc, ids, keep=gc.witness_set("return_values"), devices=[0] * 8, tile=64)
    first = node.solve(values, 600)
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.release_tables(0)  # in use
    node.free()
    freed = acvm_amd.release_tables(0)
    assert freed > (1 << 28)  # the 16-bit windows alone are 268 MB
    assert acvm_amd.release_tables(0) == 0  # nothing left
    node = acvm_amd.Node(gc, ids, keep=gc.witness_set("return_values"), devices=[0, 0], tile=128)  # rebuilt on demand, same results
    again = node.solve(values, 600)
    assert [r.as_tuple() for r in again[1]] == [r.as_tuple() for r in first[1]] and np.array_equal(again[4], first[4])
    node.free()
    acvm_amd.tuning_set("tables_keep", 0)  # the last handle of the device frees the tables itself
    try:
        node = acvm_amd.Node(gc, ids, keep=gc.witness_set("return_values"), devices=[0], tile=128)
        node.solve(values, 600)
        node.free()
        assert acvm_amd.release_tables(0) == 0
    finally:
        acvm_amd.tuning_set("tables_keep", 1)


def test_node_auto_tile_and_empty_batch():
    import acvm_amd
    from acvm_amd import synth
    circ, ids = synth.arithmetic_circuit(300, seed=3)
    gc = acvm_amd.Circuit(circ.to_bytes())
    node = acvm_amd.Node(gc, ids, keep=gc.witness_set("return_values"))
    assert node.tile == 1 << 17 and node.n_devices == acvm_amd.device_count()
    assert node.solve(b"", 0)[0] == 0
    values = synth.witness_batch(100, seed=3, edge_cases=False)
    not_solved, res, kept, asg, dig = node.solve(values, 100)
    assert not_solved == 0 and asg.all()
    node.free()


def test_two_batch_handles_from_two_host_threads(oracle):
    """the threading contract of include/acvm_amd.h through the plain batch API: independent handles on one device, each driven by its own host
    thread at the same time (ctypes releases the GIL inside the calls), give what they give one after the other"""
    import threading
    import acvm_amd
    from acvm_amd import synth
    circ, ids = synth.mixed_circuit(700, seed=0x40DE0007)
    data = circ.to_bytes()
    B = 500
    vals = [synth.witness_batch(B, seed=0x40DE0007 + k, edge_cases=True) for k in range(2)]
    want = [plain_batch(data, ids, vals[k], B, [ids[0]]) for k in range(2)]
    got = [None, None]

    def run(k):
        acvm_amd.set_device(0)
        for _ in range(3):
            b = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids, fold_digest=bool(k))
            b.set_initial_witness(vals[k])
            b.solve()
            got[k] = ([r.as_tuple() for r in b.results()], b.digest())
            b.free()

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        assert got[k][0] == want[k][0] and np.array_equal(got[k][1], want[k][3])


def test_node_refuses_what_it_cannot_serve():
    """acvm_node_new fails loudly -- a null handle and the reason in acvm_last_error -- for a device that is not there, a duplicate initial id and
    slot reuse on a circuit the level kernels do not cover entirely; nothing is left half-built. A kept witness beyond the circuit is not an
    error: it is simply never assigned (WitnessMap::get -> None)."""
    import acvm_amd
    from acvm_amd import synth
    from acvm_amd.acir import Brillig, Circuit, Expression as E
    circ, ids = synth.arithmetic_circuit(200, seed=5)
    gc = acvm_amd.Circuit(circ.to_bytes())
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.Node(gc, ids, devices=[acvm_amd.device_count() + 7], tile=64)
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.Node(gc, ids + [ids[0]], devices=[0], tile=64)
    for flags in ({}, {"reuse_slots": True}):
        node = acvm_amd.Node(gc, ids, keep=[1 << 30, ids[0], 0xFFFFFFFE], devices=[0], tile=64, **flags)
        values = synth.witness_batch(100, seed=5, edge_cases=True)
        not_solved, res, kept, asg, dig = node.solve(values, 100)
        assert not asg[:, 0].any() and not asg[:, 2].any() and asg[:, 1].all() and not kept[:, 0].any() and not kept[:, 2].any()
        assert np.array_equal(kept[:, 1], plain_batch(circ.to_bytes(), ids, values, 100, [ids[0]])[1][:, 0])
        node.free()
    fc = Circuit(3, [Brillig(inputs=[E.from_witness(1)], outputs=[2], bytecode=[("ForeignCall", "f", [("Register", 0)], [("Register", 0)]), ("Stop",)])])
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.Node(acvm_amd.Circuit(fc.to_bytes()), [1], keep=[2], devices=[0], tile=64, reuse_slots=True)
    node = acvm_amd.Node(gc, ids, devices=[0], tile=64)  # and the library is still usable
    assert node.solve(synth.witness_batch(10, seed=5, edge_cases=False), 10)[0] == 0
    node.free()
