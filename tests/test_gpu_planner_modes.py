"""Every planner / scheduler mode of csrc/tuning.hpp against the oracle, bit for bit (results, assigned sets, witnesses, digests), on
config-5 style circuits that hold every kernel class (acvm_amd.synth.mixed_circuit) -- plain, with slot reuse, with the folded digest and
through the exact path. The modes only change WHERE and WHEN a record runs; none may change a result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = [
    {},
    {"scale": 0}, {"pairs": 0}, {"chains": 0}, {"max_tails": 1}, {"max_tails": 8}, {"inv_epoch": 1}, {"inv_epoch": 9}, {"inv_latency": 0}, {"inv_latency": 3},
    {"heavy_epoch": 4, "heavy_latency": 4}, {"pedersen_latency": 6}, {"pedersen_epoch": 1, "pedersen_latency": 0}, {"pedersen_epoch": 5}, {"digest_epoch": 1}, {"digest_epoch": 32},
    {"range_fuse": 0}, {"range_merge": 0}, {"range_fuse": 0, "range_merge": 0}, {"brillig_inline": 0}, {"sl_lane": 1}, {"hash_chain": 0}, {"light_fuse": 0}, {"pedersen_waves": 1}, {"pedersen_waves": 4}, {"pedersen_prio": 0}, {"pedersen_window_bits": 0}, {"pedersen_window_bits": 0, "pedersen_waves": 1},
    {"relax": 0}, {"pedersen_bundle": 2}, {"pedersen_bundle": 2, "pedersen_epoch": 8}, {"pedersen_bundle": 0}, {"inv_chunk": 3}, {"inv_chunk": 1000, "inv_epoch": 9}, {"byte_plane": 0},
    {"overlap": 0}, {"heavy_streams": 0}, {"scale": 0, "pairs": 0, "range_fuse": 0, "range_merge": 0, "brillig_inline": 0, "light_fuse": 0, "overlap": 0},
]


@pytest.mark.parametrize("mode", MODES, ids=lambda m: ",".join(f"{k}={v}" for k, v in m.items()) or "default")
def test_mode_is_bit_exact(oracle, mode):
    import acvm_amd
    from acvm_amd import synth
    seed = 0x300D0000 + MODES.index(mode)
    circ, ids = synth.mixed_circuit(900, seed=seed, heavy=True, blocks=4, cells=16)
    B = 130
    values = synth.witness_batch(B, seed=seed, edge_cases=True)
    data = circ.to_bytes()
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, values, B)
    odig = [oracle.witness_map_digest(oasg[j], ovals[j]) for j in range(B)]
    with acvm_amd.tuning(**mode):
        for variant in ("plain", "fold", "reuse", "exact"):
            gc = acvm_amd.Circuit(data)
            kw = {}
            if variant == "reuse":
                kw.update(reuse_slots=True, keep=gc.witness_set("return_values"))
            if variant == "fold":
                kw.update(fold_digest=True)
            try:
                batch = acvm_amd.Batch(gc, B, ids, **kw)
            except acvm_amd.AcvmError:  # slot reuse refuses a circuit whose plan is truncated (an opcode no generic instance can execute)
                assert variant == "reuse"
                continue
            if variant == "exact":
                batch.set_force_slow_path(True)
            batch.set_initial_witness(values)
            batch.solve()
            gres = batch.results()
            for j in range(B):
                assert gres[j].as_tuple() == ores[j].as_tuple(), (mode, variant, j, gres[j].as_tuple(), ores[j].as_tuple())
            dig = batch.digest()
            assert all(bytes(dig[j]) == odig[j] for j in range(B)), (mode, variant)
            if variant != "reuse":
                gasg, gvals = batch.witness_map()
                nw = min(oasg.shape[1], gasg.shape[1])
                assert np.array_equal(oasg[:, :nw], gasg[:, :nw]) and np.array_equal(ovals[:, :nw], gvals[:, :nw]), (mode, variant)
            batch.free()


def test_tuning_keys_round_trip():
    import acvm_amd
    keys = acvm_amd.tuning_keys()
    assert "scale" in keys and "brillig_steps_max_log2" in keys and len(keys) == len(set(keys))
    for k in keys:
        v = acvm_amd.tuning_get(k)
        acvm_amd.tuning_set(k, v)
        assert acvm_amd.tuning_get(k) == v
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.tuning_set("no_such_mode", 1)
