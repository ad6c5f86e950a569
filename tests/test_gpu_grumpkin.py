"""GPU parity of FixedBaseScalarMul / Pedersen / SchnorrVerify: the reference's own golden vectors
(barretenberg_blackbox_solver/src/wasm/{scalar_mul.rs:72-97, pedersen.rs:38-54}, acvm_js/test/shared/{pedersen,
schnorr_verify,fixed_base_scalar_mul}.ts as committed fixtures) and seeded batches against the CPU oracle."""
import ctypes as C
import random

import numpy as np
import pytest

from acvm_amd.acir import P, BlackBoxFuncCall as BB, Circuit, FunctionInput as FI
from acvm_amd.synth import Q_GRUMPKIN, be32, grumpkin_circuit, grumpkin_rows, values_from_rows
from test_gpu_opcodes import both_paths, run_both

pytestmark = pytest.mark.gpu
GY = 0x0000000000000002CF135E7506A45D632D270D45F1181294833FC48D823F272C


def solve_one(circ_bytes, iw):
    import acvm_amd
    ids = sorted(iw)
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ_bytes), 1, ids)
    batch.set_initial_witness(b"".join(be32(iw[i]) for i in ids))
    n_bad = batch.solve()
    asg, vals = batch.witness_map()
    return n_bad, {w: int.from_bytes(vals[0, w].tobytes(), "big") for w in range(asg.shape[1]) if asg[0, w]}


@pytest.mark.parametrize("name", ["fixed_base_scalar_mul", "pedersen", "schnorr_verify"])
def test_acvm_js_fixture(golden, name):
    fx = golden["acvm_js"][name]
    iw = {int(k): int(v, 16) for k, v in fx["initialWitnessMap"].items()}
    n_bad, got = solve_one(bytes(fx["bytecode"]), iw)
    assert n_bad == 0
    assert got == {int(k): int(v, 16) for k, v in fx["expectedWitnessMap"].items()}


def test_rust_unit_vectors():
    # scalar_mul.rs:72-97 and pedersen.rs:38-54
    circ = Circuit(4, [BB("FixedBaseScalarMul", {"low": FI(1, 128), "high": FI(2, 128), "outputs": [3, 4]})]).to_bytes()
    _, w = solve_one(circ, {1: 1, 2: 0})
    assert (w[3], w[4]) == (1, GY)
    _, w = solve_one(circ, {1: 1, 2: 2})
    assert (w[3], w[4]) == (0x0702AB9C7038EEECC179B4F209991BCB68C7CB05BF4C532D804CCAC36199C9A9,
                            0x23F10E9E43A3AE8D75D24154E796AAE12AE7AF546716E8F81A2564F1B5814130)
    circ = Circuit(4, [BB("Pedersen", {"inputs": [FI(1, 254), FI(2, 254)], "domain_separator": 0, "outputs": [3, 4]})]).to_bytes()
    _, w = solve_one(circ, {1: 0, 2: 1})
    assert (w[3], w[4]) == (0x0C5E1DDECD49DE44ED5E5798D3F6FB7C71FE3D37F5BEE8664CF88A445B5BA0AF,
                            0x230294A041E26FE80B827C2EF5CB8784642BBAA83842DA2714D62B1F3C4F9752)


def test_fixed_base_batch_and_failures(oracle):
    r = random.Random(11)
    circ = Circuit(4, [BB("FixedBaseScalarMul", {"low": FI(1, 128), "high": FI(2, 128), "outputs": [3, 4]})])
    rows = [[r.randrange(1 << 128), r.randrange(1 << 125)] for _ in range(72)]
    rows[0] = [1 << 128, 0]
    rows[1] = [0, 1 << 200]
    rows[2] = [Q_GRUMPKIN & ((1 << 128) - 1), Q_GRUMPKIN >> 128]
    rows[3] = [(Q_GRUMPKIN - 1) & ((1 << 128) - 1), (Q_GRUMPKIN - 1) >> 128]
    rows[4] = [0, 0]  # UNPINNED: scalar 0 -> the encoding of the point at infinity is the oracle's recollection
    rows[5] = [P - 1, P - 1]
    ores, _ = both_paths(oracle, circ, [1, 2], rows)
    assert [ores[j].err for j in range(3)] == [oracle.E_BLACKBOX_FAILED] * 3 and ores[3].status == 0 and ores[4].status == 0


# hash_index != 0 and n == 0 compare the HIP kernels with the oracle's recollection of barretenberg only (SURVEY Appendix A.2:
# IV[k] = (k + 1) G and the empty commitment are not pinned by any reference vector): the ids say so
@pytest.mark.parametrize("n,ds", [(1, 0), (2, 0), (3, 0), (5, 0), pytest.param(2, 3, id="2-ds3-UNPINNED_hash_index"),
                                  pytest.param(0, 0, id="0-0-UNPINNED_empty_commitment")])
def test_pedersen_batch(oracle, n, ds):
    r = random.Random(20 + n + ds)
    ids = list(range(1, n + 1))
    circ = Circuit(n + 2, [BB("Pedersen", {"inputs": [FI(w, 254) for w in ids], "domain_separator": ds, "outputs": [n + 1, n + 2]})])
    rows = [[r.randrange(P) for _ in range(n)] for _ in range(70)]
    if n:
        rows[0] = [0] * n
        rows[1] = [P - 1] * n
        rows[2] = [1] * n
    both_paths(oracle, circ, ids, rows)


@pytest.mark.parametrize("counts", [(2, 2, 2), (1, 3, 2, 0, 5), tuple([2] * 11)], ids=["3x2", "ragged", "11x2"])
def test_pedersen_records_sharing_the_inversion(oracle, counts):
    """Several Pedersen records in ONE launch walked in lock-step by pedersen_bundle_level_kernel (one field inversion per chain step for up to
    eight records; pedersen_bundle=2 forces the path at test sizes): records of different input counts in a bundle, more records than one
    bundle holds, inputs 0 / 1 / p - 1, and a batch that ends inside a wave. Same witness maps as the oracle's and the exact kernels'."""
    import acvm_amd
    r = random.Random(sum(counts) + len(counts))
    n_in = max(counts) + 1
    ids = list(range(1, n_in + 1))
    ops, out = [], n_in
    for i, n in enumerate(counts):
        ops.append(BB("Pedersen", {"inputs": [FI(1 + (i + k) % n_in, 254) for k in range(n)], "domain_separator": 0, "outputs": [out + 1, out + 2]}))
        out += 2
    rows = [[r.randrange(P) for _ in range(n_in)] for _ in range(150)]
    rows[0] = [0] * n_in
    rows[1] = [P - 1] * n_in
    rows[2] = [1] * n_in
    with acvm_amd.tuning(pedersen_bundle=2):
        both_paths(oracle, Circuit(out, ops), ids, rows)


def test_config4_grumpkin_circuit(oracle):
    """accepting signatures and flipped-bit signatures (pinned: the Blake2s digest differs, schnorr_verify.ts)"""
    circ, ids = grumpkin_circuit()
    rows = grumpkin_rows(80)
    ores, stats = run_both(oracle, circ, ids, rows)
    run_both(oracle, circ, ids, rows[:40], force_slow=True)
    assert sum(1 for j in range(80) if ores[j].status == 0) >= 70


@pytest.mark.parametrize("B", [257, 321])
def test_batch_sizes_across_the_workgroup_boundary(oracle, B):
    """the record kernels run in workgroups of four waves (256 instances): a partly filled workgroup and a partly filled wave"""
    circ, ids = grumpkin_circuit()
    run_both(oracle, circ, ids, grumpkin_rows(B))


def test_schnorr_early_rejects_UNPINNED(oracle):
    """public key off the curve, s = 0, e = 0: HIP against the oracle's recollection of barretenberg's early exits (SURVEY A.3:
    no reference vector rejects for any reason but a differing digest) -- parity with the oracle, NOT with the reference"""
    circ, ids = grumpkin_circuit()
    rows = grumpkin_rows(24, first_instance=8)
    rows[10][4] = (rows[10][4] + 1) % P
    for i in range(32):
        rows[11][6 + i] = 0
        rows[12][6 + 32 + i] = 0
    ores, stats = run_both(oracle, circ, ids, rows)
    run_both(oracle, circ, ids, rows, force_slow=True)
    out = circ.current_witness_index
    assert all(ores[j].status == 0 for j in (10, 11, 12))


def test_schnorr_short_signature_panics(oracle):
    circ = Circuit(70, [BB("SchnorrVerify", {"public_key_x": FI(1, 254), "public_key_y": FI(2, 254), "signature": [FI(w, 8) for w in range(3, 66)],
                                            "message": [FI(66, 8)], "output": 67})])
    rows = [[1, GY] + [0] * 63 + [7]] * 2
    ores, _ = both_paths(oracle, circ, list(range(1, 67)), rows)
    assert ores[0].err == oracle.E_PANIC


@pytest.mark.parametrize("pattern", ["slice_pairs", "windows"])
def test_pedersen_pair_table_sweep(pattern):
    """The level kernel's lookup tables against the exact in-order kernel, which adds the two 512-entry table points of every slice pair
    separately (and is pinned to the oracle and the reference vectors by the tests above), bit for bit over 2^18 instances.
    slice_pairs: instance k carries the 18-bit pattern k in all of its slice pairs -- every (even slice, odd slice) combination of the 503 MB
    pair table, and as many different entries of each window of the window table. windows: instance k carries a different 24-bit word in each
    of the eleven 24-bit windows of the value (a multiplicative hash of k: high bits of the window index included), up to 253 bits.
    The chained hash output exercises random entries of the other half of the tables."""
    import acvm_amd
    B = 1 << 18
    circ = Circuit(3, [BB("Pedersen", {"inputs": [FI(1, 254)], "domain_separator": 0, "outputs": [2, 3]})])
    k = np.arange(B, dtype=np.uint64)
    vals = np.zeros((B, 1, 32), dtype=np.uint8)
    if pattern == "slice_pairs":
        # value = sum_i k << 18 i for i < 13 (234 bits): byte-wise assembly through Python ints in blocks of 4096 instances
        rep = sum(1 << (18 * i) for i in range(13))
        value = lambda x: x * rep
    else:
        value = lambda x: sum((((x * (2 * j + 1) * 0x9E3779B1) >> 7) & 0xFFFFFF) << (24 * j) for j in range(11)) & ((1 << 253) - 1)
    for s in range(0, B, 4096):
        vals[s:s + 4096, 0] = np.frombuffer(b"".join(be32(value(int(x))) for x in k[s:s + 4096]), dtype=np.uint8).reshape(-1, 32)
    data = circ.to_bytes()
    out = []
    for force_slow in (False, True):
        batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, [1])
        batch.set_force_slow_path(force_slow)
        batch.set_initial_witness(vals.tobytes())
        assert batch.solve() == 0
        x, ax = batch.witness(2)
        y, ay = batch.witness(3)
        assert ax.all() and ay.all()
        out.append((x.copy(), y.copy()))
        batch.free()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
