"""Caller-side helpers that need no GPU: Circuit::get_assert_message (acir/src/circuit/mod.rs:43-51) and the
circuit's witness sets (private / public parameters, return values, public_inputs(), circuit_arguments(); :25-32,109-121)
through the wire-format reader and the C ABI."""
import acvm_amd
from acvm_amd.acir import P, Circuit, Expression as E


def circuit():
    ops = [E([(1, 1, 2)], [(P - 1, 4)], 0), E([], [(1, 4), (P - 1, 3)], 0)]
    return Circuit(5, ops, private_parameters=[2, 1], public_parameters=[3], return_values=[4, 3],
                   assert_messages=[(1, "x * y must equal z"), ((0, 7), "inside brillig"), (1, "shadowed duplicate")])


def test_assert_messages_by_location():
    c = acvm_amd.Circuit(circuit().to_bytes())
    assert c.get_assert_message(1) == "x * y must equal z"          # first match wins, like the reference's linear find
    assert c.get_assert_message(0) is None
    assert c.get_assert_message(0, 7) == "inside brillig"
    assert c.get_assert_message(0, 6) is None and c.get_assert_message(1, 7) is None


def test_witness_sets_are_sorted_sets():
    c = acvm_amd.Circuit(circuit().to_bytes())
    assert c.witness_set("private_parameters") == [1, 2]
    assert c.witness_set("public_parameters") == [3]
    assert c.witness_set("return_values") == [3, 4]
    assert c.witness_set("public_inputs") == [3, 4]
    assert c.witness_set("circuit_arguments") == [1, 2, 3]


def test_witness_map_wire_format_against_the_reference_vector(golden):
    """acvm_js/test/shared/witness_compression.ts: the compressed map decodes to the expected map; re-encoding gives the same
    bincode bytes under the gzip layer (the compressed bytes themselves depend on the deflate implementation)."""
    import gzip
    fx = golden["acvm_js"]["witness_compression"]
    blob = bytes(fx["expectedCompressedWitnessMap"])
    want = {int(k): int(v, 16) for k, v in fx["expectedWitnessMap"].items()}
    assert acvm_amd.decompress_witness(blob) == want
    mine = acvm_amd.compress_witness(want)
    assert mine[:2] == bytes([0x1f, 0x8b]) and gzip.decompress(mine) == gzip.decompress(blob)
    assert acvm_amd.decompress_witness(mine) == want
    assert acvm_amd.decompress_witness(gzip.decompress(blob)) == want          # raw bincode is accepted too
    assert acvm_amd.decompress_witness(acvm_amd.compress_witness({})) == {}
    import pytest
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.decompress_witness(blob[:40])
