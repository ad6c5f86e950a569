"""GPU parity of Directive::PermutationSort (acvm/src/pwg/directives/{mod.rs:88-119, sorting.rs}) against the CPU oracle,
level path and exact path, plus the network-execution property on the device output."""
import random

import pytest

from acvm_amd.acir import P, Circuit, Expression as E, PermutationSort
from test_gpu_opcodes import both_paths
from test_oracle_sorting import execute_network, switch_nb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,tuple_,sort_by", [(1, 1, [0]), (2, 1, [0]), (3, 1, [0]), (4, 2, [0]), (5, 2, [1, 0]), (8, 1, [0]), (13, 3, [2]), (17, 2, [0, 1]),
                                              (32, 1, [0]), (256, 1, [0])])
def test_permutation_sort(oracle, n, tuple_, sort_by):
    r = random.Random(n * 7 + tuple_)
    ids = list(range(1, n * tuple_ + 1))
    ins = [[E.from_witness(1 + i * tuple_ + k) for k in range(tuple_)] for i in range(n)]
    nb = switch_nb(n)
    bits = list(range(n * tuple_ + 1, n * tuple_ + 1 + nb))
    circ = Circuit(n * tuple_ + nb + 1, [PermutationSort(ins, tuple_, bits, sort_by)])
    rows = []
    for j in range(66):
        small = j % 3 == 0  # many ties -> stability matters
        rows.append([r.randrange(4) if small else r.randrange(P) for _ in ids])
    rows[1] = [0] * len(ids)
    rows[2] = list(range(len(ids), 0, -1))
    ores, _ = both_paths(oracle, circ, ids, rows)
    assert all(ores[j].status == 0 for j in range(len(rows)))
    if n >= 2:
        import acvm_amd
        from acvm_amd.synth import values_from_rows
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), len(rows), ids)
        batch.set_initial_witness(values_from_rows(rows))
        batch.solve()
        asg, vals = batch.witness_map()
        for j in (0, 2, 7):
            config = [bool(vals[j, w, 31]) for w in bits]
            keyf = lambda i: tuple(rows[j][i * tuple_ + c] % P for c in sort_by)  # noqa: E731
            assert execute_network(config, list(range(n))) == sorted(range(n), key=keyf)


def test_sort_inside_a_circuit(oracle):
    """Arithmetic gates feed the sort and consume its control bits."""
    n = 6
    ops = [E([(1, 1 + i, 1 + (i + 1) % n)], [(P - 1, 10 + i)], 0) for i in range(n)]          # w10+i = w_i * w_{i+1}
    nb = switch_nb(n)
    bits = list(range(30, 30 + nb))
    ops.append(PermutationSort([[E.from_witness(10 + i)] for i in range(n)], 1, bits, [0]))
    ops.append(E([(1, bits[0], bits[1])], [(1, bits[2]), (P - 1, 60)], 0))                    # w60 = b0 * b1 + b2
    circ = Circuit(60, ops)
    r = random.Random(3)
    rows = [[r.randrange(P) for _ in range(n)] for _ in range(40)]
    rows[0] = [0] * n
    both_paths(oracle, circ, list(range(1, n + 1)), rows)


def test_sort_column_beyond_the_tuple_panics(oracle):
    """a[*i as usize] (directives/mod.rs:102-105) with sort_by = [tuple + 1]: the reference panics on the first comparison; n == 1
    never compares and solves"""
    w = E.from_witness
    circ = Circuit(8, [PermutationSort([[w(1)], [w(2)], [w(3)]], 1, [4, 5, 6], [2])])
    ores, _ = both_paths(oracle, circ, [1, 2, 3], [[3, 2, 1], [1, 1, 1], [0, 5, P - 1]])  # (both_paths compares the message texts too)
    assert all(r.err == 8 and r.message == b"index out of bounds: the len is 2 but the index is 2" for r in ores)
    one = Circuit(3, [PermutationSort([[w(1)]], 1, [], [5])])
    ores, _ = both_paths(oracle, one, [1], [[3], [0]])
    assert all(r.status == 0 for r in ores)
