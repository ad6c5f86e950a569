"""Pins oracle/sorting.c (Directive::PermutationSort): the five known answers of acvm/src/pwg/directives/sorting.rs:309-383
and its property that executing the network with the returned control bits maps inputs to outputs (sorting.rs:256-307,384-394);
the directive itself against the expected control witnesses of a small sort."""
import ctypes as C
import math
import random

from acvm_amd.acir import Circuit, Expression as E, PermutationSort


def route(oracle, a, b):
    n = len(a)
    bits = C.create_string_buffer(32 * n + 8)
    nb = oracle.lib().oracle_sorting_route((C.c_uint32 * n)(*a), (C.c_uint32 * n)(*b), n, bits)
    return [bool(x) for x in bits.raw[:nb]]


def switch_nb(n):
    return sum(math.ceil(math.log2(i + 1)) for i in range(n))


def execute_network(config, inputs):
    """sorting.rs:256-297"""
    n = len(inputs)
    if n == 1:
        return inputs
    in1, in2 = [], []
    for i in range(n // 2):
        if config[i]:
            in1.append(inputs[2 * i + 1]); in2.append(inputs[2 * i])
        else:
            in1.append(inputs[2 * i]); in2.append(inputs[2 * i + 1])
    if n % 2 == 1:
        in2.append(inputs[-1])
    n2 = n // 2 + (n - 1) // 2
    n3 = n2 + switch_nb(n // 2)
    out1 = execute_network(config[n2:n3], in1)
    out2 = execute_network(config[n3:], in2)
    result = []
    for i in range((n - 1) // 2):
        if config[n // 2 + i]:
            result += [out2[i], out1[i]]
        else:
            result += [out1[i], out2[i]]
    if n % 2 == 0:
        result += [out1[-1], out2[-1]]
    else:
        result.append(out2[-1])
    return result


def test_route_known_answers(oracle):
    assert route(oracle, [1, 2, 3], [1, 2, 3]) == [False, False, False]
    assert route(oracle, [1, 2, 3], [1, 3, 2]) == [False, False, True]
    assert route(oracle, [1, 2, 3], [3, 2, 1]) == [True, True, True]
    assert route(oracle, [0, 1, 2, 3], [2, 3, 0, 1]) == [False, True, True, True, True]
    assert route(oracle, [0, 1, 2, 3, 4], [0, 3, 4, 2, 1]) == [False, False, False, True, False, True, False, True]


def test_route_network_property(oracle):
    r = random.Random(5)
    for n in list(range(2, 50)) + [64, 100, 127]:
        a = list(range(n))
        b = a[:]
        r.shuffle(b)
        c = route(oracle, a, b)
        assert len(c) == switch_nb(n)
        assert execute_network(c, a) == b, n


def test_directive_sorts_and_routes(oracle):
    # 5 elements of 2-tuples (key, payload), sorted by column 0; the control bits must route the identity to the sorted order
    n = 5
    ins = [[E.from_witness(1 + 2 * i), E.from_witness(2 + 2 * i)] for i in range(n)]
    nb = switch_nb(n)
    bits = list(range(20, 20 + nb))
    circ = Circuit(40, [PermutationSort(ins, 2, bits, [0])])
    keys = [30, 10, 20, 10, 5]
    iw = {}
    for i in range(n):
        iw[1 + 2 * i] = keys[i]
        iw[2 + 2 * i] = 100 + i
    a = oracle.ACVM(oracle.Circuit(circ.to_bincode()), iw)
    assert a.solve() == oracle.ST_SOLVED
    wm = a.witness_map()
    config = [bool(wm[w]) for w in bits]
    order = sorted(range(n), key=lambda i: keys[i])  # stable
    assert execute_network(config, list(range(n))) == order
