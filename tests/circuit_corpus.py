"""Circuit shapes of the GPU parity suite as one list, for the host-only tests that look at the PLAN and the level SCHEDULE of each
(tests/test_schedule_hazards.py, tests/test_plan_host.py): every generator of acvm_amd.synth at the sizes the GPU tests use, the wild
black-box circuits of tools/t_fuzz.py, the reference's own byte-exact circuits (tests/golden/reference_vectors.json) and small
hand-written shapes that exercise one scheduling rule each (hash -> gate, gate -> hash -> gate, memory chains, inversions behind
heavy outputs, compared outputs, Brillig with a foreign call). Nothing here needs a GPU or the oracle."""
import json
import os
import random

from acvm_amd import synth
from acvm_amd.acir import P, BlackBoxFuncCall as BB, Brillig, Circuit, Expression as E, FunctionInput as FI, MemoryInit, MemoryOp, \
    PermutationSort, QuotientDirective, ToLeRadix

HERE = os.path.dirname(os.path.abspath(__file__))

# every planner / scheduler mode of tests/test_gpu_planner_modes.py that changes the PLAN or the SCHEDULE (kernel-internal modes such as
# pedersen_waves leave both alone and are not repeated here)
PLANNER_MODES = [
    {},
    {"scale": 0}, {"pairs": 0}, {"chains": 0}, {"max_tails": 1}, {"max_tails": 8}, {"inv_epoch": 1}, {"inv_epoch": 9}, {"inv_latency": 0}, {"inv_latency": 3},
    {"heavy_epoch": 4, "heavy_latency": 4}, {"pedersen_latency": 6}, {"pedersen_epoch": 1, "pedersen_latency": 0}, {"pedersen_epoch": 5}, {"digest_epoch": 1},
    {"digest_epoch": 32}, {"range_fuse": 0}, {"range_merge": 0}, {"range_fuse": 0, "range_merge": 0}, {"brillig_inline": 0}, {"sl_lane": 1}, {"hash_chain": 0}, {"light_fuse": 0},
    {"relax": 0}, {"pedersen_bundle": 2, "pedersen_epoch": 8}, {"inv_chunk": 1000, "inv_epoch": 9}, {"byte_plane": 0}, {"overlap": 0}, {"heavy_streams": 0},
    {"scale": 0, "pairs": 0, "range_fuse": 0, "range_merge": 0, "brillig_inline": 0, "light_fuse": 0, "overlap": 0},
]


def wild_circuit(n_ops, seed):
    """tools/t_fuzz.py wild_circuit: black-box opcodes with operands and widths drawn without regard for the values (kept in step with the tool)."""
    r = random.Random(seed)
    n_in = 12
    ids = list(range(1, n_in + 1))
    nw = n_in
    ops, bytes_w, w32 = [], [], []

    def fresh(k=1):
        nonlocal nw
        out = list(range(nw + 1, nw + 1 + k))
        nw += k
        return out

    def pick():
        return r.randrange(1, nw + 1)

    for _ in range(n_ops):
        k = r.randrange(100)
        if k < 25:
            if bytes_w and r.random() < 0.9:
                ops.append(BB("RANGE", {"input": FI(r.choice(bytes_w), r.choice([8, 8, 8, 8, 9, 16, 64, 254, 7]))}))
            else:
                ops.append(BB("RANGE", {"input": FI(pick(), r.choice([0, 1, 3, 8, 254, 254, 254]))}))
        elif k < 40:
            bits = r.choice([1, 8, 8, 13, 32, 64, 254])
            out = fresh()[0] if r.random() < 0.97 else pick()
            ops.append(BB(r.choice(["AND", "XOR"]), {"lhs": FI(pick(), bits), "rhs": FI(pick(), bits), "output": out}))
            if bits == 8:
                bytes_w.append(out)
            if bits == 32:
                w32.append(out)
        elif k < 65:
            name = r.choice(["SHA256", "Keccak256", "Blake2s"])
            n = r.choice([1, 2, 3, 4, 5, 17, 33, 64])
            src = bytes_w if len(bytes_w) >= 4 and r.random() < 0.7 else None
            ins = [FI(r.choice(src) if src else pick(), r.choice([8, 8, 8, 1, 4, 7]) if r.random() < 0.85 else r.choice([9, 32, 254])) for _ in range(n)]
            outs = fresh(32)
            if r.random() < 0.03:
                outs[r.randrange(32)] = pick()
            ops.append(BB(name, {"inputs": ins, "outputs": outs}))
            bytes_w += [w for w in outs if w > n_in]
        elif k < 72:
            d = fresh(4)
            ops.append(ToLeRadix(E.from_witness(r.choice(w32) if w32 and r.random() < 0.95 else pick()), d, 256))
            bytes_w += d
        elif k < 80:
            q, rem = fresh(2)
            ops.append(QuotientDirective(E.from_witness(pick()), E.from_witness(pick()), q, rem))
        elif k < 86:
            ops.append(BB("HashToField128Security", {"inputs": [FI(pick(), r.choice([8, 16, 254])) for _ in range(r.randrange(1, 6))], "output": fresh()[0]}))
        else:
            a, b = pick(), pick()
            out, = fresh()
            ops.append(E([(r.randrange(1, P), a, b)], [(P - 1, out), (r.randrange(P), pick())], r.randrange(P)))
    return Circuit(nw, ops), ids


def _hand_written():
    """one scheduling rule each"""
    out = []
    # a hash of two input bytes, a gate that reads a digest byte, a second hash of digest bytes (a chain), a gate behind it
    ops = [BB("SHA256", {"inputs": [FI(1, 8), FI(2, 8)], "outputs": list(range(3, 35))}),
           E([(1, 3, 4)], [(P - 1, 35)], 0),
           BB("Keccak256", {"inputs": [FI(w, 8) for w in range(3, 35)], "outputs": list(range(36, 68))}),
           E([(1, 36, 35)], [(P - 1, 68), (5, 1)], 7),
           BB("RANGE", {"input": FI(36, 8)})]
    out.append(("hash_gate_hash_gate", Circuit(68, ops), [1, 2]))
    # an unknown-in-mul gate whose multiplicand is a Pedersen output, then a gate on the result, then a Pedersen of that
    ops = [BB("Pedersen", {"inputs": [FI(1, 254), FI(2, 254)], "domain_separator": 0, "outputs": [3, 4]}),
           E([(7, 3, 5)], [(1, 2)], 3),          # 7 * w3 * w5 + w2 + 3 = 0: w5 = -(w2 + 3) / (7 w3)
           E([(1, 5, 5)], [(P - 1, 6)], 0),
           BB("Pedersen", {"inputs": [FI(6, 254), FI(4, 254)], "domain_separator": 0, "outputs": [7, 8]}),
           BB("FixedBaseScalarMul", {"low": FI(1, 128), "high": FI(2, 128), "outputs": [9, 10]}),
           E([(1, 9, 7)], [(P - 1, 11)], 0)]
    out.append(("inverse_behind_pedersen", Circuit(11, ops), [1, 2]))
    # memory: init, writes and reads interleaved on two blocks, the read targets feeding gates
    ops = [MemoryInit(0, [1, 2, 3, 4]), MemoryInit(1, [4, 3, 2, 1]),
           BB("AND", {"lhs": FI(1, 8), "rhs": FI(5, 8), "output": 6}),
           MemoryOp(0, E.constant(0), E.from_witness(6), E.from_witness(7)),
           MemoryOp(0, E.constant(1), E.from_witness(6), E.from_witness(2)),
           MemoryOp(0, E.constant(0), E.from_witness(6), E.from_witness(8)),
           MemoryOp(1, E.constant(0), E.from_witness(6), E.from_witness(9)),
           E([(1, 7, 8)], [(P - 1, 10), (1, 9)], 0),
           MemoryOp(1, E.constant(1), E.from_witness(6), E.from_witness(10)),
           MemoryOp(1, E.constant(0), E.from_witness(6), E.from_witness(11))]
    out.append(("memory_chains", Circuit(11, ops), [1, 2, 3, 4, 5]))
    # outputs that are already assigned (insert_value compares): a logic op and a hash whose outputs an earlier gate defined
    ops = [E([], [(1, 1), (P - 1, 3)], 0), BB("XOR", {"lhs": FI(1, 8), "rhs": FI(2, 8), "output": 3}),
           BB("Blake2s", {"inputs": [FI(1, 8)], "outputs": [3] + list(range(4, 35))}), E([(1, 4, 5)], [(P - 1, 35)], 0)]
    out.append(("compared_outputs", Circuit(35, ops), [1, 2]))
    # Brillig: a straight-line program, a loop-free one with a foreign call, a gate on each output
    ops = [Brillig(inputs=[E.from_witness(1), E.from_witness(2)], outputs=[3], bytecode=[("BinaryIntOp", 0, "Add", 32, 0, 1), ("Stop",)]),
           Brillig(inputs=[E.from_witness(3)], outputs=[4], bytecode=[("ForeignCall", "invert", [("Register", 0)], [("Register", 0)]), ("Stop",)]),
           E([(1, 3, 4)], [(P - 1, 5)], 0)]
    out.append(("brillig_foreign_call", Circuit(5, ops), [1, 2]))
    # ToLeRadix + Quotient + PermutationSort + ECDSA-free heavy directives
    ops = [BB("AND", {"lhs": FI(1, 32), "rhs": FI(2, 32), "output": 3}), ToLeRadix(E.from_witness(3), [4, 5, 6, 7], 256),
           QuotientDirective(E.from_witness(1), E.from_witness(4), 8, 9),
           PermutationSort([[E.from_witness(4)], [E.from_witness(5)], [E.from_witness(6)], [E.from_witness(7)]], 1, list(range(10, 15)), [0]),
           E([(1, 8, 9)], [(P - 1, 15), (1, 10)], 0)]
    out.append(("directives", Circuit(15, ops), [1, 2]))
    return out


def _reference_circuits():
    """the byte-exact circuits of the reference's own serialisation tests (tests/golden/reference_vectors.json)"""
    path = os.path.join(HERE, "golden", "reference_vectors.json")
    out = []
    try:
        doc = json.load(open(path))
    except OSError:
        return out
    for name, entry in sorted(doc.get("serialization", {}).items()):
        out.append(("ref_" + name, bytes(entry), None))
    return out


def corpus(big=False):
    """[(name, circuit bytes, initial ids)]; ids None = derive from the circuit's parameters. `big` adds the sizes that take seconds to plan."""
    items = []

    def add(name, circ, ids):
        items.append((name, circ.to_bytes() if hasattr(circ, "to_bytes") else circ, ids))

    add("arith_2000", *synth.arithmetic_circuit(2000, seed=0xAC1D0002))
    add("arith_chain_200", *synth.arithmetic_circuit(200, seed=7, chain=True))
    add("arith_hot_inputs", *synth.arithmetic_circuit(1500, seed=0xAC1D0012, hot_inputs=True))
    add("arith_inversions_40pct", *synth.arithmetic_circuit(800, seed=0x1234, mix=(20, 20, 20, 40)))
    add("wide_gates", *synth.wide_gate_circuit(300, seed=0xAC1D0A11, max_terms=12))
    add("config3_hash", *synth.hash_circuit(64, True))
    add("hash_no_range", *synth.hash_circuit(136, False))
    add("config4_grumpkin", *synth.grumpkin_circuit(10, 2))
    add("ecdsa", *synth.ecdsa_circuit())
    add("north_star_1000_4", *synth.arith_pedersen_circuit(1000, 4, seed=0xAC1D0006))
    for seed in (0xAC1D0001, 0xAC1D0005, 0x300D0000, 0x300D0007):
        add("mixed_900_%x" % seed, *synth.mixed_circuit(900, seed=seed, heavy=True, blocks=4, cells=16))
    add("mixed_noheavy", *synth.mixed_circuit(1200, seed=0xC0FFEE, heavy=False))
    for seed in range(4):
        add("wild_%d" % seed, *wild_circuit(120, 0xF022 + seed))
    for name, circ, ids in _hand_written():
        add(name, circ, ids)
    items += _reference_circuits()
    if big:
        add("config2_10k", *synth.arithmetic_circuit(10000, seed=0xAC1D0002))
        add("north_star_10k_8", *synth.arith_pedersen_circuit(10000, 8, seed=0xAC1D0006))
        add("mixed_20k", *synth.mixed_circuit(20000, seed=0xAC1D0005))
    return items


def config5_circuit(n_gates=1_000_000, seed=0xAC1D0005):
    """BASELINE config 5 at circuit size (the 10^6-opcode circuit of tests/test_gpu_config5.py)"""
    circ, ids = synth.mixed_circuit(n_gates, seed=seed)
    return circ, ids
