import sys, time
sys.path.insert(0, '.')
import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, Expression as E, FunctionInput as FI, MemoryInit, MemoryOp, QuotientDirective, ToLeRadix, Brillig
B = 1 << 16
N = 32
def run(name, ops, nw):
    circ = Circuit(nw, ops)
    ids = list(range(1, 17))
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
    batch.set_initial_witness(synth.witness_batch(B, seed=3, edge_cases=False))
    batch.set_profiling(True)
    batch.solve(); batch.reset(); batch.solve()
    st = batch.stats()
    print(f"{name:14s} levels {st['n_levels']:3d} class_ms {['%.3f' % x for x in st['class_kernel_ms']]} per-record us {1e3 * sum(st['class_kernel_ms']) / N:8.1f} slow {st['n_slow_instances']}")
    batch.free()
w = 17
run("RANGE", [BB("RANGE", {"input": FI(1 + i % 16, 254)}) for i in range(N)], 16)
run("AND", [BB("AND", {"lhs": FI(1 + i % 16, 64), "rhs": FI(1 + (i + 1) % 16, 64), "output": w + i}) for i in range(N)], w + N)
run("Quotient", [QuotientDirective(E.from_witness(1 + i % 16), E.from_witness(1 + (i + 3) % 16), w + 2 * i, w + 2 * i + 1) for i in range(N)], w + 2 * N)
run("ToLeRadix256", [ToLeRadix(E.from_witness(1 + i % 16), list(range(w + 32 * i, w + 32 * i + 32)), 256) for i in range(N)], w + 32 * N)
run("ToLeRadix2", [ToLeRadix(E.from_witness(1 + i % 16), list(range(w + 254 * i, w + 254 * i + 254)), 2) for i in range(N)], w + 254 * N)
ops = [MemoryInit(b, list(range(1, 17))) for b in range(N)]
run("MemInit", ops, 16)
ops = [MemoryInit(b, list(range(1, 17))) for b in range(N)] + [MemoryOp(b, E.constant(0), E.constant(b % 16), E.from_witness(w + b)) for b in range(N)]
run("MemInit+Read", ops, w + N)
ops = [Brillig(inputs=[E.from_witness(1 + i % 16), E.from_witness(1 + (i + 1) % 16)], outputs=[w + i], bytecode=[("BinaryIntOp", 0, "Mul", 64, 0, 1), ("Stop",)]) for i in range(N)]
run("Brillig mul64", ops, w + N)
ops = [BB("Pedersen", {"inputs": [FI(1 + i % 16, 254), FI(1 + (i + 1) % 16, 254)], "domain_separator": 0, "outputs": [w + 2 * i, w + 2 * i + 1]}) for i in range(N)]
run("Pedersen x32", ops, w + 2 * N)
