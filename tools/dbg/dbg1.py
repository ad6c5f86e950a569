import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import acvm_amd
from oracle import binding as ob
from acvm_amd.acir import P, Brillig, Circuit, Expression as E
from acvm_amd.synth import values_from_rows
W = E.from_witness
a = Brillig(inputs=[W(1), W(2)], outputs=[2], bytecode=[("BinaryFieldOp", 0, "Add", 0, 1), ("Stop",)])
b = Brillig(inputs=[W(1)], outputs=[4, 4], bytecode=[("Mov", 1, 0), ("Stop",)])
c = Brillig(inputs=[E([(3, 1, 2)], [(2, 1)], 7)], outputs=[5], bytecode=[("Const", 1, 2), ("BinaryFieldOp", 0, "Mul", 0, 1)])
d = Brillig(inputs=[W(9)], outputs=[6], bytecode=[("Stop",)])
rows = [[0, 5], [1, 5], [0, 0], [P - 1, 1]]
for name, ops in (("abcd", [a, b, c, d]), ("a", [a]), ("ad", [a, d])):
    circ = Circuit(9, ops)
    data = circ.to_bytes()
    ores, _, _ = ob.solve_batch(ob.Circuit(data), [1, 2], values_from_rows(rows), 4)
    print(name, "oracle", [r.as_tuple() for r in ores])
    for inline in (1, 0):
        for slow in (False, True):
            with acvm_amd.tuning(brillig_inline=inline):
                bt = acvm_amd.Batch(acvm_amd.Circuit(data), 4, [1, 2])
            bt.set_force_slow_path(slow)
            bt.set_initial_witness(values_from_rows(rows))
            bt.solve()
            print(name, "inline", inline, "slow", slow, [r.as_tuple() for r in bt.results()], {k: bt.stats()[k] for k in ("n_brillig_inlined", "truncated_at", "n_levels", "n_kernel_launches")})
