#!/usr/bin/env python3
"""tools/t_fuzz.py [n_seeds] [n_opcodes] -- seeded sweep of config-5 style circuits (acvm_amd.synth.mixed_circuit) through every planner
mode against the CPU oracle, bit for bit: plain, slot reuse, folded digest, reuse + folded digest, and the exact path for every instance.
Prints one line per seed and a summary; exits 1 on the first divergence. (Builder's tool: the same comparison as tests/test_gpu_opcodes.py
::test_config5_mixed_circuit over many more circuits than the suite has time for.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd  # noqa: E402
from acvm_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 700
    t0 = time.time()
    bad = 0
    for k in range(n_seeds):
        seed = 0xF0220000 + k
        heavy = k % 5 != 4
        blocks, cells = [(16, 64), (4, 16), (1, 256), (2, 2)][k % 4]
        n = n_ops + 37 * (k % 7)
        circ, ids = synth.mixed_circuit(n, seed=seed, heavy=heavy, blocks=blocks, cells=cells)
        B = [70, 1, 64, 129, 200][k % 5]
        values = synth.witness_batch(B, seed=seed, edge_cases=(k % 2 == 0))
        data = circ.to_bytes()
        ores, oasg, ovals = ob.solve_batch(ob.Circuit(data), ids, values, B)
        odig = [ob.witness_map_digest(oasg[j], ovals[j]) for j in range(B)]
        line = [f"seed {k} ops {n} B {B} heavy {int(heavy)} mem {blocks}x{cells} solved {sum(1 for r in ores if r.status == 0)}/{B}"]
        for mode in ("plain", "reuse", "fold", "reuse+fold", "exact"):
            gc = acvm_amd.Circuit(data)
            kw = {}
            if "reuse" in mode:
                kw.update(reuse_slots=True, keep=gc.witness_set("return_values"))
            if "fold" in mode:
                kw.update(fold_digest=True)
            try:
                batch = acvm_amd.Batch(gc, B, ids, **kw)
            except acvm_amd.AcvmError as e:  # slot reuse refuses circuits it cannot cover (foreign calls, truncated plans)
                line.append(f"{mode}: refused ({str(e)[:40]})")
                continue
            if mode == "exact":
                batch.set_force_slow_path(True)
            batch.set_initial_witness(values)
            try:
                batch.solve()
            except acvm_amd.AcvmError as e:  # slot reuse with more flagged instances than its compact exact table is worth (documented refusal)
                if "reuse" not in mode:
                    raise
                line.append(f"{mode}: refused at solve ({str(e)[26:60]})")
                batch.free()
                continue
            gres = batch.results()
            ok = all(gres[j].as_tuple() == ores[j].as_tuple() for j in range(B))
            dig = batch.digest()
            ok = ok and all(bytes(dig[j]) == odig[j] for j in range(B))
            if "reuse" not in mode:
                gasg, gvals = batch.witness_map()
                nw = min(oasg.shape[1], gasg.shape[1])
                ok = ok and np.array_equal(oasg[:, :nw], gasg[:, :nw]) and np.array_equal(ovals[:, :nw], gvals[:, :nw])
            st = batch.stats()
            batch.free()
            line.append(f"{mode}: {'ok' if ok else 'MISMATCH'} (slow {st['n_slow_instances']}, launches {st['n_kernel_launches']})")
            if not ok:
                bad += 1
        print(" | ".join(line), flush=True)
        if bad:
            print("DIVERGENCE at seed", k)
            sys.exit(1)
    print(f"{n_seeds} circuits x 5 modes bit-exact against the oracle in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
