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


def wild_circuit(n_ops, seed):
    """Black-box opcodes with operands and widths drawn without regard for what the values are: hash inputs of any width fed with field-sized
    values, byte RANGE checks on them (passing and failing), outputs that are already assigned (insert_value compares), several hashes and
    checks per level. Most instances fail somewhere; what counts is WHERE and with what, bit for bit."""
    import random
    from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, Expression as E, FunctionInput as FI, P, ToLeRadix, QuotientDirective
    r = random.Random(seed)
    n_in = 12
    ids = list(range(1, n_in + 1))
    nw = n_in
    ops = []
    bytes_w = []
    w32 = []

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
                outs[r.randrange(32)] = pick()  # an output that is already assigned
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


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 700
    wild = len(sys.argv) > 3 and sys.argv[3] == "wild"
    t0 = time.time()
    bad = 0
    for k in range(n_seeds):
        seed = 0xF0220000 + int(os.environ.get("ACVM_FUZZ_BASE", "0")) + k  # (ACVM_FUZZ_BASE: another stretch of seeds)
        heavy = k % 5 != 4
        blocks, cells = [(16, 64), (4, 16), (1, 256), (2, 2)][k % 4]
        n = n_ops + 37 * (k % 7)
        B = [70, 1, 64, 129, 200][k % 5]
        if wild:
            circ, ids = wild_circuit(n, seed)
            values = bytearray(synth.witness_batch(B, n_in=len(ids), seed=seed, edge_cases=(k % 2 == 0)))
            for j in range(B):  # most inputs byte-sized, so that checks pass often enough for later opcodes to run
                for i in range(len(ids)):
                    if (j * 31 + i * 7 + k) % 5:
                        o = (j * len(ids) + i) * 32
                        values[o:o + 31] = bytes(31)
            values = bytes(values)
        else:
            circ, ids = synth.mixed_circuit(n, seed=seed, heavy=heavy, blocks=blocks, cells=cells)
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
            except (acvm_amd.AcvmError, ValueError) as e:  # slot reuse refuses circuits it cannot cover (foreign calls, truncated plans)
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
