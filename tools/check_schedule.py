#!/usr/bin/env python3
"""tools/check_schedule.py CIRCUIT [n_instances=8192] [plain|fold|reuse|solver] -- the host-only hazard checker (acvm_circuit_check_schedule) over a circuit in
the reference's wire format (Circuit::write: gzip + bincode, what nargo compiles to): the initial witnesses are the circuit's arguments (private + public
parameters), the kept ones its return values. Prints the statistics of the walk and the findings, exits 1 if there are any. No GPU is needed.
With --mutate every cross-stream wait is dropped in turn and the outcome counted (what the schedule's waits are there for)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if not args:
        raise SystemExit(__doc__)
    data = open(args[0], "rb").read()
    n = int(args[1]) if len(args) > 1 else 8192
    variant = args[2] if len(args) > 2 else "plain"
    gc = acvm_amd.Circuit(data)
    ids = gc.witness_set("circuit_arguments")
    kw = {"fold": {"fold_digest": True}, "reuse": {"reuse_slots": True, "keep": gc.witness_set("return_values")}, "solver": {"host_solver": True}}.get(variant, {})
    r = gc.check_schedule(ids, n_instances=n, **kw)
    st = gc.plan_stats(ids, **{k: v for k, v in kw.items() if k != "host_solver"})
    print(f"{st['n_opcodes']} opcodes, {st['n_witnesses']} witnesses, {st['n_levels']} levels, {st['n_table_rows']} rows; {variant}, tiles of {n}")
    print(r["report"])
    if "--mutate" in sys.argv:
        needed = redundant = 0
        for k in range(r["n_waits"]):
            m = gc.check_schedule(ids, n_instances=n, drop_wait=k, **kw)
            if m["ok"]:
                redundant += 1
            else:
                needed += 1
        print(f"mutations: {r['n_waits']} waits, {needed} needed (dropping one is a hazard the checker names), {redundant} covered by another wait")
    raise SystemExit(0 if r["ok"] else 1)


if __name__ == "__main__":
    main()
