#!/usr/bin/env python3
"""tools/trace_timeline.py TRACE_DIR [first_kernel_substr] -- who holds the device when, from a rocprofv3 --kernel-trace CSV of ONE solve-heavy
run: per kernel class the busy time (union of its launches), the time it runs alone, the time NOTHING runs, and how the wall clock of the
window splits by the set of classes in flight. The window is the last but one `solve` in the trace: the level-schedule kernels between two
imports (import_witness_kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict

CLASSES = [("arith", "arith_l"), ("inv", "inverse_batch_kernel"), ("light", "LightOp"), ("lightsl", "LightSlOp"), ("hash", "hash_coop_level_kernel"),
           ("hash", "HashOp"), ("pedersen", "pedersen_quad"), ("grumpkin", "GrumpkinOp"), ("brillig", "BrilligOp"), ("ecdsa", "EcdsaOp"), ("digest", "digest_")]


def cls_of(name):
    for c, sub in CLASSES:
        if sub in name:
            return c
    return None


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a solve = the level-schedule kernels between two imports (since round 6 the import resets the event words and the kernels that flag
    # count: there is no event_reset_kernel / event_count_kernel to delimit it)
    imports = [s for s, e, n in rows if "import_witness_kernel" in n]
    if not imports:
        print("no solve found")
        return
    # the solve to analyse: the last but one (the last may carry profiling events), or the last
    k = -2 if len(imports) >= 2 else -1
    lo = imports[k]
    hi = imports[k + 1] if k + 1 < 0 else float("inf")
    win = [(s, e, cls_of(n)) for s, e, n in rows if s > lo and s < hi and cls_of(n) and "digest_final" not in n and "digest_chunk" not in n]
    if not win:
        print("no solve found")
        return
    t0 = min(s for s, e, c in win)
    t1 = max(e for s, e, c in win)
    print(f"window {(t1 - t0) / 1e6:.2f} ms, {len(win)} launches")
    ev = []
    for s, e, c in win:
        ev.append((s, 1, c))
        ev.append((e, -1, c))
    ev.sort()
    active = defaultdict(int)
    last = t0
    by_set = defaultdict(int)
    for t, dlt, c in ev:
        key = tuple(sorted(k for k, v in active.items() if v > 0))
        by_set[key] += t - last
        last = t
        active[c] += dlt
    by_set[()] += t1 - last
    busy = defaultdict(int)
    alone = defaultdict(int)
    for key, dt in by_set.items():
        for c in key:
            busy[c] += dt
        if len(key) == 1:
            alone[key[0]] += dt
    n_l = defaultdict(int)
    own = defaultdict(int)
    for s, e, c in win:
        n_l[c] += 1
        own[c] += e - s
    print(f"{'class':10s} {'launches':>8s} {'sum_own_ms':>11s} {'busy_ms':>9s} {'alone_ms':>9s}")
    for c in sorted(busy, key=lambda c: -busy[c]):
        print(f"{c:10s} {n_l[c]:8d} {own[c] / 1e6:11.2f} {busy[c] / 1e6:9.2f} {alone[c] / 1e6:9.2f}")
    print(f"idle (no kernel in flight): {by_set[()] / 1e6:.2f} ms")
    print("wall clock by set of classes in flight (top 12):")
    for key, dt in sorted(by_set.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {dt / 1e6:8.2f} ms  {'+'.join(key) or '(idle)'}")


if __name__ == "__main__":
    main()
