#!/usr/bin/env python3
"""tools/plan_fingerprint.py [--big] [--config5] [out.json] -- fingerprints of the static plan (acvm_debug_plan_fingerprint) of every
circuit of tests/circuit_corpus.py under every planner mode x {plain, fold, reuse}, as JSON. Run before and after a change of the
planner that must not move a word, and diff the two files. Host only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import acvm_amd  # noqa: E402
import circuit_corpus as cc  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = {}
    items = cc.corpus(big="--big" in sys.argv)
    if "--config5" in sys.argv:
        circ, ids = cc.config5_circuit()
        items.append(("config5_1m", circ.to_bytes(), ids))
    for name, data, ids in items:
        gc = acvm_amd.Circuit(data)
        if ids is None:
            ids = gc.witness_set("circuit_arguments")
        keep = gc.witness_set("return_values")
        modes = cc.PLANNER_MODES if len(data) < 400000 else [{}]
        for mi, mode in enumerate(modes):
            with acvm_amd.tuning(**mode):
                for variant, kw in (("plain", {}), ("fold", {"fold_digest": True}), ("reuse", {"reuse_slots": True, "keep": keep}), ("solver", {"host_solver": True})):
                    try:
                        fp = gc.plan_fingerprint(ids, **kw)
                    except acvm_amd.AcvmError as e:
                        fp = "refused: " + str(e)[:60]
                    out["%s|%d|%s" % (name, mi, variant)] = fp
    text = json.dumps(out, indent=0, sort_keys=True)
    if args:
        open(args[0], "w").write(text)
    print(len(out), "plans fingerprinted")


if __name__ == "__main__":
    main()
