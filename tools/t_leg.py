"""One leg of bench.py's default run on its own (bench.run_leg): python tools/t_leg.py NAME TOTAL_LOG2 TILE_LOG2 [STEPS WARMUP].  ACVM_TUNING applies."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

name, total, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps, warmup = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (5, 1)
leg = bench.run_leg(name, total_log2=total, tile_log2=tile, steps=steps, warmup=warmup, pmc=False)
r = leg["roofline"]
print(json.dumps({"tuning": os.environ.get("ACVM_TUNING", ""), "value": round(leg["value"]), "ms_per_step": round(leg["ms_per_step"], 2), "solve_only": round(leg["value_solve_only"] or 0),
                  "frac": r.get("frac"), "kernel_ms": r.get("kernel_ms_per_tile") or r.get("ms"), "parity": leg["parity"].get("bit_exact") if isinstance(leg["parity"], dict) else leg["parity"]}))
