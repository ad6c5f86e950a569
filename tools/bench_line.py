"""one short line per bench.py JSON line on stdin"""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        if line:
            print(line)
        continue
    d = json.loads(line)
    r = d["roofline"]
    a = d.get("alu_roofline") or {}
    e = d.get("end_to_end") or {}
    c = d.get("cpu_baseline") or {}
    print(round(d["value"]), "w/s", round(d["ms_per_step"], 2), "ms/step | tile", d["config"]["tile_instances"], "frac", round(r["frac"], 3), r["kernel"],
          round(r["kernel_ms_per_tile"], 2), "ms/tile", {k: round(v, 2) for k, v in r["other_kernels_ms_per_tile"].items()},
          "| traffic", None if r.get("traffic") is None else round(r["traffic"] / 1e9, 3), "GB | alu", None if not a.get("frac") else round(a["frac"], 3),
          "| e2e", None if not e else round(e["value"]), "| cpu", None if not c else round(c["value"], 1),
          "| dod", (d.get("digest_of_digests") or {}).get("value", "")[:12], "| legs", d.get("legs"))
