import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): 
        if line: print(line)
        continue
    d=json.loads(line); r=d["roofline"]
    print(round(d["value"]), round(d["ms_per_step"],2), "frac", round(r["frac"],3), r["kernel"], round(r["kernel_ms_per_step"],2), {k:round(v,2) for k,v in r["other_kernels_ms_per_step"].items()})
