#!/bin/bash
# tools/gpu_trace_arith.sh TAG -- kernel-trace timeline of one tile of the metric's workload (tools/trace_timeline.py): what is in flight when
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/tl_arith_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ACVM_BENCH_NO_PMC=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/bench.py --inner --workload arith --total-log2 17 --tile-log2 17 --steps 3 --warmup 2 > "$OUT/run.log" 2>&1
python $ROOT/tools/trace_timeline.py "$OUT/trace" > "$OUT/timeline.txt"
cat "$OUT/timeline.txt"
python - "$OUT/trace" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-32:]))
rows.sort()
resets = [i for i, r in enumerate(rows) if "import_witness" in r[2]]  # a tile = from one import to the next (the import resets the event words since round 6)
i0 = resets[-2]; i1 = resets[-1]
seq = [r for r in rows[i0:i1] if "arith" in r[2]]
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(seq, seq[1:])]
print("gate launches", len(seq), "gaps between consecutive gate launches (us):", [round(g, 1) for g in gaps])
print("sum of gaps", round(sum(gaps) / 1e3, 3), "ms; gate time", round(sum(e - s for s, e, n in seq) / 1e6, 3), "ms; tile", round((rows[i1][0] - rows[i0][0]) / 1e6, 3), "ms")
print("before the first gate launch:", round((seq[0][0] - rows[i0][0]) / 1e3, 1), "us; after the last:", round((rows[i1][0] - seq[-1][1]) / 1e3, 1), "us")
PY
find "$OUT" -name '*.db' -delete
