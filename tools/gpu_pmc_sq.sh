#!/bin/bash
# tools/gpu_pmc_sq.sh TAG [bench args...] -- instruction-mix counters of the bench kernels (own PMC passes, kernel-trace only)
set -u
TAG=${1:-rXX}
shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- $BENCH > "$OUT/$name.log" 2>&1
done
find "$OUT" -name '*.db' -delete
python - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in tot:
    if "arith" in k or "level" in k:
        print(k, {c: (round(v / cnt[k][c]), cnt[k][c]) for c, v in tot[k].items()})
PY
