#!/bin/bash
# tools/gpu_pmc_sq.sh TAG [bench args...] -- instruction-mix counters of the bench kernels (own PMC passes, kernel-trace only); TOTAL="--total-log2 17" for another batch
set -u
TAG=${1:-rXX}
shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --inner ${TOTAL:---total-log2 16} $*"
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- $BENCH > "$OUT/$name.log" 2>&1
done
find "$OUT" -name '*.db' -delete
python - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("# per-launch averages of the SQ counters (rocprofv3 --pmc, one pass per group of three, kernel-trace only) of")
print("# `python bench.py --steps 2 --warmup 1 --inner --total-log2 16 <args>`; the last column of each pair is the number of launches seen")
for k in sorted(tot):
    if "arith" in k or "level" in k or "inverse" in k:
        c = tot[k]; n = cnt[k]
        print(k)
        for name in sorted(c):
            print(f"    {name:24s} {c[name] / n[name]:16.0f}   ({n[name]} launches)")
        if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
            w = c["SQ_WAVES"] / n["SQ_WAVES"]
            print(f"    -> per wave: VALU {c['SQ_INSTS_VALU'] / n['SQ_INSTS_VALU'] / w:.0f}, SALU {c['SQ_INSTS_SALU'] / n['SQ_INSTS_SALU'] / w:.0f}")
PY
cat "$OUT/summary.txt"
