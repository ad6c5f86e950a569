#!/bin/bash
# tools/gpu_pmc_cmd.sh TAG CMD... -- SQ counters (own PMC passes, kernel-trace only) of one command, per-launch averages per kernel
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sqc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  name=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- "$@" > "$OUT/$name.log" 2>&1
done
find "$OUT" -name '*.db' -delete
python - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-44:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    if "level" in k or "quad" in k:
        print(k)
        for c in sorted(tot[k]):
            print(f"    {c:28s} {tot[k][c] / cnt[k][c]:14.0f}   ({cnt[k][c]} launches)")
PY
