#!/bin/bash
# per-launch counters of the ECDSA / Grumpkin record kernels behind a reset and behind an import (tools/t_step_mode.py)
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd /tmp
for wl in ecdsa grumpkin; do for mode in reset import; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    name=$(echo $set | tr ' ' '_')
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${wl}_${mode}_$name -o p -- python $ROOT/tools/t_step_mode.py $wl $mode > /dev/null 2>&1
  done
done; done
find $OUT -name '*.db' -delete
python - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
tab = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "*"))):
    base = os.path.basename(d)
    wl, mode = base.split("_")[0], base.split("_")[1]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "record_level_kernel" in k:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            v = v[1:]  # (the first launch follows the handle's first import in both modes)
            tab[(wl, c)][mode] = sum(v) / max(len(v), 1)
for (wl, c), m in sorted(tab.items()):
    a, b = m.get("reset", 0), m.get("import", 0)
    print(f"{wl:9s} {c:26s} reset {a:16.0f} import {b:16.0f} ratio {b / a if a else 0:.3f}")
PY
