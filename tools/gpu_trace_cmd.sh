#!/bin/bash
# tools/gpu_trace_cmd.sh TAG CMD... -- rocprofv3 --kernel-trace of one command; prints the per-kernel launch list of the last 40 launches
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/trace_$TAG
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/trace_$TAG" -o t -- "$@" > "$ROOT/gpurun_out/trace_$TAG/log.txt" 2>&1 )
f=$(find gpurun_out/trace_$TAG -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-40:]:
    n = r['Kernel_Name'].split('(')[0][-44:]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:12.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} q{r.get('Queue_Id','?')} {n} grid {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')} wg {r.get('Workgroup_Size_X','')} vgpr {r.get('VGPR_Count','')} lds {r.get('LDS_Block_Size','')}")
PY
find gpurun_out/trace_$TAG -name '*.db' -delete
