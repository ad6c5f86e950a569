#!/bin/bash
# tools/gpu_trace_c5.sh TAG [opcodes] [tile] [tuning] -- kernel-trace timeline of the config-5 circuit (tools/trace_timeline.py)
set -u
TAG=${1:-rXX}; G=${2:-1000000}; TILE=${3:-4096}; TUNE=${4:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ACVM_TUNING="$TUNE" rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/tools/t_config5.py $G $TILE 3 2 > "$OUT/run.json" 2> "$OUT/trace.log"
tail -1 "$OUT/run.json" | cut -c1-900
python $ROOT/tools/trace_timeline.py "$OUT/trace" > "$OUT/timeline.txt"
cat "$OUT/timeline.txt"
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*kernel_trace.csv' -size +30M -delete
