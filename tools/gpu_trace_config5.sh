#!/bin/bash
# tools/gpu_trace_config5.sh TAG [opcodes] [tile] -- rocprofv3 kernel trace (timestamps per launch) of tools/t_config5.py
set -u
TAG=${1:-rXX}; G=${2:-250000}; TILE=${3:-4096}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/c5_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python $ROOT/tools/t_config5.py $G $TILE 2 4 > "$OUT/plain.json" 2> "$OUT/plain.err"
cat "$OUT/plain.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/tools/t_config5.py $G $TILE 2 2 > "$OUT/trace.log" 2>&1
find "$OUT" -name '*.db' -delete
ls -la "$OUT/trace"/* | head
