#!/bin/bash
# tools/gpu_profile.sh TAG [bench args...] -- run on the GPU box (through gpurun) from the repo root:
#   1. rocprofv3 --kernel-trace --stats of bench.py           -> gpurun_out/prof_TAG/trace
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only)
#   3. the same two PMC passes over tools/microbench's 4 GiB uint4 copy = calibration of the counters on a
#      known byte count in the same access pattern (16 B/lane coalesced), as MI355X_MICROARCH.md §HBM prescribes
# Summaries are produced locally with tools/prof_summary.py and committed under profiles/.
set -u
TAG=${1:-rXX}
shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --inner --total-log2 16 $*"
echo "python bench.py --steps 3 --warmup 1 --inner --total-log2 16 $* (one tile of 2^16 instances per step)" > "$OUT/command.txt"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/pmc_write.log" 2>&1
if [ -x "$ROOT/tools/microbench" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/cal_fetch" -o pmc -- "$ROOT/tools/microbench" copy > "$OUT/cal_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/cal_write" -o pmc -- "$ROOT/tools/microbench" copy > "$OUT/cal_write.log" 2>&1
fi
# keep only the small CSVs (the merge-back limit is 64 MiB)
find "$OUT" -name '*.db' -delete
ls -laR "$OUT" | head -60
