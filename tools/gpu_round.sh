#!/bin/bash
# tools/gpu_round.sh TAG -- one GPU round: GPU tests, the bench workloads (full JSON lines), rocprofv3 --kernel-trace --stats of the
# default bench command, SQ counters of the arith kernel, config 5 at circuit size. Summaries are made locally (tools/prof_summary.py)
TAG=${1:-rXX}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/pytest_$TAG.txt
cat gpurun_out/pytest_$TAG.txt
timeout 900 python bench.py 2> gpurun_out/bench_${TAG}_arith.err | tail -1 > gpurun_out/bench_${TAG}_arith.json
python tools/bench_line.py < gpurun_out/bench_${TAG}_arith.json
for wl in hash grumpkin arith_pedersen mixed; do
  timeout 900 python bench.py --workload $wl 2> gpurun_out/bench_${TAG}_$wl.err | tail -1 > gpurun_out/bench_${TAG}_$wl.json
  python tools/bench_line.py < gpurun_out/bench_${TAG}_$wl.json
done
# the driver's command under rocprofv3 (PMC passes of bench.py itself off: one trace of one process)
ROOT=$(pwd)
mkdir -p gpurun_out/prof_${TAG}_bench
( cd /tmp && ACVM_BENCH_NO_PMC=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_bench/trace" -o trace -- python $ROOT/bench.py --no-cpu-baseline > "$ROOT/gpurun_out/prof_${TAG}_bench/trace.log" 2>&1 )
find gpurun_out/prof_${TAG}_bench -name '*.db' -delete
# keep the stats and drop the per-launch trace if it is large
find gpurun_out/prof_${TAG}_bench -name '*kernel_trace.csv' -size +20M -delete
for wl in arith grumpkin hash; do
  timeout 900 bash tools/gpu_profile.sh ${TAG}_$wl --workload $wl > /dev/null 2>&1
  find gpurun_out/prof_${TAG}_$wl -name '*kernel_trace.csv' -size +20M -delete
done
timeout 600 bash tools/gpu_pmc_sq.sh ${TAG}_arith --workload arith > /dev/null 2>&1
timeout 900 python tools/t_config5.py 1000000 4096 2 4 2>&1 | tail -1 > gpurun_out/config5_${TAG}_1m.json
cat gpurun_out/config5_${TAG}_1m.json | cut -c1-1500
ls gpurun_out | head -50
