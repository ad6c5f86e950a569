#!/bin/bash
# tools/gpu_round.sh TAG -- one GPU round: GPU tests, the four bench workloads, profiles of each (see gpu_profile.sh)
TAG=${1:-rXX}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/pytest_$TAG.txt
cat gpurun_out/pytest_$TAG.txt
for wl in arith hash grumpkin arith_pedersen mixed; do
  python bench.py --workload $wl 2>&1 | tail -1 > gpurun_out/bench_${TAG}_$wl.json
  cat gpurun_out/bench_${TAG}_$wl.json
  bash tools/gpu_profile.sh ${TAG}_$wl --workload $wl > /dev/null 2>&1
done
ls gpurun_out
