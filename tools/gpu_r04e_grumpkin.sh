#!/bin/bash
# tools/gpu_r04e_grumpkin.sh TAG -- the Grumpkin class after a change: its parity tests, three bench lines, the HBM counters of one tile
TAG=${1:-r04e}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_grumpkin.py tests/test_gpu_fullsize.py tests/test_gpu_config5.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2 3; do timeout 600 python bench.py --workload grumpkin 2> /dev/null | tail -1 > gpurun_out/bench_${TAG}_grumpkin.json; python tools/bench_line.py < gpurun_out/bench_${TAG}_grumpkin.json; done
timeout 600 python bench.py --workload arith_pedersen 2> /dev/null | tail -1 > gpurun_out/bench_${TAG}_arith_pedersen.json; python tools/bench_line.py < gpurun_out/bench_${TAG}_arith_pedersen.json
timeout 900 bash tools/gpu_profile.sh ${TAG}_grumpkin --workload grumpkin > /dev/null 2>&1
find gpurun_out/prof_${TAG}_grumpkin -name '*kernel_trace.csv' -size +20M -delete
python tools/prof_summary.py gpurun_out/prof_${TAG}_grumpkin | grep -A8 "HBM traffic\|kernel stats"
