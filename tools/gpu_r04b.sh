#!/bin/bash
# round 4: the pipelined byte-message hash kernel -- parity first, then the sweep and the bench legs
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_opcodes.py tests/test_gpu_planner_modes.py tests/test_gpu_fullsize.py tests/test_gpu_node.py tests/test_gpu_multirank.py tests/test_gpu_reuse.py tests/test_gpu_brillig.py -x -q > gpurun_out/r04b_tests.txt 2>&1
tail -15 gpurun_out/r04b_tests.txt
timeout 600 python tools/t_hash_sweep.py > gpurun_out/r04b_hash_sweep.txt 2>&1
cat gpurun_out/r04b_hash_sweep.txt
for wl in hash grumpkin ecdsa; do
  timeout 600 python bench.py --workload $wl 2> gpurun_out/r04b_bench_$wl.err | tail -1 > gpurun_out/r04b_bench_$wl.json
  python tools/bench_line.py < gpurun_out/r04b_bench_$wl.json
done
