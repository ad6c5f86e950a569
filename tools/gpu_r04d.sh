#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_grumpkin.py tests/test_gpu_ecdsa.py tests/test_gpu_parity.py tests/test_gpu_opcodes.py tests/test_gpu_brillig.py tests/test_gpu_boundary.py tests/test_gpu_node.py -x -q > gpurun_out/r04d_tests.txt 2>&1
tail -8 gpurun_out/r04d_tests.txt
for w in grumpkin ecdsa hash; do timeout 250 python tools/t_step_gap.py $w 2>&1 | head -3; done | tee gpurun_out/r04d_step_gap.txt
for wl in hash grumpkin ecdsa; do
  timeout 600 python bench.py --workload $wl 2> gpurun_out/r04d_bench_$wl.err | tail -1 > gpurun_out/r04d_bench_$wl.json
  python tools/bench_line.py < gpurun_out/r04d_bench_$wl.json
done
