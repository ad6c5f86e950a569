#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_grumpkin.py tests/test_gpu_grumpkin_probe.py tests/test_gpu_ecdsa.py tests/test_gpu_brillig.py tests/test_gpu_opcodes.py tests/test_gpu_parity.py -x -q > gpurun_out/r04g_tests.txt 2>&1
tail -5 gpurun_out/r04g_tests.txt
for w in grumpkin ecdsa; do timeout 250 python tools/t_step_gap.py $w 2>&1 | sed -n "1,2p"; done | tee gpurun_out/r04g_step_gap.txt
timeout 300 python tools/t_grumpkin.py 2>&1 | tail -3
timeout 300 python tools/t_ecdsa.py 2>&1 | tail -2
timeout 300 python tools/t_pedersen_sweep.py 2>&1 | tail -12
for wl in arith_pedersen mixed; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline 2> gpurun_out/r04g_bench_$wl.err | tail -1 > gpurun_out/r04g_bench_$wl.json
  python tools/bench_line.py < gpurun_out/r04g_bench_$wl.json
done
