#!/bin/bash
# round 4, first call: the node / multirank / table tests after the refactor, the default bench line, the hash sweep before any kernel change
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_node.py tests/test_gpu_multirank.py tests/test_gpu_parity.py tests/test_gpu_grumpkin.py tests/test_gpu_ecdsa.py -x -q > gpurun_out/r04a_tests.txt 2>&1
tail -15 gpurun_out/r04a_tests.txt
timeout 600 python tools/t_hash_sweep.py > gpurun_out/r04a_hash_sweep.txt 2>&1
cat gpurun_out/r04a_hash_sweep.txt
ACVM_TUNING=hash_chain=0 timeout 600 python tools/t_hash_sweep.py > gpurun_out/r04a_hash_sweep_nochain.txt 2>&1
cat gpurun_out/r04a_hash_sweep_nochain.txt
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/r04a_bench.err | tail -1 > gpurun_out/r04a_bench.json
python tools/bench_line.py < gpurun_out/r04a_bench.json
tail -c 1500 gpurun_out/r04a_bench.json
