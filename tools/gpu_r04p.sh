#!/bin/bash
# tools/gpu_r04p.sh -- ECDSA decompression shortcut + the Grumpkin / Pedersen scratch changes against the mid-round library (tools/ab/libacvm_amd_r04d.so)
# on ONE box: parity tests, then A-B-A-B of the per-opcode timing scripts
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ecdsa.py tests/test_gpu_brillig.py tests/test_gpu_grumpkin.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_ab_lib.sh tools/ab/libacvm_amd_r04d.so tools/t_ecdsa.py
bash tools/gpu_ab_lib.sh tools/ab/libacvm_amd_r04d.so tools/t_grumpkin.py
for i in 1 2; do
  timeout 600 python bench.py --workload ecdsa --no-legs 2> /dev/null | tail -1 | python tools/bench_line.py
  ACVM_AMD_LIB=tools/ab/libacvm_amd_r04d.so timeout 600 python bench.py --workload ecdsa --no-legs 2> /dev/null | tail -1 | python tools/bench_line.py
  timeout 600 python bench.py --workload grumpkin --no-legs 2> /dev/null | tail -1 | python tools/bench_line.py
  ACVM_AMD_LIB=tools/ab/libacvm_amd_r04d.so timeout 600 python bench.py --workload grumpkin --no-legs 2> /dev/null | tail -1 | python tools/bench_line.py
done
