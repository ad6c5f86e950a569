#!/bin/bash
# tools/gpu_r04q.sh -- SchnorrVerify's window rows (lane-major + slot swizzle + early request) against the two libraries before it, on ONE box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_grumpkin.py -x -q -m gpu 2>&1 | tail -3
for round in 1 2; do
  for lib in "" tools/ab/libacvm_amd_r04f.so tools/ab/libacvm_amd_r04d.so; do
    echo "== ${lib:-this tree} (round $round)"
    ACVM_AMD_LIB=$lib timeout 600 python tools/t_grumpkin.py 2>&1 | tail -1
    for i in 1 2 3; do ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload grumpkin --no-legs --no-cpu-baseline 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-110; done
  done
done
