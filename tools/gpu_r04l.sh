#!/bin/bash
# tools/gpu_r04l.sh -- the Grumpkin and ECDSA record kernels in workgroups of four waves (this tree) against one wave per workgroup (tools/ab/libacvm_amd_r04k.so)
# on ONE box: config 4 and ECDSA alone, then the workloads where these records share the chip with the gate kernel (config-5 mix, 10^6-opcode tile)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ecdsa.py tests/test_gpu_grumpkin.py tests/test_gpu_config5.py -x -q -m gpu 2>&1 | tail -2
for round in 1 2; do
  for lib in "" tools/ab/libacvm_amd_r04k.so; do
    echo "== ${lib:-this tree} (round $round)"
    for wl in grumpkin grumpkin grumpkin ecdsa ecdsa ecdsa mixed mixed; do ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload $wl --no-legs --no-cpu-baseline 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-210; done
    ACVM_AMD_LIB=$lib timeout 900 python tools/t_config5.py 1000000 4096 3 4 plain 2>&1 | tail -1 | cut -c1-700
  done
done
