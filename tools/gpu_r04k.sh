#!/bin/bash
# tools/gpu_r04k.sh -- (1) the fused secp256k1 fold / multiplied shifts of secp256r1 against the build before (tools/ab/libacvm_amd_r04j.so): parity, timing;
# (2) the Grumpkin record kernel in workgroups of four waves, without and with a barrier per ladder window, against one wave per workgroup: config 4, five runs each
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ecdsa.py tests/test_gpu_secp_probe.py tests/test_gpu_brillig.py -x -q -m gpu 2>&1 | tail -2
bash tools/gpu_ab_lib.sh tools/ab/libacvm_amd_r04j.so tools/t_ecdsa.py
for lib in "" tools/ab/libacvm_amd_r04j.so; do ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload ecdsa --no-legs 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-150; done
for round in 1 2; do
  for lib in "" tools/ab/libacvm_amd_exp_b256.so tools/ab/libacvm_amd_exp_b256bar.so; do
    echo "== ${lib:-this tree} (round $round)"
    ACVM_AMD_LIB=$lib timeout 600 python tools/t_grumpkin.py 2>&1 | tail -1
    for i in 1 2 3 4 5; do ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload grumpkin --no-legs --no-cpu-baseline 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-100; done
  done
done
