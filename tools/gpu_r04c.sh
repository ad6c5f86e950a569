#!/bin/bash
# round 4: per-instance device limits, the error expression as data, opcode kinds; then r03's library against this tree's on one box (Grumpkin, ECDSA)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_brillig_limits.py tests/test_gpu_post_solve.py tests/test_gpu_node.py tests/test_gpu_boundary.py -x -q > gpurun_out/r04c_tests.txt 2>&1
tail -15 gpurun_out/r04c_tests.txt
bash tools/gpu_ab_lib.sh tools/ab/libacvm_amd_r03.so tools/t_grumpkin.py > gpurun_out/r04c_ab_grumpkin.txt 2>&1
cat gpurun_out/r04c_ab_grumpkin.txt
bash tools/gpu_ab_lib.sh tools/ab/libacvm_amd_r03.so tools/t_ecdsa.py > gpurun_out/r04c_ab_ecdsa.txt 2>&1
cat gpurun_out/r04c_ab_ecdsa.txt
rocm-smi --showclocks --showpower 2>&1 | head -30
