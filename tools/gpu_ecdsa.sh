#!/bin/bash
# ECDSA check on the GPU box: parity tests of the ECDSA paths, per-curve timing, then the ecdsa workload of bench.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ecdsa.py tests/test_gpu_brillig.py -x -q 2>&1 | tail -4
timeout 300 python tools/t_ecdsa.py 2>&1 | tail -3
timeout 600 python bench.py --workload ecdsa --steps 10 --warmup 3 --no-legs 2> gpurun_out/ecdsa_bench.err | tee gpurun_out/ecdsa_bench.json | python tools/bench_line.py
