#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_planner_modes.py tests/test_gpu_grumpkin.py -x -q 2>&1 | tail -3
ACVM_TUNING="pedersen_waves=1" timeout 900 python -m pytest tests/test_gpu_grumpkin.py tests/test_gpu_opcodes.py -x -q 2>&1 | tail -2
for f in 4 0; do
  echo "pedersen_waves=$f"
  for wl in arith_pedersen grumpkin; do
    ACVM_TUNING="pedersen_waves=$f" timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs 2>/dev/null | python tools/bench_line.py
  done
done
timeout 300 python tools/t_pedersen.py 2>&1 | tail -4
