#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ACVM_TUNING="pedersen_window_bits=22" timeout 900 python -m pytest tests/test_gpu_grumpkin.py tests/test_gpu_opcodes.py -x -q 2>&1 | tail -3
ACVM_TUNING="pedersen_window_bits=22,pedersen_waves=1" timeout 900 python -m pytest tests/test_gpu_grumpkin.py -x -q 2>&1 | tail -2
ACVM_TUNING="pedersen_window_bits=22" timeout 600 python tools/t_fuzz.py 15 700 2>&1 | tail -1
for t in "pedersen_window_bits=0" "pedersen_window_bits=22" "pedersen_window_bits=0" "pedersen_window_bits=22"; do
  echo "== $t"
  for wl in arith_pedersen mixed grumpkin; do
    ACVM_TUNING="$t" timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | python tools/bench_line.py | cut -c1-120
  done
done
ACVM_TUNING="pedersen_window_bits=22" timeout 300 python tools/t_pedersen_sweep.py 2>&1 | grep "B  65536\|B   4096 records 4"
