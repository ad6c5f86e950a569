#!/bin/bash
# tools/gpu_ab4.sh "tuning A" ... -- the headline workload (5 steps) and the north-star shape under tuning presets
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for t in "$@"; do
  echo "== $t"
  ACVM_TUNING="$t" timeout 600 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline --no-end-to-end 2>/dev/null | python tools/bench_line.py | cut -c1-150
  ACVM_TUNING="$t" timeout 600 python bench.py --workload arith_pedersen --steps 10 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | python tools/bench_line.py | cut -c1-120
done
