#!/bin/bash
# tools/gpu_ab3.sh "tuning A" ... -- the north-star shape and the 10k mix under tuning presets
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for t in "$@"; do
  echo "== $t"
  for wl in arith_pedersen mixed; do
    ACVM_TUNING="$t" timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs 2>/dev/null | python tools/bench_line.py | cut -c1-120
  done
done
