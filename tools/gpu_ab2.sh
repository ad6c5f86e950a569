#!/bin/bash
# tools/gpu_ab2.sh WORKLOAD "tuning A" "tuning B" ... -- A/B of tuning presets on one bench workload
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
wl=$1; shift
for t in "$@"; do
  echo -n "$t: "; ACVM_TUNING="$t" timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs 2>/dev/null | python tools/bench_line.py | cut -c1-240
done
