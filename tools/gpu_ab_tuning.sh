#!/bin/bash
# tools/gpu_ab_tuning.sh "key=value[,key=value]" ROUNDS CMD... -- A-B-A-B of one planner / scheduler mode on one box: CMD runs with the library's
# defaults and with ACVM_TUNING="key=value" in turn, ROUNDS times each (tuning.hpp lists the keys). One script instead of a file per experiment.
#   tools/gpu_ab_tuning.sh relax=0 2 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline --no-end-to-end
ALT=$1; ROUNDS=$2; shift 2
for round in $(seq 1 "$ROUNDS"); do
  echo "== defaults (round $round)"; timeout 900 "$@" 2>&1 | tail -4
  echo "== $ALT (round $round)"; ACVM_TUNING="$ALT" timeout 900 "$@" 2>&1 | tail -4
done
