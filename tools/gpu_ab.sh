#!/bin/bash
# tools/gpu_ab.sh "tuning A" "tuning B" ... -- A/B of tuning presets on one box: the 10^6-opcode circuit (plain tile of 4 096), the north-star shape, the 10k mix
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for t in "$@"; do
  echo "== $t"
  ACVM_TUNING="$t" timeout 900 python tools/t_config5.py 1000000 4096 3 0 plain 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config5', [ (t['solve_device_ms'], t['launches']) for t in d['tiles']], d['tiles'][-1].get('kernel_ms'))"
  for wl in arith_pedersen mixed; do
    ACVM_TUNING="$t" timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs 2>/dev/null | python tools/bench_line.py | cut -c1-200
  done
done
