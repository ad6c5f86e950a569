#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for f in 1 4 1 4; do
  echo "pedersen_waves=$f"
  for wl in arith_pedersen mixed; do
    ACVM_TUNING="pedersen_waves=$f" timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs 2>/dev/null | python tools/bench_line.py | cut -c1-200
  done
done
for f in 1 4; do
  echo "config5 pedersen_waves=$f"
  ACVM_TUNING="pedersen_waves=$f" timeout 900 python tools/t_config5.py 1000000 4096 3 0 plain 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print([ (t['solve_device_ms'], t['launches']) for t in d['tiles']], d['tiles'][-1].get('kernel_ms'))"
done
