#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for t in ${SWEEP:-"heavy_epoch=1"}; do
  ACVM_TUNING="$t" timeout 900 python tools/t_config5.py 1000000 ${TILE:-4096} 3 0 ${MODE:-plain} 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$t', 'levels', d['levels'], [ (t['solve_device_ms'], t['launches']) for t in d['tiles']], d['tiles'][-1].get('kernel_ms'))"
done
