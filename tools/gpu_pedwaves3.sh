#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_planner_modes.py tests/test_gpu_grumpkin.py tests/test_gpu_config5.py -x -q 2>&1 | tail -3
for wl in arith_pedersen mixed grumpkin; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs 2>/dev/null | python tools/bench_line.py | cut -c1-230
done
timeout 900 python tools/t_config5.py 1000000 4096 3 0 plain 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print([ (t['solve_device_ms'], t['launches']) for t in d['tiles']], d['tiles'][-1].get('kernel_ms'))"
timeout 900 python tools/t_config5.py 1000000 8192 3 0 reuse 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print([ (t['solve_device_ms'], t['launches']) for t in d['tiles']], d['tiles'][-1].get('kernel_ms'))"
ACVM_TUNING="pedersen_waves=4" timeout 900 python tools/t_config5.py 1000000 8192 3 0 reuse 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('waves=4', [ (t['solve_device_ms'], t['launches']) for t in d['tiles']], d['tiles'][-1].get('kernel_ms'))"
