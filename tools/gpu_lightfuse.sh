#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_planner_modes.py tests/test_gpu_opcodes.py tests/test_gpu_config5.py tests/test_gpu_reuse.py -x -q 2>&1 | tail -4
timeout 600 python tools/t_fuzz.py 20 700 2>&1 | tail -1
for f in 1 0; do
  echo "light_fuse=$f"
  ACVM_TUNING="light_fuse=$f" timeout 900 python tools/t_config5.py 1000000 4096 3 0 plain 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print([ (t['solve_device_ms'], t['launches']) for t in d['tiles']], d['tiles'][-1].get('kernel_ms'))"
  ACVM_TUNING="light_fuse=$f" timeout 600 python bench.py --workload mixed --steps 10 --warmup 3 --no-legs 2>/dev/null | python tools/bench_line.py
done
