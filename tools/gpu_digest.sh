#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_config5.py tests/test_gpu_reuse.py tests/test_gpu_node.py tests/test_gpu_post_solve.py -x -q 2>&1 | tail -3
timeout 600 python tools/t_fuzz.py 20 700 2>&1 | tail -1
for m in "4096 3 0 plain" "4096 3 0 fold" "8192 3 0 reuse"; do
  timeout 900 python tools/t_config5.py 1000000 $m 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['mode'], d['tile'], [ (t['solve_device_ms'], t['digest_wall_ms'], round(t['witnesses_per_s'])) for t in d['tiles']])"
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-legs 2>/dev/null | python tools/bench_line.py | cut -c1-200
