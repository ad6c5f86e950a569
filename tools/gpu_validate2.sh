#!/bin/bash
# gpu suite + planner-mode sweep + the 10^6-opcode circuit (plain, one tile pair) after a planner change
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${1:-val2}
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/${tag}_pytest.txt
( timeout 900 python tools/t_fuzz.py 20 700 2>&1 | tail -2; timeout 600 python tools/t_fuzz_node.py 20 2>&1 | tail -2 ) > gpurun_out/${tag}_fuzz.txt
timeout 900 python tools/t_config5.py 1000000 4096 3 4 plain > gpurun_out/${tag}_config5.txt 2>&1
cat gpurun_out/${tag}_pytest.txt gpurun_out/${tag}_fuzz.txt; cut -c1-900 gpurun_out/${tag}_config5.txt | tail -3
