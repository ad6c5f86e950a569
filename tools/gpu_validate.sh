#!/bin/bash
# Full validation on the GPU box: the gpu test suite, the seeded planner-mode sweeps against the oracle, ECDSA timing per curve, the default bench line.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${1:-val}
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/${tag}_pytest.txt
( timeout 900 python tools/t_fuzz.py 30 700 2>&1 | tail -5; timeout 600 python tools/t_fuzz.py 30 300 wild 2>&1 | tail -3 ) > gpurun_out/${tag}_fuzz.txt
timeout 300 python tools/t_ecdsa.py > gpurun_out/${tag}_ecdsa.txt 2>&1
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_pytest.txt gpurun_out/${tag}_fuzz.txt gpurun_out/${tag}_ecdsa.txt
python tools/bench_line.py < gpurun_out/${tag}_bench.json 2>/dev/null || tail -c 600 gpurun_out/${tag}_bench.json
