#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_reuse.py tests/test_gpu_boundary.py -x -q > gpurun_out/r04h_tests.txt 2>&1
tail -8 gpurun_out/r04h_tests.txt
for rep in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-end-to-end --no-digest 2>/dev/null | tail -1 | python tools/bench_line.py
  timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-end-to-end --no-digest --no-pipeline 2>/dev/null | tail -1 | python tools/bench_line.py
done | tee gpurun_out/r04h_abab.txt
