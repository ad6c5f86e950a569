#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04i_tests.txt 2>&1
tail -6 gpurun_out/r04i_tests.txt
timeout 1200 python bench.py --steps 20 --warmup 5 2> gpurun_out/r04i_bench.err | tail -1 > gpurun_out/r04i_bench.json
python tools/bench_line.py < gpurun_out/r04i_bench.json
tail -c 900 gpurun_out/r04i_bench.json
