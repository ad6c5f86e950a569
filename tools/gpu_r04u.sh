#!/bin/bash
# tools/gpu_r04u.sh -- gate sums of three and more terms through fr29_weak (this tree) against conditional subtractions (tools/ab/libacvm_amd_r04u.so) on ONE box:
# parity, then the metric's workload, the north-star shape and a 10^6-opcode tile, the two builds interleaved
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_planner_modes.py -x -q -m gpu 2>&1 | tail -2
for round in 1 2 3; do
  for lib in "" tools/ab/libacvm_amd_r04u.so; do
    echo "== ${lib:-this tree} (round $round)"
    ACVM_AMD_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline --no-end-to-end 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-150
    ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload arith_pedersen --steps 10 --warmup 3 --no-legs --no-cpu-baseline --no-end-to-end 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-150
  done
done
for lib in "" tools/ab/libacvm_amd_r04u.so; do ACVM_AMD_LIB=$lib timeout 900 python tools/t_config5.py 1000000 4096 3 4 plain 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 tile', [(t['solve_device_ms'], round(t['witnesses_per_s'])) for t in d['tiles']])"; done
