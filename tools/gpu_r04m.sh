#!/bin/bash
# tools/gpu_r04m.sh -- this tree against the build before it (tools/ab/libacvm_amd_r04w.so) on ONE box: Pedersen alone, the north-star shape, config 4, the config-5 mix, a 10^6-opcode tile
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_grumpkin.py tests/test_gpu_config5.py tests/test_gpu_planner_modes.py tests/test_gpu_reuse.py -x -q -m gpu 2>&1 | tail -2
for round in 1 2; do
  for lib in "" tools/ab/libacvm_amd_r04w.so; do
    echo "== ${lib:-this tree} (round $round)"
    ACVM_AMD_LIB=$lib timeout 300 python tools/t_pedersen_sweep.py 2>&1 | tail -6
    for wl in arith_pedersen arith_pedersen grumpkin grumpkin mixed; do ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload $wl --no-legs --no-cpu-baseline 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-210; done
    ACVM_AMD_LIB=$lib timeout 900 python tools/t_config5.py 1000000 4096 3 4 plain 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([ (t['solve_device_ms'], t['witnesses_per_s']) for t in d['tiles']])"
  done
done
