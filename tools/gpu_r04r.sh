#!/bin/bash
# tools/gpu_r04r.sh LIB -- this tree against another build of the library on ONE box: the Grumpkin opcodes alone, config 4 and the north-star shape
OTHER=${1:-tools/ab/libacvm_amd_r04d.so}
for round in 1 2; do
  for lib in "" $OTHER; do
    echo "== ${lib:-this tree} (round $round)"
    ACVM_AMD_LIB=$lib timeout 600 python tools/t_grumpkin.py 2>&1 | tail -3
    for i in 1 2 3; do ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload grumpkin --no-legs --no-cpu-baseline 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-110; done
    ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload arith_pedersen --no-legs --no-cpu-baseline 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-180
  done
done
