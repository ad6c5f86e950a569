#!/bin/bash
# tools/gpu_r04t.sh -- HEAD against the mid-round build (tools/ab/libacvm_amd_r04d.so = commit e6df882, before this round's Grumpkin / Pedersen / ECDSA kernel changes),
# every workload, the two builds interleaved on ONE box, two rounds
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "" tools/ab/libacvm_amd_r04d.so; do
    echo "== ${lib:-HEAD} (round $round)"
    for wl in arith hash grumpkin ecdsa arith_pedersen mixed; do
      ACVM_AMD_LIB=$lib timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-legs --no-cpu-baseline --no-end-to-end 2> /dev/null | tail -1 | python tools/bench_line.py | cut -c1-150
    done
    ACVM_AMD_LIB=$lib timeout 900 python tools/t_config5.py 1000000 4096 3 4 plain 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 tile', [(t['solve_device_ms'], round(t['witnesses_per_s'])) for t in d['tiles']])"
  done
done
