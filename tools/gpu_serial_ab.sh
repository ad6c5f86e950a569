#!/bin/bash
# tools/gpu_serial_ab.sh TAG -- the gate kernel's asm-block column scans (fr_blocks.inc; only gate_eval.hpp uses them): the probes, the GPU suite on the tree's
# library, then A-B-A-B against the build before it (tools/ab/libacvm_amd_base.so) on the metric's step, the north-star shape, config 4 and a 10^6-opcode tile
TAG=${1:-rXX}
OUT=gpurun_out/blocks_$TAG.txt
mkdir -p gpurun_out
{
echo "== probes"; timeout 60 tools/mad_hazard_probe; timeout 120 tools/serial_mul_probe
echo "== GPU suite (tree)"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(json.dumps({'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],3),'frac':r.get('frac'),'kernel_ms_per_tile':r.get('kernel_ms_per_tile')}))"; }
for round in 1 2; do
  for lib in "" tools/ab/libacvm_amd_base.so; do
    echo "== metric step, lib=${lib:-tree} (round $round)"
    ACVM_AMD_LIB=$lib ACVM_BENCH_NO_PMC=1 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | line
    echo "== north-star shape 2^20, lib=${lib:-tree} (round $round)"
    ACVM_AMD_LIB=$lib ACVM_BENCH_NO_PMC=1 timeout 600 python bench.py --workload arith_pedersen --total-log2 20 --tile-log2 17 --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-digest 2>/dev/null | line
  done
done
for lib in "" tools/ab/libacvm_amd_base.so "" tools/ab/libacvm_amd_base.so; do
  echo "== config 4 (grumpkin), lib=${lib:-tree}"
  ACVM_AMD_LIB=$lib timeout 600 python tools/t_leg.py grumpkin 16 16 30 5 2>&1 | tail -1
done
for lib in "" tools/ab/libacvm_amd_base.so "" tools/ab/libacvm_amd_base.so; do
  echo "== 10^6-opcode tile of 8 192, slot reuse, lib=${lib:-tree}"
  ACVM_AMD_LIB=$lib timeout 900 python tools/t_config5.py 1000000 8192 3 4 reuse 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([(t['solve_device_ms'], round(t['witnesses_per_s'])) for t in d['tiles']])"
done
} > $OUT 2>&1
cat $OUT
