#!/bin/bash
# tools/gpu_blocks_abc.sh -- A-B-C-A-B-C on one box: the tree's library against two variants in which EVERY fr29_mul of a translation unit takes the asm-block
# form (NOTEBOOK 6.15):  bash tools/build_variant.sh inv "-DFR_BLOCKS_ALL" "kernels.hip.o"   (the inversion kernel)
#                        bash tools/build_variant.sh ped "-DFR_BLOCKS_ALL" "kernels_grumpkin.hip.o"   (Pedersen / Grumpkin)
# on the north-star shape at 2^20, the metric's step and config 4. Result (profiles/r06x_asm_blocks_other_kernels.txt): +0.3 % each, not taken.
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(json.dumps({'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],3),'frac':round(r.get('frac') or 0,4),'kernel_ms_per_tile':r.get('kernel_ms_per_tile'),'others':r.get('other_kernels_ms_per_tile')}))"; }
for round in 1 2; do
  for lib in "" tools/ab/libacvm_amd_inv.so tools/ab/libacvm_amd_ped.so; do
    echo "== north-star shape 2^20, lib=${lib:-tree} (round $round)"
    ACVM_AMD_LIB=$lib ACVM_BENCH_NO_PMC=1 timeout 600 python bench.py --workload arith_pedersen --total-log2 20 --tile-log2 17 --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-digest 2>/dev/null | line
  done
  for lib in "" tools/ab/libacvm_amd_inv.so; do
    echo "== metric step, lib=${lib:-tree} (round $round)"
    ACVM_AMD_LIB=$lib ACVM_BENCH_NO_PMC=1 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | line
  done
  for lib in "" tools/ab/libacvm_amd_ped.so; do
    echo "== config 4, lib=${lib:-tree} (round $round)"
    ACVM_AMD_LIB=$lib timeout 600 python tools/t_leg.py grumpkin 16 16 30 5 2>&1 | tail -1
  done
done
