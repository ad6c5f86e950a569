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
