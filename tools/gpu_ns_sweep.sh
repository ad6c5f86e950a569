#!/bin/bash
# tools/gpu_ns_sweep.sh TAG "tuning;tuning;..." -- the north-star shape (10k gates + 8 Pedersen, 2^20 in tiles of 2^17) under a list of ACVM_TUNING settings
TAG=${1:-rXX}; LIST=${2:-";pedersen_epoch=8,pedersen_latency=4"}
OUT=gpurun_out/ns_$TAG.txt
: > $OUT
IFS=';' read -ra SPECS <<< "$LIST"
for spec in "${SPECS[@]}"; do
  ACVM_TUNING="$spec" ACVM_BENCH_NO_PMC=1 python bench.py --workload arith_pedersen --total-log2 20 --tile-log2 17 --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-digest 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(json.dumps({'spec':'$spec','value':round(d['value']),'ms_per_step':round(d['ms_per_step'],2),'levels':d['config']['levels'],'gate_frac':round(r['frac'],4),'gate_ms_tile':round(r['kernel_ms_per_tile'],2),'others':{k:round(v,2) for k,v in r['other_kernels_ms_per_tile'].items()}}))" >> $OUT
done
cat $OUT
