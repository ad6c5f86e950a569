#!/bin/bash
# tools/gpu_c5.sh TAG G TILE [env assignments...] -- tools/t_config5.py under a set of planner / driver switches, one JSON line each
TAG=$1; G=$2; TILE=$3; shift 3
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg" >> gpurun_out/c5_$TAG.txt
  env $cfg python tools/t_config5.py $G $TILE 3 4 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'levels': d['levels'], 'tiles': [(t['solve_device_ms'], t['launches'], t['witnesses_per_s']) for t in d['tiles']], 'kernel_ms': d['tiles'][-1].get('kernel_ms'), 'audit': d['audit']['bit_exact']}))" >> gpurun_out/c5_$TAG.txt 2>&1
done
cat gpurun_out/c5_$TAG.txt
