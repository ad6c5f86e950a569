#!/bin/bash
# tools/gpu_c5tiles.sh TAG G tile... -- throughput of the config-5 mix at G opcodes as a function of the tile size
TAG=$1; G=$2; shift 2
for T in "$@"; do
  echo "== tile $T" >> gpurun_out/c5tiles_$TAG.txt
  python tools/t_config5.py $G $T 3 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'levels': d['levels'], 'GB': d['witness_table_GB'], 'tiles': [(t['solve_device_ms'], t['digest_wall_ms'], t['witnesses_per_s']) for t in d['tiles']], 'kernel_ms': d['tiles'][-1].get('kernel_ms'), 'audit': d['audit']['bit_exact']}))" >> gpurun_out/c5tiles_$TAG.txt 2>&1
done
cat gpurun_out/c5tiles_$TAG.txt
