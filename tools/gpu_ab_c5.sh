#!/bin/bash
# tools/gpu_ab_c5.sh "variant.so variant.so ..." [spec] -- the config-5 tile (tools/t_c5_sweep.py, one setting) with the tree's library and with each variant build
# (tools/build_variant.sh) in turn, twice round: A B C A B C on one box
SPEC=${2:-tile=8192}
for round in 1 2; do
  echo "== tree (round $round)"; ACVM_SWEEP="$SPEC" python tools/t_c5_sweep.py 1000000 5 2>/dev/null | cut -c1-200
  for so in $1; do echo "== $so (round $round)"; ACVM_AMD_LIB=$so ACVM_SWEEP="$SPEC" python tools/t_c5_sweep.py 1000000 5 2>/dev/null | cut -c1-200; done
done
