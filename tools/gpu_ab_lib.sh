#!/bin/bash
# tools/gpu_ab_lib.sh OTHER.so SCRIPT [args...] -- A-B-A-B of two builds of the library on one box: SCRIPT (a tools/t_*.py timing script) runs with
# the tree's library and with OTHER.so (ACVM_AMD_LIB) in turn, twice each
OTHER=$1; shift
for round in 1 2; do
  echo "== this tree (round $round)"; timeout 600 python "$@" 2>&1 | tail -8
  echo "== $OTHER (round $round)"; ACVM_AMD_LIB=$OTHER timeout 600 python "$@" 2>&1 | tail -8
done
