#!/bin/bash
# tools/gpu_sweep_tuning.sh "k=v,k=v k=v ..." CMD... -- CMD once per tuning set (ACVM_TUNING, tuning.hpp lists the keys; "-" = the defaults), value and
# ms_per_step of its JSON line. One box, one pass: order the sets so that the defaults come first and last.
SETS=$1; shift
for t in $SETS; do
  if [ "$t" = "-" ]; then unset ACVM_TUNING; else export ACVM_TUNING="$t"; fi
  echo -n "$t  "; timeout 900 "$@" 2>&1 | grep '"value"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
