#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/r04e
cd /tmp
for wl in grumpkin ecdsa; do for mode in reset import; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04e/${wl}_$mode -o t -- python $ROOT/tools/t_step_mode.py $wl $mode > $ROOT/gpurun_out/r04e/${wl}_$mode.log 2>&1
  tail -1 $ROOT/gpurun_out/r04e/${wl}_$mode.log
  f=$(find $ROOT/gpurun_out/r04e/${wl}_$mode -name '*kernel_stats.csv' | head -1)
  echo "== $wl $mode"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
done; done
find $ROOT/gpurun_out/r04e -name '*.db' -delete; find $ROOT/gpurun_out/r04e -name '*kernel_trace.csv' -delete
