#!/bin/bash
# tools/gpu_c5_counters.sh TAG [opcodes] [tile] [mode] [tuning] -- the config-5 tile as counters: one kernel-trace pass (timeline by class) and PMC passes
# (SQ_INSTS_VALU + SQ_WAVES; FETCH_SIZE; WRITE_SIZE -- each its own run, kernel-trace only) over tools/t_config5.py, summed per kernel class per tile,
# and the tile-wide VALU issue fraction = sum over classes of SQ_INSTS_VALU x 4 / (SIMDs x solve seconds x sclk)
set -u
TAG=${1:-rXX}; G=${2:-1000000}; TILE=${3:-4096}; MODE=${4:-reuse}; TUNE=${5:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/c5c_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
NT=3
ACVM_TUNING="$TUNE" rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/tools/t_config5.py $G $TILE $NT 2 $MODE > "$OUT/run.json" 2> "$OUT/trace.log"
tail -1 "$OUT/run.json" | cut -c1-1200
python $ROOT/tools/trace_timeline.py "$OUT/trace" > "$OUT/timeline.txt"
cat "$OUT/timeline.txt"
for set in "SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_')
  ACVM_TUNING="$TUNE" rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python $ROOT/tools/t_config5.py $G $TILE $NT 2 $MODE > "$OUT/$name.log" 2>&1
done
python $ROOT/tools/t_config5.py $G $TILE 6 0 $MODE > "$OUT/run_plain.json" 2>/dev/null
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*kernel_trace.csv' -size +30M -delete
python - "$OUT" $NT $TILE > "$OUT/summary.txt" <<'PY'
import csv, glob, sys, os, re, json
from collections import defaultdict
out, nt, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
CLASSES = [("arith", "arith_l"), ("inv", "inverse_batch_kernel"), ("light", "LightOp"), ("lightsl", "LightSlOp"), ("hash", "hash_coop_level_kernel"), ("hash", "HashOp"),
           ("pedersen", "pedersen_quad"), ("pedersen", "pedersen_bundle"), ("tables (once per process)", "_table_kernel"), ("tables (once per process)", "pedersen_seed"), ("grumpkin", "GrumpkinOp"), ("brillig", "BrilligOp"), ("digest", "digest_"), ("import", "import_witness"), ("exact", "exact_")]
def cls_of(n):
    for c, s in CLASSES:
        if s in n: return c
    return "other"
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        c = cls_of(r["Kernel_Name"]); tot[c][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[c][r["Counter_Name"]] += 1
# (the shader clock under this load is sampled by bench.py's config-5 leg, 2.1-2.2 GHz on the boxes of this pool: the fraction is printed for a range)
run = json.loads(open(os.path.join(out, "run_plain.json")).read().strip().splitlines()[-1])
ms = sorted(t["solve_device_ms"] for t in run["tiles"][1:])[len(run["tiles"][1:]) // 2]
print(f"# config-5 tile of {tile} instances: per kernel class and tile (sums over {nt} tiles / {nt}); solve_device_ms (median of 5 unprofiled tiles) = {ms}")
print(f"{'class':26s} {'launches':>9s} {'VALU wave-insts':>16s} {'waves':>12s} {'VALU/wave':>10s} {'read GB':>9s} {'write GB':>9s}")
total_valu = 0
for c in sorted(tot, key=lambda c: -tot[c].get("SQ_INSTS_VALU", 0)):
    v = tot[c].get("SQ_INSTS_VALU", 0) / nt; w = tot[c].get("SQ_WAVES", 0) / nt
    rd = tot[c].get("FETCH_SIZE", 0) / nt * 1024 * 2 / 1e9; wr = tot[c].get("WRITE_SIZE", 0) / nt * 1024 / 1e9
    if c not in ("exact", "other", "tables (once per process)", "import"): total_valu += v
    print(f"{c:26s} {cnt[c].get('SQ_INSTS_VALU', 0) / nt:9.0f} {v:16.0f} {w:12.0f} {v / w if w else 0:10.0f} {rd:9.2f} {wr:9.2f}")
if True:
    for clk in (1900, 2160, 2400):
        print(f"tile-wide VALU issue fraction at {clk} MHz: {total_valu * 4 / (1024 * ms / 1e3 * clk * 1e6):.3f}  (sum of the level classes' SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x {ms} ms x sclk))")
PY
cat "$OUT/summary.txt"
