#!/bin/bash
# tools/collect_profiles.sh TAG -- copy what tools/gpu_round5.sh TAG left under gpurun_out/ (scratch) into profiles/ (tracked): the bench lines,
# the rocprofv3 summaries (tools/prof_summary.py), the SQ counters, config 5 at circuit size and the timelines.
TAG=${1:?tag}
cd "$(dirname "$0")/.."
for wl in arith hash grumpkin ecdsa arith_pedersen mixed config5 n2_shared_gpu; do
  [ -s gpurun_out/bench_${TAG}_$wl.json ] && cp gpurun_out/bench_${TAG}_$wl.json profiles/${TAG}_bench_$wl.json
done
[ -d gpurun_out/prof_${TAG}_bench ] && python tools/prof_summary.py gpurun_out/prof_${TAG}_bench > profiles/${TAG}_profile_bench.txt
for wl in arith hash grumpkin ecdsa; do
  [ -d gpurun_out/prof_${TAG}_$wl ] && python tools/prof_summary.py gpurun_out/prof_${TAG}_$wl > profiles/${TAG}_profile_$wl.txt
done
[ -s gpurun_out/sq_${TAG}_arith/summary.txt ] && cp gpurun_out/sq_${TAG}_arith/summary.txt profiles/${TAG}_sq_arith.txt
[ -s gpurun_out/pytest_${TAG}.txt ] && cp gpurun_out/pytest_${TAG}.txt profiles/${TAG}_pytest.txt
[ -s gpurun_out/c5c_${TAG}/summary.txt ] && cp gpurun_out/c5c_${TAG}/summary.txt profiles/${TAG}_config5_counters.txt
[ -s gpurun_out/c5c_${TAG}/timeline.txt ] && cp gpurun_out/c5c_${TAG}/timeline.txt profiles/${TAG}_config5_timeline.txt
for f in config5_${TAG}_1m.txt config5_${TAG}_2p17.json node_create_${TAG}.txt config5_${TAG}_timeline.txt arith_${TAG}_timeline.txt hash_sweep_${TAG}.txt; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/${TAG}_${f/_${TAG}/}
done
ls -la profiles | grep " ${TAG}_"
