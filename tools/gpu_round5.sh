#!/bin/bash
# tools/gpu_round5.sh TAG -- the evidence of a round in one GPU call: bench lines of every workload (the default one with its legs),
# rocprofv3 --kernel-trace --stats of the driver's command, per-workload profiles with HBM counters, SQ counters of the gate kernel,
# config 5 at circuit size (plain / folded digest / slot reuse, through the batch API and through the node driver) and its timeline.
TAG=${1:-rXX}
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_${TAG}_arith.err | tail -1 > gpurun_out/bench_${TAG}_arith.json
python tools/bench_line.py < gpurun_out/bench_${TAG}_arith.json
for wl in hash grumpkin ecdsa arith_pedersen mixed; do
  timeout 900 python bench.py --workload $wl 2> gpurun_out/bench_${TAG}_$wl.err | tail -1 > gpurun_out/bench_${TAG}_$wl.json
  python tools/bench_line.py < gpurun_out/bench_${TAG}_$wl.json
done
timeout 1500 python bench.py --workload config5 --steps 2 --warmup 1 2> gpurun_out/bench_${TAG}_config5.err | tail -1 > gpurun_out/bench_${TAG}_config5.json
python tools/bench_line.py < gpurun_out/bench_${TAG}_config5.json
# the driver's command under rocprofv3 (PMC passes and legs of bench.py itself off: one trace of one process)
mkdir -p gpurun_out/prof_${TAG}_bench
echo "python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs (the driver's command: 2^20 instances in 8 tiles of 2^17 per step; bench.py's own PMC passes off)" > gpurun_out/prof_${TAG}_bench/command.txt
( cd /tmp && ACVM_BENCH_NO_PMC=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_bench/trace" -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > "$ROOT/gpurun_out/prof_${TAG}_bench/trace.log" 2>&1 )
find gpurun_out/prof_${TAG}_bench -name '*.db' -delete
find gpurun_out/prof_${TAG}_bench -name '*kernel_trace.csv' -size +20M -delete
for wl in arith grumpkin hash ecdsa; do
  timeout 900 bash tools/gpu_profile.sh ${TAG}_$wl --workload $wl > /dev/null 2>&1
  find gpurun_out/prof_${TAG}_$wl -name '*kernel_trace.csv' -size +20M -delete
done
timeout 600 bash tools/gpu_pmc_sq.sh ${TAG}_arith --workload arith > /dev/null 2>&1
for m in "4096 3 4 plain" "4096 3 4 fold" "8192 3 4 reuse"; do timeout 900 python tools/t_config5.py 1000000 $m 2>&1 | tail -1; done > gpurun_out/config5_${TAG}_1m.txt
timeout 900 python tools/t_node.py 1000000 16384 4096 1 plain 2>&1 | tail -1 >> gpurun_out/config5_${TAG}_1m.txt
timeout 900 python tools/t_node.py 1000000 32768 8192 1 reuse 2>&1 | tail -1 >> gpurun_out/config5_${TAG}_1m.txt
cut -c1-700 gpurun_out/config5_${TAG}_1m.txt
timeout 900 bash tools/gpu_trace_c5.sh ${TAG} 1000000 4096 > gpurun_out/config5_${TAG}_timeline.txt 2>&1
tail -25 gpurun_out/config5_${TAG}_timeline.txt
ls gpurun_out | head -80
timeout 600 bash tools/gpu_trace_arith.sh ${TAG} > gpurun_out/arith_${TAG}_timeline.txt 2>&1
tail -12 gpurun_out/arith_${TAG}_timeline.txt
timeout 300 python tools/t_hash_sweep.py > gpurun_out/hash_sweep_${TAG}.txt 2>&1
cat gpurun_out/hash_sweep_${TAG}.txt
# an N = 2 line on the one GPU (two ranks sharing it): cpu_baseline, parity and roofline.traffic present at N > 1
ACVM_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --steps 5 --warmup 2 2> gpurun_out/bench_${TAG}_n2.err | tail -1 > gpurun_out/bench_${TAG}_n2_shared_gpu.json
python tools/bench_line.py < gpurun_out/bench_${TAG}_n2_shared_gpu.json
