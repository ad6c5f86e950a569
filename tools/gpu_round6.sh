#!/bin/bash
# tools/gpu_round6.sh TAG -- the evidence of round 6 in one GPU call (tools/collect_profiles.sh TAG copies it into profiles/): the GPU suite, bench lines of
# every workload (the default one with its legs), rocprofv3 --kernel-trace --stats of the driver's command, per-workload profiles with HBM counters,
# SQ counters of the gate kernel, config 5 at circuit size (batch API, node driver, counters + timeline at the leg's tile, the byte-wise tree
# digest, the 2^17 per-GPU share with a 256-instance audit), node creation with 1 and 8 lanes, N = 2 on the shared GPU.
TAG=${1:-rXX}
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -x tools/microbench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip  # the 4 GiB copy that calibrates the HBM counters (gpu_profile.sh)
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/pytest_${TAG}.txt; cat gpurun_out/pytest_${TAG}.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/pytest_${TAG}.txt
timeout 1200 python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_${TAG}_arith.err | tail -1 > gpurun_out/bench_${TAG}_arith.json
python tools/bench_line.py < gpurun_out/bench_${TAG}_arith.json
for wl in hash grumpkin ecdsa arith_pedersen mixed; do
  case $wl in hash) K="--steps 200 --warmup 20";; grumpkin|ecdsa) K="--steps 50 --warmup 5";; *) K="";; esac  # (sub-millisecond steps: time tens of milliseconds of them, not the clock ramp)
  timeout 900 python bench.py --workload $wl $K 2> gpurun_out/bench_${TAG}_$wl.err | tail -1 > gpurun_out/bench_${TAG}_$wl.json
  python tools/bench_line.py < gpurun_out/bench_${TAG}_$wl.json
done
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
for m in "4096 3 4 plain" "4096 3 4 fold" "8192 3 4 reuse"; do ACVM_T_B2S=1 timeout 900 python tools/t_config5.py 1000000 $m 2>&1 | tail -1; done > gpurun_out/config5_${TAG}_1m.txt
timeout 900 python tools/t_node.py 1000000 32768 8192 1 reuse 2>&1 | tail -1 >> gpurun_out/config5_${TAG}_1m.txt
cut -c1-700 gpurun_out/config5_${TAG}_1m.txt
timeout 1200 bash tools/gpu_c5_counters.sh ${TAG} 1000000 8192 reuse > gpurun_out/config5_${TAG}_counters.log 2>&1
tail -22 gpurun_out/config5_${TAG}_counters.log
timeout 900 python tools/t_node.py 1000000 131072 8192 1 reuse 256 1 2>&1 | tail -1 > gpurun_out/config5_${TAG}_2p17.json
cut -c1-600 gpurun_out/config5_${TAG}_2p17.json
timeout 600 python tools/t_node_create.py > gpurun_out/node_create_${TAG}.txt 2>&1; cat gpurun_out/node_create_${TAG}.txt
timeout 600 bash tools/gpu_trace_arith.sh ${TAG} > gpurun_out/arith_${TAG}_timeline.txt 2>&1
tail -12 gpurun_out/arith_${TAG}_timeline.txt
ACVM_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --steps 5 --warmup 2 2> gpurun_out/bench_${TAG}_n2.err | tail -1 > gpurun_out/bench_${TAG}_n2_shared_gpu.json
python tools/bench_line.py < gpurun_out/bench_${TAG}_n2_shared_gpu.json
