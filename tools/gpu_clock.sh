#!/bin/bash
# tools/gpu_clock.sh -- shader / memory clocks and power while the default bench runs (rocm-smi polled every 0.2 s)
python bench.py --no-cpu-baseline --no-end-to-end --no-digest --steps 40 > gpurun_out/clock_bench.json 2>/dev/null &
PID=$!
sleep 8
for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 0.25; done > gpurun_out/clock_samples.txt
wait $PID
echo idle; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '
cat gpurun_out/clock_samples.txt | cut -c1-300
python tools/bench_line.py < gpurun_out/clock_bench.json
