#!/bin/bash
# tools/gpu_clock_cmd.sh SECONDS_BEFORE N_SAMPLES CMD... -- shader clock and socket power while CMD runs (rocm-smi polled every 0.25 s, first sample after SECONDS_BEFORE)
WAIT=$1; N=$2; shift 2
"$@" > /tmp/clock_cmd.log 2>&1 &
PID=$!
sleep "$WAIT"
for i in $(seq 1 "$N"); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*(\([0-9]*Mhz\)).*/\1/' -e 's/.*(W): //' | tr '\n' ' '; echo; sleep 0.25; done
wait $PID
tail -2 /tmp/clock_cmd.log | cut -c1-160
