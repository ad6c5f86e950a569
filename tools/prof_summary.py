#!/usr/bin/env python3
"""Summarise one tools/gpu_profile.sh output directory into the text committed under profiles/:

    python tools/prof_summary.py gpurun_out/prof_r01c > profiles/r01c_bench_profile.txt

* per-kernel stats from `rocprofv3 --kernel-trace --stats` (trace/trace_kernel_stats.csv)
* per-kernel HBM traffic from the two separate PMC passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch),
  corrected as MI355X_MICROARCH.md §HBM prescribes: on gfx950 FETCH_SIZE reports half the bytes of a
  16 B/lane coalesced stream, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.
  The correction is re-derived here from the calibration run (tools/microbench copy: 4 GiB read + 4 GiB
  written per launch) and printed.
"""
import csv
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0].replace("acvm::", "").replace("void ", "")
    for cls, nice in (("LightOp", "light"), ("HashOp", "hash"), ("GrumpkinOp", "grumpkin"), ("BrilligOp", "brillig")):
        if name.startswith("record_level_kernel<" + cls):
            return nice + "_level_kernel"
        if name.startswith("record_exact_kernel<" + cls):
            return nice + "_exact_kernel"
    return name


def read_pmc(path, counter):
    per = defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):
        return per
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = per[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return per


def main(d):
    # the command that was profiled, as the script that ran it wrote it down (command.txt in the run directory): never a guess
    try:
        cmd = open(os.path.join(d, "command.txt")).read().strip()
    except OSError:
        cmd = "(command not recorded in this run directory)"
    have_pmc = os.path.exists(os.path.join(d, "pmc_fetch", "pmc_counter_collection.csv"))
    print(f"# source: {d}: rocprofv3 --kernel-trace --stats" + ("; --pmc FETCH_SIZE; --pmc WRITE_SIZE: three separate runs" if have_pmc else "") + " of")
    print(f"#         `{cmd}`")
    stats = os.path.join(d, "trace", "trace_kernel_stats.csv")
    print("\n== kernel stats (rocprofv3 --kernel-trace --stats)")
    print(f"{'kernel':36s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    with open(stats) as f:
        for row in csv.DictReader(f):
            print(f"{short(row['Name'])[:36]:36s} {int(row['Calls']):6d} {float(row['TotalDurationNs']) / 1e6:10.3f} "
                  f"{float(row['AverageNs']) / 1e3:10.2f} {float(row['MinNs']) / 1e3:9.2f} {float(row['MaxNs']) / 1e3:9.2f} "
                  f"{float(row['Percentage']):6.2f}")
    # the --stats average mixes the warm-up solve (first touch of a fresh witness table) with the timed ones: per solve, from the trace
    trace = os.path.join(d, "trace", "trace_kernel_trace.csv")
    if os.path.exists(trace) and "--inner" in cmd:  # (the four-solve layout below is the inner command's)
        per = defaultdict(list)
        with open(trace) as f:
            for row in csv.DictReader(f):
                per[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        print("\n== average launch duration per solve (same kernel trace; solve 0 = warm-up, 1..3 = timed steps; bench.py's HIP events bracket the last one)")
        for k, v in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
            if len(v) < 8 or len(v) % 4:
                continue
            v.sort()
            n = len(v) // 4
            cells = []
            for i in range(4):
                seg = v[i * n:(i + 1) * n]
                cells.append(f"{sum(e - s for s, e in seg) / n / 1e3:9.2f} us x {n}")
            print(f"{k[:36]:36s} " + "   ".join(cells))
    if not have_pmc:  # a trace-only run (the driver's command under rocprofv3): no counter sections to print
        return
    cal_f = read_pmc(os.path.join(d, "cal_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    cal_w = read_pmc(os.path.join(d, "cal_write", "pmc_counter_collection.csv"), "WRITE_SIZE")
    fcorr, wcorr = 2.0, 1.0
    print("\n== PMC calibration (tools/microbench copy: 4 GiB read + 4 GiB written per launch, uint4 per lane)")
    if "copy_kernel" in cal_f and cal_f["copy_kernel"][0]:
        n, s = cal_f["copy_kernel"]
        kib = s / n
        fcorr = (4 << 20) / kib
        print(f"FETCH_SIZE per launch = {kib:.1f} KiB for 4194304 KiB read  -> read correction x{fcorr:.4f}")
    if "copy_kernel" in cal_w and cal_w["copy_kernel"][0]:
        n, s = cal_w["copy_kernel"]
        kib = s / n
        wcorr = (4 << 20) / kib
        print(f"WRITE_SIZE per launch = {kib:.1f} KiB for 4194304 KiB written -> write correction x{wcorr:.4f}")
    fetch = read_pmc(os.path.join(d, "pmc_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    write = read_pmc(os.path.join(d, "pmc_write", "pmc_counter_collection.csv"), "WRITE_SIZE")
    print("\n== HBM traffic per kernel (separate --pmc passes; bytes = KiB x 1024 x correction)")
    print(f"{'kernel':36s} {'launches':>8s} {'read_GB/launch':>15s} {'write_GB/launch':>16s} {'total_GB/launch':>16s} {'total_GB_all':>13s}")
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch[k][1] + write[k][1])):
        n = max(fetch[k][0], write[k][0]) or 1
        rd = fetch[k][1] * 1024 * fcorr / 1e9
        wr = write[k][1] * 1024 * wcorr / 1e9
        print(f"{k[:36]:36s} {n:8d} {rd / n:15.4f} {wr / n:16.4f} {(rd + wr) / n:16.4f} {rd + wr:13.3f}")


def traffic_json(d, workload, alg=None):
    """Per-kernel measured HBM bytes per launch for profiles/traffic.json (read by bench.py)."""
    cal_f = read_pmc(os.path.join(d, "cal_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    cal_w = read_pmc(os.path.join(d, "cal_write", "pmc_counter_collection.csv"), "WRITE_SIZE")
    fcorr = (4 << 20) / (cal_f["copy_kernel"][1] / cal_f["copy_kernel"][0]) if cal_f.get("copy_kernel", [0])[0] else 2.0
    wcorr = (4 << 20) / (cal_w["copy_kernel"][1] / cal_w["copy_kernel"][0]) if cal_w.get("copy_kernel", [0])[0] else 1.0
    fetch = read_pmc(os.path.join(d, "pmc_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    write = read_pmc(os.path.join(d, "pmc_write", "pmc_counter_collection.csv"), "WRITE_SIZE")
    out = {}
    for k in set(fetch) | set(write):
        n = max(fetch[k][0], write[k][0]) or 1
        out[k] = {"bytes_per_launch": (fetch[k][1] * fcorr + write[k][1] * wcorr) * 1024 / n, "launches_profiled": n,
                  "read_correction": fcorr, "write_correction": wcorr, "source": f"{os.path.basename(d)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    return {workload: out}


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--json":
        import json
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
        try:
            cur = json.load(open(path))
        except (OSError, ValueError):
            cur = {}
        cur.update(traffic_json(sys.argv[1], sys.argv[3]))
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
        print("updated", path)
    else:
        main(sys.argv[1])
