#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = f"select s.{name_col}, d.end - d.start, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {disp} d join {sym} s on d.kernel_id = s.id"
    rows = db.execute(q).fetchall()
    agg = {}
    for name, dur, gx, gy, wx in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# source: {path}")
    print(f"# columns in dispatch table: {cols}")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:70]:70s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:10.2f} {a[3] / 1e3:10.2f} {100 * a[1] / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
