#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table
(the same columns `--stats` prints: calls, total / average / min / max duration, share).
    python tools/rocpd_summary.py gpurun_out/prof/orb_results.db > profiles/r01_orb_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    stats = {}
    for name, s, e in rows:
        st = stats.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    tot = sum(v[1] for v in stats.values()) or 1
    print("# rocprofv3 --kernel-trace summary of %s" % path)
    print("%-64s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 64 else name[:61] + "..."
        print("%-64s %8d %14d %12.0f %12d %12d %6.2f%%" % (short, v[0], v[1], v[1] / v[0], v[2], v[3], 100.0 * v[1] / tot))


if __name__ == "__main__":
    main(sys.argv[1])
