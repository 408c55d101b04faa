#!/usr/bin/env python3
"""Per-kernel statistics (the `--stats` table) out of a rocprofv3 rocpd sqlite file.
Usage: rocpd_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), "
        "min(d.end - d.start), max(d.end - d.start), max(s.arch_vgpr_count), max(s.sgpr_count), "
        "max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["%-92s %6s %12s %11s %11s %11s %6s %5s %5s %7s" % (
        "kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "sgpr", "lds")]
    for name, calls, tot, avg, mn, mx, vg, sg, lds in rows:
        short = name if len(name) <= 92 else name[:89] + "..."
        out.append("%-92s %6d %12d %11.0f %11d %11d %6.2f %5s %5s %7s" % (
            short, calls, tot, avg, mn, mx, 100.0 * tot / total, vg, sg, lds))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
