"""Per-kernel totals of a rocprofv3 result database (rocpd / sqlite), largest first.

    python tools/prof_summary.py gpurun_out/prof/r_results.db [rows]
"""
import sqlite3
import sys


def summarise(path, limit=14):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = f"select s.kernel_name, count(*), sum(d.end - d.start) from {kd} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    print(f"total kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
    for name, calls, ns in rows[:limit]:
        print(f"{ns / 1e6:10.2f} ms  {calls:6d}  {name[:110]}")


if __name__ == "__main__":
    summarise(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
