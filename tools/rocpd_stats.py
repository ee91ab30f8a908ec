"""Per-kernel time summary from a rocprofv3 SQLite result (``*_results.db``), written as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv] [--top N]
"""
import csv
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = 40
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
        args = [a for a in args if a != str(top)]
    db = sqlite3.connect(args[0])
    rows = db.execute("select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels "
                      "group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = [("name", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent")]
    for name, n, tot, mn, mx in rows:
        out.append((name, n, round(tot / 1e6, 3), round(tot / n / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2), round(100.0 * tot / total, 2)))
    if len(args) > 1:
        with open(args[1], "w", newline="") as f:
            csv.writer(f).writerows(out)
    for r in out[: top + 1]:
        print(f"{str(r[0])[:100]:100s} {r[1]:>7} {r[2]:>10} {r[3]:>9} {r[6]:>6}")
    print("total kernel ms", round(total / 1e6, 2))


if __name__ == "__main__":
    main()
