"""rocprofv3 driver and summariser for the benchmark (``python -m distributed_pytorch_amd.bench.profile``).

    profile.py stats  [--out DIR] -- <program args...>     one --kernel-trace --stats run, per-kernel CSV
    profile.py pmc    [--out DIR] [--filter NAME] -- <program args...>
                                                            counter passes (SQ set A, SQ set B, cache), one
                                                            process each, summarised per kernel
    profile.py summarize FILE [--filter NAME]               re-summarise an existing .db / counter CSV

The program after ``--`` is run as ``python <args>`` directly under rocprofv3 (no shell, no env
wrapper: the profiler initialises the GPU before the program starts, so an exec hop is not
allowed).  Counter passes are kept within the per-block hardware limits (8 SQ, 4 TCC, 4 TCP
counters per run) and are never combined with API traces.  Timings of profiled runs are not
comparable with un-profiled ones; use them for ratios between kernels only.
"""
from __future__ import annotations

import argparse
import csv
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict
from typing import Dict, List

PMC_PASSES: Dict[str, List[str]] = {
    "issue": ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
              "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_VALU"],
    "lds": ["GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS",
            "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VMEM"],
    "memory": ["GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"],
}


def kernel_stats_from_db(path: str):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels "
                      "group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = [("name", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent")]
    for name, n, tot, mn, mx in rows:
        out.append((name, n, round(tot / 1e6, 3), round(tot / n / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2), round(100.0 * tot / total, 2)))
    return out, total / 1e6


def pmc_summary(path: str, name_filter: str = ""):
    acc = defaultdict(lambda: defaultdict(list))
    times = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if name_filter and name_filter not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            times[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = []
    for k, d in acc.items():
        row = {"kernel": k, "dispatches": len(next(iter(d.values()))), "avg_us": round(sum(times[k]) / len(times[k]), 2)}
        for c, v in sorted(d.items()):
            row[c] = round(sum(v) / len(v), 1)
        gui = row.get("GRBM_GUI_ACTIVE")
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            row["mfma_busy_pct"] = round(100.0 * row["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), 1)
            row["clock_GHz"] = round(gui / 8 / (row["avg_us"] * 1e3), 3)
        out.append(row)
    out.sort(key=lambda r: -r["avg_us"] * r["dispatches"])
    return out


def _write_csv(path: str, rows):
    if not rows:
        return
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", newline="") as f:
        if isinstance(rows[0], dict):
            keys: List[str] = []
            for r in rows:
                for k in r:
                    if k not in keys:
                        keys.append(k)
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(rows)
        else:
            csv.writer(f).writerows(rows)


def _run(cmd: List[str], timeout: int) -> int:
    env = dict(os.environ)
    env.setdefault("TMPDIR", "/tmp")
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    prog: List[str] = []
    if "--" in argv:
        i = argv.index("--")
        argv, prog = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser(prog="profile")
    ap.add_argument("mode", choices=["stats", "pmc", "summarize"])
    ap.add_argument("file", nargs="?")
    ap.add_argument("--out", default="gpurun_out/profile")
    ap.add_argument("--filter", default="", help="substring of the kernel names to keep in counter summaries")
    ap.add_argument("--passes", default="issue,lds", help="comma list out of: " + ",".join(PMC_PASSES))
    ap.add_argument("--timeout", type=int, default=300)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args(argv)

    if a.mode == "summarize":
        if a.file.endswith(".db"):
            rows, total = kernel_stats_from_db(a.file)
            for r in rows[: a.top + 1]:
                print(f"{str(r[0])[:96]:96s} {r[1]:>7} {r[2]:>10} {r[3]:>9} {r[6]:>6}")
            print("total kernel ms", round(total, 2))
        else:
            for r in pmc_summary(a.file, a.filter)[: a.top]:
                print(r)
        return 0

    if not prog:
        ap.error("give the program to profile after '--' (it is run as: python <args>)")
    os.makedirs(a.out, exist_ok=True)
    if a.mode == "stats":
        rc = _run(["rocprofv3", "--kernel-trace", "--stats", "-d", a.out, "-o", "trace", "--", sys.executable] + prog, a.timeout)
        if rc != 0:
            return rc
        rows, total = kernel_stats_from_db(os.path.join(a.out, "trace_results.db"))
        _write_csv(os.path.join(a.out, "kernel_stats.csv"), rows)
        for r in rows[: a.top + 1]:
            print(f"{str(r[0])[:96]:96s} {r[1]:>7} {r[2]:>10} {r[3]:>9} {r[6]:>6}")
        print("total kernel ms", round(total, 2))
        return 0
    for name in a.passes.split(","):
        counters = PMC_PASSES[name]
        d = os.path.join(a.out, "pmc_" + name)
        rc = _run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "c", "--", sys.executable] + prog, a.timeout)
        if rc != 0:
            return rc  # a failed pass ends the session: nothing more is started on the GPU
        rows = pmc_summary(os.path.join(d, "c_counter_collection.csv"), a.filter)
        _write_csv(os.path.join(a.out, f"pmc_{name}.csv"), rows)
        for r in rows[: a.top]:
            print({k: v for k, v in r.items() if k != "kernel"}, r["kernel"][:80])
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
