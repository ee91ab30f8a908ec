"""Text summary of a ``rocprofv3 --kernel-trace --marker-trace`` database of a training step under the
reducer: the phase ranges (forward / backward / optimizer, emitted by ``bench.py --trace``), the
reducer's per-bucket launch ranges, and where the collectives' kernels ran -- which hardware queue,
and how much of their run time fell inside the window in which backward kernels were still running
on the compute queue (the overlap the design is after; reference behaviour: torch DDP's overlapped
bucket all-reduce behind ``DistributedDataParallel``, reference distributed.py:112-115).

    python -m distributed_pytorch_amd.bench.trace_summary <results.db> [--out summary.txt]
"""
from __future__ import annotations

import argparse
import json
import sqlite3
import sys


def summarize(db_path: str) -> str:
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    regions = []
    for name, start, end, tid, ext in cur.execute("select name, start, end, tid, extdata from regions order by start"):
        try:
            msg = json.loads(ext).get("message", name)
        except Exception:
            msg = name
        regions.append((msg, start, end, tid))
    kernels = cur.execute("select name, queue_id, stream_id, start, end, tid from kernels order by start").fetchall()
    if not regions or not kernels:
        return "no marker ranges or no kernels in the trace\n"
    fwd = [r for r in regions if r[0] == "forward"]
    opt = [r for r in regions if r[0] == "optimizer"]
    lines = []
    if fwd and opt:
        f, o = fwd[-1], opt[-1]
        lines.append("last traced step, host side (ms since its forward range began):")
        for msg, s, e, tid in regions:
            if s >= f[1] and e <= o[2] + 1:
                lines.append(f"  {msg:<34s} {(s - f[1]) / 1e6:9.3f} .. {(e - f[1]) / 1e6:9.3f}  ({(e - s) / 1e6:8.3f} ms, host thread {tid})")
        lines.append("")
    is_comm = lambda n: any(s in n for s in ("nccl", "Nccl", "rccl", "oneRank", "OneRank"))
    comm = [k for k in kernels if is_comm(k[0])]
    compute = [k for k in kernels if not is_comm(k[0])]
    counts = {}
    for k in compute:
        counts[(k[1], k[2])] = counts.get((k[1], k[2]), 0) + 1
    cq = max(counts.items(), key=lambda kv: kv[1])[0]
    main = [k for k in compute if (k[1], k[2]) == cq]
    # Backward kernels are launched by the autograd engine's thread -- the thread that also emits the
    # reducer's bucket ranges; forward and optimizer kernels by the main thread.  The last maximal run of
    # autograd-thread kernels on the compute queue is the device-side backward window of the last step.
    bucket_tids = {r[3] for r in regions if r[0].startswith("bucket ") or r[0] == "reducer finalize"}
    if not bucket_tids:
        return "\n".join(lines + ["no reducer ranges in the trace (was the model wrapped and tracing enabled?)"]) + "\n"
    groups, cur_g = [], None
    for k in main:
        if k[5] in bucket_tids:
            if cur_g is None:
                cur_g = [k[3], k[4], 1]
            else:
                cur_g[1], cur_g[2] = k[4], cur_g[2] + 1
        elif cur_g is not None:
            groups.append(cur_g)
            cur_g = None
    if cur_g is not None:
        groups.append(cur_g)
    groups = [g for g in groups if g[2] > 20]
    g0, g1, gn = groups[-1]
    rel = lambda t: (t - g0) / 1e6
    lines.append(f"device side: compute kernels on hardware queue {cq[0]} (stream {cq[1]}); backward window of the last step: "
                 f"{gn} kernels, {rel(g1):.3f} ms")
    step_comm = [k for k in comm if k[3] >= g0]
    queues = sorted({(k[1], k[2]) for k in step_comm})
    lines.append("collective kernels run on " + ", ".join(f"hardware queue {q} (stream {s})" for q, s in queues)
                 + " (ms since the first backward kernel):")
    hidden = total = 0.0
    for name, q, s, st, en, _tid in step_comm:
        dur = (en - st) / 1e6
        inside = max(0.0, min(en, g1) - st) / 1e6
        total += dur
        hidden += min(inside, dur)
        lines.append(f"  queue {q}  {rel(st):9.3f} .. {rel(en):9.3f}  {dur:7.3f} ms  {'inside ' if en <= g1 else ('partly ' if st < g1 else 'after  ')} {name[:60]}")
    lines.append("")
    lines.append(f"collective kernel time {total:.3f} ms; {hidden:.3f} ms ({100 * hidden / max(total, 1e-9):.0f} %) of it ran while backward "
                 f"kernels were still executing on the compute queue")
    return "\n".join(lines) + "\n"


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("db")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    text = summarize(a.db)
    sys.stdout.write(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
