"""All-reduce bus-bandwidth sweep over message sizes, on the native RCCL communicator.

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        -m distributed_pytorch_amd.bench.allreduce_bw [--min-kb 64] [--max-mb 1024] [--dtype bf16|fp32]

Bucket sizes of the gradient reducer are meant to be read off this curve: on a fully connected
xGMI mesh each link carries bucket/N per step, so the knee sits at larger messages than on a
switched fabric.  Bus bandwidth = bytes / time * 2(N-1)/N; rank 0 prints one JSON line per size.
With a single rank the collective is a no-op and only the launch + fence latency is measured.
"""
from __future__ import annotations

import argparse
import json
import os

import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-kb", type=int, default=64)
    ap.add_argument("--max-mb", type=int, default=1024)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--comm-stream", action="store_true", help="use the reducer's path: dedicated stream + event fences")
    a = ap.parse_args(argv)

    import distributed as dist
    from distributed_pytorch_amd.parallel import process_group as pg
    from distributed_pytorch_amd.utils.timer import allreduce_bus_bandwidth_gbps

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("MASTER_PORT"):
        dist.init_process_group(rank, max(world, 1))
    dev = dist.get_device()
    if dev.type != "cuda":
        raise SystemExit("allreduce_bw needs GPUs (the RCCL data plane)")
    dtype = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    esz = torch.empty((), dtype=dtype).element_size()
    comm = pg.get_comm()
    size = a.min_kb << 10
    while size <= (a.max_mb << 20):
        t = torch.ones(size // esz, dtype=dtype, device=dev)

        def once(slot=0):
            if comm is None:
                return
            if a.comm_stream:
                comm.all_reduce_async(t, pg.OP_AVG, slot)
                comm.wait_slot(slot)
            else:
                comm.all_reduce(t, pg.OP_AVG)

        for _ in range(3):
            once()
        dist.barrier()
        torch.cuda.synchronize(dev)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(a.iters):
            once(i % 4)
        e.record()
        e.synchronize()
        sec = s.elapsed_time(e) / a.iters * 1e-3
        tt = torch.tensor([sec], dtype=torch.float64, device=dev)
        pg.all_reduce(tt, "max")
        sec = float(tt.item())
        if dist.is_primary():
            print(json.dumps({"bytes": size, "dtype": a.dtype, "world": max(world, 1), "time_us": round(sec * 1e6, 2),
                              "alg_bw_GBps": round(size / sec / 1e9, 2),
                              "bus_bw_GBps": round(allreduce_bus_bandwidth_gbps(size, sec, max(world, 1)), 2),
                              "path": "comm-stream" if a.comm_stream else "current-stream"}), flush=True)
        del t
        size *= 2
    dist.cleanup()


if __name__ == "__main__":
    main()
