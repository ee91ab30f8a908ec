"""Bounded host waits of the RCCL communicator (one GPU): `ok` runs bucket all-reduces and barriers under a
generous DIST_WATCHDOG_S; `stall` queues ~0.3 s of GEMMs in front of a barrier with a 50 ms limit, which
must surface as a RuntimeError naming rank and operation."""
import os
import sys

mode = sys.argv[1]
os.environ["DIST_WATCHDOG_S"] = "10" if mode == "ok" else "0.05"
import torch  # noqa: E402

from distributed_pytorch_amd.parallel import process_group as pg  # noqa: E402
from distributed_pytorch_amd.parallel.store import Store  # noqa: E402

pg.init_process_group(0, 1, "rccl", store=Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1"))
comm = pg.get_comm()
assert comm.watchdog_seconds() > 0
t = torch.ones(1 << 20, device="cuda")
if mode == "ok":
    for i in range(20):
        comm.all_reduce_async(t, 0, i % 2)  # slot reuse exercises the bounded event wait
    comm.wait_all_slots()
    comm.synchronize_comm_stream()
    comm.barrier()
    torch.cuda.synchronize()
    print(f"watchdog ok: async {comm.num_async_ops()}", flush=True)
else:
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for _ in range(400):
        a @ a
    try:
        comm.barrier()
        print("watchdog did not fire", flush=True)
    except RuntimeError as e:
        print(f"raised: {e}", flush=True)
    torch.cuda.synchronize()
pg.destroy_process_group()
