"""Run under `python -m torch.distributed.run --nproc-per-node N ...`: creates the RCCL process group through
the launcher's own store (the rendezvous path the multi-GPU benchmark takes) and checks one all-reduce."""
import os
import torch
import distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(rank, world, "rccl")
dev = dist.get_device()
t = torch.full((1 << 20,), float(rank + 1), device=dev)
dist.all_reduce(t, "sum")
torch.cuda.synchronize()
expect = world * (world + 1) / 2
assert torch.all(t == expect), (t[:4], expect)
model = dist.prepare_ddp_model(torch.nn.Linear(64, 64).to(dev), device_ids=[dev.index])
model(torch.randn(8, 64, device=dev)).sum().backward()
torch.cuda.synchronize()
dist.barrier()
dist.print_primary(f"rccl over the launcher store ok: world {world}, device {dev}, all_reduce = {t[0].item()}")
dist.cleanup()
