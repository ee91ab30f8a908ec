"""Experiment: the GPT-2 training step as N micro-batches, issued one after another on one stream
("serial") or each on a HIP stream of its own ("streams"), so that the bandwidth-bound kernels of one
micro-batch (layer norm, loss) can share the chip with the matrix products of another.

    PYTHONPATH=. python tools/exp_two_stream.py <per-GPU batch> <micro-batches> <serial|streams> [steps]

Prints ms per step and the loss after the last step (the two modes must agree on it).
"""
import sys
import time

import torch

from distributed_pytorch_amd.models import GPT2, GPT2Config
from distributed_pytorch_amd.ops import FusedAdamW


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    nmb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    mode = sys.argv[3] if len(sys.argv) > 3 else "streams"
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = GPT2Config.small(block_size=1024)
    model = GPT2(cfg).to(dev, torch.bfloat16)
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01, master_weights=True)
    g = torch.Generator().manual_seed(1234)
    idx = torch.randint(0, cfg.vocab_size, (B, 1024), generator=g).to(dev)
    tgt = torch.randint(0, cfg.vocab_size, (B, 1024), generator=g).to(dev)
    chunks = list(zip(idx.chunk(nmb), tgt.chunk(nmb)))
    streams = [torch.cuda.Stream() for _ in range(nmb)]

    def step():
        opt.zero_grad(set_to_none=True)
        cur = torch.cuda.current_stream()
        if mode == "serial" or nmb == 1:
            total = 0.0
            for i, t in chunks:
                loss = model(i, t) / nmb
                loss.backward()
                total = total + loss.detach()
        else:
            losses = []
            for s, (i, t) in zip(streams, chunks):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    losses.append(model(i, t) / nmb)
            for s, l in zip(streams, losses):
                with torch.cuda.stream(s):
                    l.backward()
            for s in streams:
                cur.wait_stream(s)
            total = sum(l.detach() for l in losses)
        opt.step()
        return total

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"batch {B} micro {nmb} mode {mode}: {ms:.2f} ms/step, {B / ms * 1e3:.1f} samples/s, loss {float(last):.5f}, "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
