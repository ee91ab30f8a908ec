"""Data-parallel GPT-2 training on synthetic token sequences, written against the ``distributed`` facade.

Where the reference's example (reference min_DDP.py:53-130) trains a two-layer toy model for two epochs,
this script shows the same skeleton -- launch, process group, sampler, DDP wrapper, primary-only printing,
metric reduction -- on the transformer the benchmark uses, plus the pieces a real run needs: a pinned
staging ring that overlaps host-to-device copies with compute, bf16 parameters with fp32 master weights
in the fused optimizer, checkpoint save / resume, a replica-divergence check and JSONL logging.

    HIP_VISIBLE_DEVICES=0,1,2,3 python examples/train_gpt2.py --steps 200 --ckpt out/gpt2.pt
    DIST_WORLD_SIZE=2 python examples/train_gpt2.py --n-layer 2 --n-embd 64 --n-head 2 --seq-len 32 --steps 6   # CPU, gloo
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
from torch.utils.data import DataLoader, Dataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import distributed as dist  # noqa: E402
from distributed_pytorch_amd.models import GPT2, GPT2Config  # noqa: E402
from distributed_pytorch_amd.ops import FusedAdamW  # noqa: E402
from distributed_pytorch_amd.parallel import ShardedOptimizer  # noqa: E402
from distributed_pytorch_amd.utils.checkpoint import load_checkpoint, save_checkpoint  # noqa: E402
from distributed_pytorch_amd.utils.data import DevicePrefetcher  # noqa: E402
from distributed_pytorch_amd.utils.debug import assert_in_sync  # noqa: E402
from distributed_pytorch_amd.utils.logging import JsonlLogger  # noqa: E402
from distributed_pytorch_amd.utils.timer import StepTimer  # noqa: E402


class SyntheticTokens(Dataset):
    """Deterministic pseudo-text: every sample is a noisy arithmetic progression over the vocabulary, so the
    next token is predictable from the previous two and the loss has somewhere to go."""

    def __init__(self, n: int, seq_len: int, vocab: int, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        start = torch.randint(0, vocab, (n, 1), generator=g)
        stride = torch.randint(1, 17, (n, 1), generator=g)
        noise = (torch.rand(n, seq_len + 1, generator=g) < 0.02).long() * torch.randint(0, vocab, (n, seq_len + 1), generator=g)
        self.tokens = (start + stride * torch.arange(seq_len + 1) + noise) % vocab

    def __len__(self):
        return self.tokens.shape[0]

    def __getitem__(self, i):
        t = self.tokens[i]
        return t[:-1], t[1:]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--n-layer", type=int, default=12)
    p.add_argument("--n-head", type=int, default=12)
    p.add_argument("--n-embd", type=int, default=768)
    p.add_argument("--vocab-size", type=int, default=50257)
    p.add_argument("--seq-len", type=int, default=1024)
    p.add_argument("--batch-size", type=int, default=16, help="per process")
    p.add_argument("--data-size", type=int, default=4096)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--lr", type=float, default=3e-4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--ckpt", default=None, help="checkpoint path (written at the end and every --save-every steps)")
    p.add_argument("--save-every", type=int, default=0)
    p.add_argument("--resume", action="store_true", help="continue from --ckpt if it exists")
    p.add_argument("--log", default=None, help="JSONL log path (default: stdout of the primary rank)")
    p.add_argument("--log-every", type=int, default=10)
    p.add_argument("--shard-optimizer", action="store_true",
                   help="keep master weights and moments only for the parameters this rank owns (1/N of the optimizer state per GPU)")
    return p


def main_worker(gpu, world_size, argv=None):
    args = build_parser().parse_args(argv)
    if world_size > 1:
        dist.init_process_group(gpu, world_size)
    device = dist.get_device()
    on_gpu = device.type == "cuda"
    torch.manual_seed(args.seed)  # same initial weights everywhere (DDP broadcasts rank 0's anyway)

    cfg = GPT2Config(block_size=args.seq_len, vocab_size=args.vocab_size, n_layer=args.n_layer, n_head=args.n_head, n_embd=args.n_embd)
    model = GPT2(cfg).to(device=device, dtype=torch.bfloat16 if on_gpu else torch.float32)
    model = dist.prepare_ddp_model(model, device_ids=[gpu])
    make_opt = ShardedOptimizer if args.shard_optimizer else FusedAdamW
    opt = make_opt(model.parameters(), lr=args.lr, weight_decay=0.1, master_weights=on_gpu, max_grad_norm=1.0)

    step = 0
    if args.resume and args.ckpt and os.path.exists(args.ckpt):
        step = load_checkpoint(args.ckpt, model, opt)["step"]
        dist.print_primary(f"resumed from {args.ckpt} at step {step}")

    data = SyntheticTokens(args.data_size, args.seq_len, args.vocab_size, seed=args.seed)
    sampler = dist.data_sampler(data, world_size > 1, shuffle=True)
    loader = DataLoader(data, batch_size=args.batch_size, sampler=sampler, shuffle=sampler is None, drop_last=True,
                        pin_memory=on_gpu, num_workers=0)
    log = JsonlLogger(args.log)
    timer = StepTimer(device)
    loss_sum = torch.zeros((), device=device)
    epoch = step // max(len(loader), 1)
    timer.start()
    window = 0
    while step < args.steps:
        if sampler is not None:
            sampler.set_epoch(epoch)
        for idx, tgt in DevicePrefetcher(loader, device):
            loss = model(idx, tgt)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            loss_sum += loss.detach().float()
            step += 1
            window += 1
            if step % args.log_every == 0 or step == args.steps:
                mean = loss_sum / window  # one host sync per logging window, none in between
                dist.all_reduce(mean, "avg")
                ms = timer.stop(window)
                log.log(step=step, epoch=epoch, loss=round(mean.item(), 4), ms_per_step=round(ms, 2),
                        samples_per_s=round(args.batch_size * world_size / (ms / 1e3), 1))
                loss_sum.zero_()
                window = 0
                assert_in_sync(model, step)  # no-op unless DIST_DEBUG_SYNC=1
                timer.start()
            if args.ckpt and args.save_every and step % args.save_every == 0:
                save_checkpoint(args.ckpt, model, opt, step)
            if step >= args.steps:
                break
        epoch += 1
    if args.ckpt:
        save_checkpoint(args.ckpt, model, opt, step)
        dist.print_primary(f"saved {args.ckpt} at step {step}")
    dist.cleanup()


if __name__ == "__main__":
    dist.launch(main_worker, sys.argv[1:])
