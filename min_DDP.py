"""Minimal data-parallel training example on the MI355X-native ``distributed`` module.

Usage mirrors the reference example (reference README.md:107-119), with the HIP spelling of
the visibility variable:

    HIP_VISIBLE_DEVICES="2,3" python3 min_DDP.py        # one process per listed GPU
    HIP_VISIBLE_DEVICES="2"   python3 min_DDP.py        # single GPU, in-process
    DIST_WORLD_SIZE=2         python3 min_DDP.py        # two CPU processes over gloo

Same flags and per-iteration output as reference min_DDP.py (per-rank block, then the
global line on the primary rank), plus:
  --seed N       seed the model initialisation (the reference leaves it unseeded)
  --quiet        no per-iteration prints and no per-iteration host synchronisation; metrics
                 are accumulated on the device and reported once per epoch
  --no-native    use stock nn.Linear / CrossEntropyLoss / AdamW instead of the HIP kernels
  --backend B    rccl | gloo (default: rccl on GPU, gloo on CPU)
"""
import argparse
import time

import torch
import torch.nn as nn
from torch.utils.data import DataLoader

import distributed as dist
from distributed_pytorch_amd.models.dummy import DummyDataset, DummyModel
from distributed_pytorch_amd.ops import FusedAdamW, SoftmaxCrossEntropy
from distributed_pytorch_amd.utils.data import DevicePrefetcher


def build_parser():
    p = argparse.ArgumentParser(description="MI355X multi-GPU training example")
    for flag, default, text in (
        ("--epochs", 2, "Number of training epochs."),
        ("--batch-size", 8, "Batch size."),
        ("--n-classes", 4, "Number of classes for fake dataset."),
        ("--data-size", 32, "Size of fake dataset."),
        ("--hidden-dim", 32, "Hidden dimension."),
    ):
        p.add_argument(flag, default=default, type=int, metavar="N", help=text)
    p.add_argument("--seed", default=None, type=int, metavar="N", help="Seed for the model initialisation.")
    p.add_argument("--quiet", action="store_true", help="Per-epoch metrics only, no per-iteration host sync.")
    p.add_argument("--no-native", action="store_true", help="Stock torch layers, loss and optimizer.")
    p.add_argument("--backend", default=None, choices=["rccl", "nccl", "gloo"], help="Communication backend.")
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


class EpochMeter:
    """Device-resident running sums: [loss sum over iterations, correct, seen]."""

    def __init__(self, device):
        self.acc = torch.zeros(3, dtype=torch.float32, device=device)

    def update(self, loss, correct):
        self.acc[0] += loss.detach().float()
        self.acc[1] += correct.sum()
        self.acc[2] += correct.numel()

    def report(self, n_iters):
        total = dist.reduce(self.acc.clone())  # one 3-float collective per epoch
        l, c, n = total.tolist()
        return l / max(n_iters, 1), c, n


def train(model, loader, criterion, optimizer, args):
    model.train()
    device = dist.get_device()
    native_loss = isinstance(criterion, SoftmaxCrossEntropy)
    meter = EpochMeter(device) if args.quiet else None
    n_iters = 0

    # batches arrive on the device one step ahead (pinned staging + copy stream); CPU: pass-through
    for it, (x, y) in enumerate(DevicePrefetcher(loader, device)):

        y_hat = model(x)
        loss = criterion(y_hat, y)

        optimizer.zero_grad()
        loss.backward()
        optimizer.step()

        # the fused loss kernel already produced argmax == label; the stock path recomputes it
        correct = criterion.last_correct if native_loss else torch.argmax(y_hat, dim=1).eq(y)
        n = y.shape[0]
        n_iters += 1

        if args.quiet:
            meter.update(loss, correct)
            continue

        # this rank's own numbers
        print(f"Device: {x.device}"
              f"\n\tInput: \t{x.squeeze().to(torch.uint8)}"
              f"\n\tLabel: \t{y.squeeze()}"
              f"\n\tPred:  \t{torch.argmax(y_hat, -1)}"
              f"\n\tCorr.: \t{correct.to(torch.uint8)}"
              f"\n\tAcc:   \t{correct.sum() / n:.5f} ({correct.sum()}/{n})"
              f"\n\tLoss:  \t{loss.cpu().item():.5f}")

        # line the ranks up so the per-rank blocks above are not interleaved with the summary
        dist.wait_for_everyone()

        # combine over ranks: SUM of the losses to rank 0 (as the reference does), gather of the hit masks
        loss = dist.reduce(loss.detach().clone())       # SUM over ranks, valid on the primary
        correct = dist.gather(correct.detach())         # all ranks' flags, valid on the primary
        correct = torch.cat(correct, dim=0)
        acc = correct.sum() / correct.numel()

        # job-wide line, primary rank only
        dist.print_primary(f"Finish iteration {it}"
                           f" - acc: {acc.cpu().item():.4f} ({correct.sum()}/{correct.shape[0]})"
                           f" - loss: {loss.cpu().item():.4f}")

    if args.quiet:
        mean_loss_sum, c, n = meter.report(n_iters)
        dist.print_primary(f"Epoch summary - iterations: {n_iters} - acc: {c / max(n, 1):.4f} ({int(c)}/{int(n)})"
                           f" - loss (sum over ranks, mean over iterations): {mean_loss_sum:.4f}")


# ---------------------------------------------------------------- worker
def main_worker(gpu, world_size, argv=None):
    is_distributed = world_size > 1
    args = parse_args(argv)
    if is_distributed:
        dist.init_process_group(gpu, world_size, args.backend)

    for name, val in vars(args).items():
        dist.print_primary("{:<12}: {}".format(name, val))

    if args.seed is not None:
        torch.manual_seed(args.seed)
    device = dist.get_device()
    native = not args.no_native

    """ Data """
    dataset = DummyDataset(args.data_size, args.n_classes)
    sampler = dist.data_sampler(dataset, is_distributed, shuffle=False)
    loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=(sampler is None), sampler=sampler,
                        pin_memory=False)  # DevicePrefetcher stages through its own pinned ring

    # model
    model = DummyModel(in_dim=1, hidden_dim=args.hidden_dim, n_classes=args.n_classes, native=native)
    model.to(device)
    model = dist.prepare_ddp_model(model, device_ids=[gpu])

    # optimizer and loss
    if native:
        optimizer = FusedAdamW(model.parameters(), 0.0001)
        criterion = SoftmaxCrossEntropy()
    else:
        optimizer = torch.optim.AdamW(model.parameters(), 0.0001)
        criterion = nn.CrossEntropyLoss()

    # epochs
    print("Run epochs")
    t0 = time.perf_counter()
    for epoch in range(args.epochs):
        dist.print_primary(f"------- Epoch {epoch+1}")
        if is_distributed:
            sampler.set_epoch(epoch)
        train(model, loader, criterion, optimizer, args)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if args.quiet:
        dist.print_primary(f"Wall time: {time.perf_counter() - t0:.3f} s")

    # leave the job
    dist.cleanup()


if __name__ == "__main__":
    # start one process per GPU if several are visible, otherwise run main_worker in-process
    dist.launch(main_worker)
