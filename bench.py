#!/usr/bin/env python3
"""Training-throughput benchmark: samples/sec of the whole job on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W [--model gpt2|widemlp|resnet50|dummy]

Default workload: GPT-2 small, synthetic 1024-token sequences, bf16 parameters with fp32
master weights, data parallel over N ranks (one process per GPU, RCCL over xGMI).  For
N > 1 start it with ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`` (RANK / LOCAL_RANK /
WORLD_SIZE are read from the environment); started plainly with ``--gpus N`` it spawns the
ranks itself through ``distributed.launch``.

Every step is a full training step -- forward, loss, backward (with bucketed gradient
all-reduce when N > 1), optimizer update -- on device-resident synthetic data and
random-initialised weights.  W warm-up steps are not timed; exactly K steps are timed between
``barrier + torch.cuda.synchronize()`` on both sides; the slowest rank's time counts.  Rank 0
prints one JSON line; ``value`` is the aggregate samples/sec over all N GPUs (a sample is one
sequence / image / row).  Per-GPU work is fixed as N grows (weak scaling).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--model", default="gpt2", choices=["gpt2", "gpt2-medium", "widemlp", "resnet50", "dummy"])
    p.add_argument("--batch-size", type=int, default=None, help="per-GPU batch (default depends on the model)")
    p.add_argument("--seq-len", type=int, default=1024)
    p.add_argument("--dtype", default=None, choices=["bf16", "fp32"], help="compute dtype (default bf16; dummy: fp32)")
    p.add_argument("--linear-impl", default=os.environ.get("BENCH_LINEAR_IMPL", "native"), choices=["native", "library", "stock"],
                   help="GPT-2 GEMMs: hand-written MFMA kernels (default); 'library' sends them to the vendor GEMM for A/B runs")
    p.add_argument("--widemlp-gemm", default="native", choices=["native", "library"],
                   help="Wide MLP: forward / input-gradient GEMMs on the hand-written kernel (default) or the vendor library")
    p.add_argument("--attention-impl", default="auto", choices=["auto", "native", "sdpa"], help="GPT-2 attention kernel (ablation)")
    p.add_argument("--no-fused-mlp", action="store_true", help="GPT-2: separate c_fc / GELU / c_proj autograd nodes (ablation)")
    p.add_argument("--no-fused-qkv-attention", action="store_true", help="GPT-2: c_attn and attention as separate autograd nodes (ablation)")
    p.add_argument("--no-fused-loss", action="store_true", help="GPT-2: separate lm-head GEMM, loss forward and loss backward passes (ablation)")
    p.add_argument("--no-fuse-add-norm", action="store_true", help="GPT-2: separate residual adds and layer norms (ablation)")
    p.add_argument("--reduce-dtype", default=None, choices=["fp32", "bf16"], help="gradient all-reduce dtype")
    p.add_argument("--bucket-mb", type=float, default=None)
    p.add_argument("--shard-optimizer", action="store_true",
                   help="N > 1: gradients are reduce-scattered, every rank keeps optimizer state for and updates its 1/N range of every "
                        "bucket, parameters are all-gathered (same bytes on the links as the all-reduce)")
    p.add_argument("--no-busbw", action="store_true", help="skip the all-reduce bus-bandwidth measurement after the timed steps (N > 1)")
    p.add_argument("--backend", default=None, choices=["rccl", "gloo"])
    p.add_argument("--force-ddp", action="store_true",
                   help="N = 1: create the process group and wrap the model in the reducer anyway (one-rank RCCL all-reduces on the "
                        "communication stream), to profile the overlap machinery on a single GPU")
    p.add_argument("--emulate-ranks", type=int, default=0,
                   help="with --force-ddp on ONE GPU: replace the one-rank all-reduces by a kernel on the communication stream that holds "
                        "--emulate-wgs workgroups for the time an all-reduce over this many ranks would take at --emulate-busbw (a load "
                        "model for tuning bucket sizes; the result is labelled emulated and is not a multi-GPU measurement)")
    p.add_argument("--emulate-busbw", type=float, default=300.0, help="emulated all-reduce bus bandwidth, GB/s")
    p.add_argument("--emulate-wgs", type=int, default=32, help="workgroups (CUs) the emulated collective keeps resident")
    p.add_argument("--max-collective-mb", type=float, default=None, help="largest single all-reduce (bigger buckets go out as slices)")
    p.add_argument("--no-overlap-optimizer", action="store_true",
                   help="wait for every gradient all-reduce at the end of backward instead of bucket by bucket inside the optimizer step")
    p.add_argument("--trace", action="store_true", help="emit roctx ranges (forward / backward / optimizer) for rocprofv3 --marker-trace")
    p.add_argument("--stock", action="store_true", help="comparison run: stock torch ops and torch.optim.AdamW(fused=True)")
    p.add_argument("--graph", action="store_true", help="capture the training step into a hipGraph (default for --model dummy on one GPU, whose step is launch-bound)")
    p.add_argument("--no-graph", action="store_true", help="launch every kernel of the step eagerly")
    p.add_argument("--json-out", default=None, help="also write the JSON line to this file")
    return p.parse_args(argv)


# --------------------------------------------------------------------------- workloads
def build_workload(args, device):
    """Returns (model, step_inputs_fn, loss_fn, per_gpu_batch, info)."""
    import torch

    from distributed_pytorch_amd.models import GPT2, DummyModel, GPT2Config, WideMLP, resnet50
    from distributed_pytorch_amd.ops import SoftmaxCrossEntropy

    name = args.model
    g = torch.Generator(device="cpu").manual_seed(1234 + int(os.environ.get("RANK", "0")))
    if name in ("gpt2", "gpt2-medium"):
        dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
        # per-GPU batch sized for 288 GB of HBM: 384 sequences of GPT-2 small peak at 149 GB allocated / 185 GB
        # reserved and run 4.5 % more samples/s than 64 (longer contractions in the weight-gradient GEMMs,
        # fewer launches per sample; 256: -0.6 %, 512: +0.6 % but 246 GB reserved)
        # (without a GPU the run is a plumbing check only: two sequences keep it within minutes)
        B = args.batch_size or ((384 if name == "gpt2" else 64) if device.type == "cuda" else 2)
        T = args.seq_len
        cfg = (GPT2Config.small if name == "gpt2" else GPT2Config.medium)(
            block_size=max(T, 1024), linear_impl="stock" if args.stock else args.linear_impl, native_ops=not args.stock,
            attention_impl=args.attention_impl, fuse_add_norm=not args.no_fuse_add_norm,
            fused_lm_head_loss=not args.no_fused_loss, fused_mlp=not args.no_fused_mlp,
            fused_qkv_attention=not args.no_fused_qkv_attention)
        model = GPT2(cfg).to(device=device, dtype=dtype)
        idx = torch.randint(0, cfg.vocab_size, (B, T), generator=g).to(device)
        tgt = torch.randint(0, cfg.vocab_size, (B, T), generator=g).to(device)
        loss_fn = lambda m: m(idx, tgt)
        info = dict(model="gpt2-small" if name == "gpt2" else "gpt2-medium", seq_len=T, params=model.num_parameters(),
                    flops_per_sample=model.flops_per_token(T) * T, tokens_per_sample=T,
                    flops_per_sample_noncausal=model.flops_per_token(T, causal=False) * T)
        return model, loss_fn, B, dtype, info
    if name == "widemlp":
        dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
        B = args.batch_size or 8192
        model = WideMLP(8192, 8192, 8, 1024, linear_impl=args.widemlp_gemm).to(device=device, dtype=dtype)
        x = torch.randn(B, 8192, generator=g).to(device=device, dtype=dtype)
        y = torch.randint(0, 1024, (B,), generator=g).to(device)
        crit = SoftmaxCrossEntropy()
        loss_fn = lambda m: crit(m(x), y)
        info = dict(model="wide-mlp-8x8192", seq_len=None, params=sum(p.numel() for p in model.parameters()),
                    flops_per_sample=model.flops_per_sample(), tokens_per_sample=1)
        return model, loss_fn, B, dtype, info
    if name == "resnet50":
        dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
        B = args.batch_size or 256
        model = resnet50().to(device=device, dtype=dtype).to(memory_format=torch.channels_last)
        x = torch.randn(B, 3, 224, 224, generator=g).to(device=device, dtype=dtype).contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, 1000, (B,), generator=g).to(device)
        crit = SoftmaxCrossEntropy()
        loss_fn = lambda m: crit(m(x), y)
        info = dict(model="resnet50", seq_len=None, params=sum(p.numel() for p in model.parameters()),
                    flops_per_sample=3 * 4.09e9 * 2, tokens_per_sample=1)
        return model, loss_fn, B, dtype, info
    # dummy: the reference model at hidden_dim 4096 on the native K=1 / N=4 kernels
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B = args.batch_size or 65536
    model = DummyModel(1, 4096, 4, native=True).to(device=device, dtype=dtype)
    x = (torch.arange(B, dtype=torch.float32).unsqueeze(1) / B).to(device=device, dtype=dtype)
    y = torch.randint(0, 4, (B,), generator=g).to(device)
    crit = SoftmaxCrossEntropy()
    loss_fn = lambda m: crit(m(x), y)
    info = dict(model="min_DDP-DummyModel-hidden4096", seq_len=None, params=sum(p.numel() for p in model.parameters()),
                flops_per_sample=6.0 * (4096 + 4 * 4096), tokens_per_sample=1)
    return model, loss_fn, B, dtype, info


# --------------------------------------------------------------------------- one rank
def run_rank(rank: int, world_size: int, argv):
    args = parse_args(argv)
    hang_s = os.environ.get("DIST_HANG_DUMP_S", "")
    if hang_s not in ("", "0"):
        # a stuck run prints the Python stack of every thread (repeatedly) instead of sitting silent
        import faulthandler

        faulthandler.dump_traceback_later(float(hang_s), repeat=True, file=sys.stderr)
    import torch

    import distributed as dist
    from distributed_pytorch_amd import _ext
    from distributed_pytorch_amd.ops import FusedAdamW
    from distributed_pytorch_amd.parallel import process_group as pg

    use_gpu = torch.cuda.is_available()
    if use_gpu:
        _ext.C()  # the HIP extension is mandatory on a GPU box
    distributed_run = world_size > 1 or args.force_ddp
    if distributed_run:
        if world_size <= 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                os.environ["MASTER_PORT"] = str(dist.find_free_port())
        dist.init_process_group(rank, max(world_size, 1), args.backend)
    device = dist.get_device()
    if use_gpu:
        torch.cuda.set_device(device)
    torch.manual_seed(0)

    model, loss_fn, per_gpu_batch, dtype, info = build_workload(args, device)
    ddp_kwargs = {}
    if args.reduce_dtype:
        ddp_kwargs["reduce_dtype"] = args.reduce_dtype
    if args.bucket_mb:
        ddp_kwargs["bucket_cap_mb"] = args.bucket_mb
    if args.max_collective_mb:
        ddp_kwargs["max_collective_mb"] = args.max_collective_mb
    if args.shard_optimizer and distributed_run:
        # reduce-scatter -> range update -> all-gather on the wrapper's own buffers (parallel/zero.py, arena mode)
        ddp_kwargs["shard_gradients"] = True
    if args.emulate_ranks > 1:
        if not (args.force_ddp and world_size <= 1 and use_gpu):
            raise SystemExit("--emulate-ranks needs --force-ddp on one GPU")
        pg.get_comm().set_emulation(args.emulate_ranks, args.emulate_busbw, args.emulate_wgs)
    if args.force_ddp and world_size <= 1:
        from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

        model = DistributedDataParallel(model, **ddp_kwargs)
    else:
        model = dist.prepare_ddp_model(model, device_ids=[device.index if device.type == "cuda" else None], **ddp_kwargs)
    if args.stock:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01, fused=use_gpu)
    elif args.shard_optimizer:
        from distributed_pytorch_amd.parallel import ShardedOptimizer

        opt = ShardedOptimizer(model.parameters(), lr=1e-4, weight_decay=0.01, master_weights=(dtype != torch.float32))
    else:
        opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01, master_weights=(dtype != torch.float32),
                         overlap_grad_reduce=not args.no_overlap_optimizer)

    if args.trace:
        from distributed_pytorch_amd.utils.timer import range as trace_range
    else:
        import contextlib

        trace_range = lambda _name: contextlib.nullcontext()

    def train_step():
        with trace_range("forward"):
            loss = loss_fn(model)
        opt.zero_grad(set_to_none=True)
        with trace_range("backward"):
            loss.backward()
        with trace_range("optimizer"):
            opt.step()
        return loss

    def sync():
        dist.barrier()
        if use_gpu:
            torch.cuda.synchronize(device)

    step = train_step
    graph = None
    want_graph = (args.graph or args.model == "dummy") and not args.no_graph and not args.force_ddp
    if want_graph and use_gpu and world_size == 1:
        from distributed_pytorch_amd.utils.graph import GraphedStep

        graph = GraphedStep(model, opt, loss_fn)
        step = graph.step

    first_loss = None
    for i in range(args.warmup):
        l = step()
        if i == 0:
            first_loss = l.detach().clone() if l is not None else None  # a graph replay rewrites its output tensor in place
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    sync()
    elapsed = time.perf_counter() - t0

    on_rccl = distributed_run and pg.get_backend() == "rccl"
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if on_rccl else "cpu")
    if world_size > 1:
        pg.all_reduce(t, "max")
    elapsed = float(t.item())
    final_loss = float(last.detach().float().item()) if last is not None else float("nan")

    n = max(world_size, 1)
    ms = elapsed / args.steps * 1e3
    samples_per_sec = per_gpu_batch * n * args.steps / elapsed
    result = {
        "metric": "samples_per_sec",
        "value": round(samples_per_sec, 3),
        "unit": "samples/s",
        "n_gpus": n if use_gpu else 0,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16" if dtype == torch.bfloat16 else "fp32",
        "data": "synthetic",
        "config": {
            "model": info["model"],
            "global_batch": per_gpu_batch * n,
            "per_gpu_batch": per_gpu_batch,
            "seq_len": info["seq_len"],
            "parallelism": f"dp{n}",
            "params": info["params"],
            "optimizer": "torch.optim.AdamW(fused)" if args.stock else ("ShardedOptimizer(" if args.shard_optimizer else "") + "FusedAdamW"
                         + ("+fp32-master" if dtype != torch.float32 else "") + (")" if args.shard_optimizer else ""),
            "stock_torch_ops": bool(args.stock),
            "linear_impl": args.linear_impl,
            "grad_reduce_dtype": (args.reduce_dtype or ("param dtype" if n > 1 else None)),
            "comm": (pg.get_backend() if distributed_run else None),
            "hip_graph": bool(graph is not None),
        },
        "tokens_per_sec": round(samples_per_sec * info["tokens_per_sample"], 1),
        # FLOPs the kernels execute (causal attention counted as half the full score matrix)
        "model_tflops_per_gpu": round(info["flops_per_sample"] * per_gpu_batch / (ms * 1e-3) / 1e12, 2),
        "final_loss": round(final_loss, 5),
        "first_loss": round(float(first_loss.detach().float().item()), 5) if first_loss is not None else None,
    }
    if "flops_per_sample_noncausal" in info:  # the convention that charges every query the whole sequence
        result["model_tflops_per_gpu_full_attention_convention"] = round(
            info["flops_per_sample_noncausal"] * per_gpu_batch / (ms * 1e-3) / 1e12, 2)
    if args.emulate_ranks > 1:
        result["emulated_allreduce_load"] = {"ranks": args.emulate_ranks, "busbw_gbps": args.emulate_busbw, "workgroups": args.emulate_wgs,
                                             "note": "one GPU; a stand-in kernel occupies the communication stream, no data crosses a link"}
    if on_rccl and pg.get_comm().one_rank_identity():
        result["config"]["one_rank_allreduce"] = "skipped (identity): reducer machinery only"
    if on_rccl:
        info_cs = pg.comm_stream_info() or {}
        result["config"]["comm_stream"] = {k: info_cs.get(k) for k in ("priority", "independent", "candidates_tried")}
    if use_gpu:
        result["peak_mem_gb"] = round(torch.cuda.max_memory_allocated(device) / 2**30, 2)
        result["peak_reserved_gb"] = round(torch.cuda.max_memory_reserved(device) / 2**30, 2)
    bail = None
    rccl_extras = use_gpu and on_rccl and not args.no_busbw
    if world_size > 1 or rccl_extras:
        # Everything below is collective and runs after the timed region: a rank that fails inside it leaves the
        # others waiting.  Every rank therefore arms a deadline; when it passes, rank 0 prints the line it already
        # has, every rank says on stderr which extra it was in, the communicator is aborted and the process ends
        # with the hang monitor's status (86): a collective that hangs after the timed steps is a hang, and must
        # be recorded as one by whoever launched the job.
        import threading

        from distributed_pytorch_amd.parallel import heartbeat as hb

        def _give_up():
            sys.stderr.write(f"[bench rank {rank}] post-run extras did not finish within {int(extras_deadline)} s; "
                             f"stuck in: {extras_phase[0]}\n")
            sys.stderr.flush()
            if dist.is_primary():
                result["extras"] = "abandoned after %d s in %s" % (int(extras_deadline), extras_phase[0])
                print(json.dumps(result), flush=True)
            pg.abort()
            os._exit(hb.EXIT_CODE_ABORTED)

        extras_phase = ["start"]

        extras_deadline = float(os.environ.get("BENCH_EXTRAS_DEADLINE_S", "120"))
        bail = threading.Timer(extras_deadline, _give_up)
        bail.daemon = True
        bail.start()
    if world_size > 1 and hasattr(model, "check_replicas_in_sync"):
        # Every rank's parameters must still be bit-identical (same averaged gradients, same update), which is what
        # a stream-ordering or bucket-layout bug in the data-parallel path would break.
        try:
            if bail is not None:
                extras_phase[0] = "check_replicas_in_sync"
            result["replicas_in_sync"] = bool(model.check_replicas_in_sync())
        except Exception as exc:  # noqa: BLE001 - diagnostic extra only
            result["replicas_in_sync"] = f"failed: {type(exc).__name__}: {exc}"
    if rccl_extras:
        # Never allowed to cost the result line:
        #  * overlap accounting over three extra steps: run time of the bucket all-reduces on the communication
        #    stream and how much of it the compute stream had to wait for (the part backward did not hide);
        #  * all-reduce bus bandwidth over bucket-sized messages, on the reducer's own path (communication
        #    stream + event fences), so the bucket size can be read off the curve.
        try:
            extras_phase[0] = "overlap report (3 extra steps)"
            result.update(_overlap_report(pg, step, device, model))
        except Exception as exc:  # noqa: BLE001 - diagnostic extra only
            result["exposed_comm_ms"] = f"failed: {type(exc).__name__}: {exc}"
        try:
            extras_phase[0] = "all-reduce bus-bandwidth sweep"
            result["allreduce_busbw_gbps"] = _allreduce_busbw(pg, dist, device, max(world_size, 1))
        except Exception as exc:  # noqa: BLE001 - diagnostic extra only
            result["allreduce_busbw_gbps"] = f"failed: {type(exc).__name__}: {exc}"
    if bail is not None:
        bail.cancel()
    if dist.is_primary():
        line = json.dumps(result)
        print(line, flush=True)
        if args.json_out:
            Path(args.json_out).parent.mkdir(parents=True, exist_ok=True)
            Path(args.json_out).write_text(line + "\n")
    dist.cleanup()


def _overlap_report(pg, step, device, model, n_steps=3):
    """Mean per-step communication-stream busy time, exposed (not overlapped) time and collective count."""
    import torch

    comm = pg.get_comm()
    torch.cuda.synchronize(device)
    comm.set_timing(True)
    busy = exposed = ops = 0.0
    try:
        for _ in range(n_steps):
            step()
            torch.cuda.synchronize(device)
            b, e, n = comm.read_timing()
            busy, exposed, ops = busy + b, exposed + e, ops + n
    finally:
        comm.set_timing(False)
    out = {"comm_busy_ms": round(busy / n_steps, 3), "exposed_comm_ms": round(exposed / n_steps, 3),
           "collectives_per_step": round(ops / n_steps, 1)}
    if hasattr(model, "bucket_sizes"):
        out["bucket_mib"] = [round(b / 2**20, 1) for b in model.bucket_sizes()]
        out["buckets_rebuilt"] = bool(getattr(model, "buckets_rebuilt", False))
        # bytes a rank puts on the links per step: 2 (N-1)/N of this for an all-reduce, and the same for a
        # reduce-scatter plus an all-gather of the same buffers ((N-1)/N each)
        out["gradient_mib_per_step"] = round(sum(model.bucket_sizes()) / 2**20, 1)
        out["collective_kind"] = "reduce_scatter + all_gather" if getattr(model, "shard_gradients", False) else "all_reduce"
        out["grads_written_in_place"] = int(model.reducer.num_grads_in_place())
        out["grads_packed_by_copy"] = int(model.reducer.num_grads_packed())
    return out


def _allreduce_busbw(pg, dist, device, world_size, sizes_mb=(16, 32, 64, 128, 256), iters=10):
    """Bus bandwidth (GB/s, 2(N-1)/N * bytes / time) of bf16 all-reduces of the given sizes."""
    import torch

    from distributed_pytorch_amd.utils.timer import allreduce_bus_bandwidth_gbps

    comm = pg.get_comm()
    out = {}
    for mb in sizes_mb:
        t = torch.ones((mb << 20) // 2, dtype=torch.bfloat16, device=device)
        for _ in range(2):
            comm.all_reduce_async(t, pg.OP_AVG, 0)
            comm.wait_slot(0)
        dist.barrier()
        torch.cuda.synchronize(device)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            comm.all_reduce_async(t, pg.OP_AVG, 0)
            comm.wait_slot(0)
        e.record()
        e.synchronize()
        sec = torch.tensor([s.elapsed_time(e) / 1e3 / iters], dtype=torch.float64, device=device)
        pg.all_reduce(sec, "max")
        out[f"{mb}MiB"] = round(allreduce_bus_bandwidth_gbps(mb << 20, float(sec.item()), world_size), 1)
        del t
    return out


def _spawned_worker(rank, world_size, argv):
    run_rank(rank, world_size, argv)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    # A benchmark step enters a collective several times a second: if NO rank enters one for five minutes the job
    # is stuck as a whole (e.g. inside its very first collective) and the hang monitor should end it with a
    # message instead of leaving it to the caller's time limit.  (The monitor's default only acts on a rank that
    # falls behind its peers, because a long phase without collectives is legitimate in general.)
    # Not for ResNet-50: the convolution library's first-run kernel search can take minutes without a collective.
    if args.model != "resnet50":
        os.environ.setdefault("DIST_STALL_TIMEOUT_S", "300")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world not in (None, ""):  # started by torchrun (or any launcher exporting the rendezvous env)
        run_rank(int(os.environ.get("RANK", "0")), int(env_world), argv)
        return
    if args.gpus <= 1:
        run_rank(0, 1, argv)
        return
    # plain `python bench.py --gpus N`: spawn the ranks ourselves
    import distributed as dist

    if not any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")):
        os.environ["HIP_VISIBLE_DEVICES"] = ",".join(str(i) for i in range(args.gpus))
    dist.launch(_spawned_worker, argv, nprocs=args.gpus)


if __name__ == "__main__":
    main()
