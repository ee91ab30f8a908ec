"""Single-GPU end-to-end checks: the RCCL communicator and the reducer at world size 1
(stream fencing, bucket life cycle, gradient views), and the min_DDP-equivalent model with
hidden_dim = 4096 on the HIP kernels against the CPU oracle."""
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture()
def rccl_group():
    """A real RCCL communicator with one rank, rendezvoused through our TCP store."""
    from distributed_pytorch_amd.parallel import process_group as pg
    from distributed_pytorch_amd.parallel.store import Store

    store = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
    pg.init_process_group(0, 1, "rccl", store=store)
    yield pg
    pg.destroy_process_group()


def test_rccl_collectives_single_rank(rccl_group):
    pg = rccl_group
    comm = pg.get_comm()
    assert comm.world == 1 and comm.rank == 0
    t = torch.arange(1000, device=DEV, dtype=torch.float32)
    ref = t.clone()
    comm.all_reduce(t, 0)
    comm.reduce(t, 0, 0)
    comm.broadcast(t, 0)
    assert torch.equal(t, ref)
    out = torch.zeros(1, 1000, device=DEV)
    comm.gather(t, out, 0)
    assert torch.equal(out[0], ref)
    b = torch.tensor([True, False, True], device=DEV)
    ob = torch.zeros(1, 3, dtype=torch.bool, device=DEV)
    comm.gather(b, ob, 0)
    assert ob[0].tolist() == [True, False, True]
    for dt in (torch.bfloat16, torch.float16, torch.float64, torch.int64, torch.int32, torch.uint8):
        x = torch.arange(17, device=DEV).to(dt)
        comm.all_reduce(x, 4 if dt.is_floating_point else 0)
        assert torch.equal(x, torch.arange(17, device=DEV).to(dt))
    ag = torch.zeros(1, 1000, device=DEV)
    comm.all_gather(t, ag)
    assert torch.equal(ag[0], ref)
    rs = torch.zeros(1000, device=DEV)
    comm.reduce_scatter(t, rs, 0)
    assert torch.equal(rs, ref)
    with pytest.raises(RuntimeError):
        comm.reduce_scatter(t, torch.zeros(999, device=DEV), 0)  # input must hold world * output elements
    assert pg.reduce_scatter(t) is t and pg.all_gather(t)[0] is t  # one rank: nothing to exchange
    comm.barrier()
    # comm-stream path: fill on the compute stream, reduce on the comm stream, consume after wait
    bufs = [torch.zeros(1 << 20, device=DEV) for _ in range(4)]   # one bucket per slot, like the reducer
    for i, buf in enumerate(bufs):
        buf.add_(float(i + 1))
        comm.all_reduce_async(buf, 4, i)
    comm.wait_all_slots()
    assert [b.sum().item() for b in bufs] == [float(i + 1) * (1 << 20) for i in range(4)]
    # the same bucket again: the producer is ordered behind the previous collective by the wait above
    for i, buf in enumerate(bufs):
        buf.add_(1.0)
        comm.all_reduce_async(buf, 4, i)
    comm.wait_all_slots()
    assert [b.sum().item() for b in bufs] == [float(i + 2) * (1 << 20) for i in range(4)]
    assert comm.num_async_ops() == 8


def test_ddp_wrapper_over_rccl_matches_plain_module(rccl_group):
    from distributed_pytorch_amd.ops import FusedAdamW
    from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

    torch.manual_seed(0)
    plain = nn.Sequential(nn.Linear(64, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 10)).to(DEV)
    twin = nn.Sequential(nn.Linear(64, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 10)).to(DEV)
    twin.load_state_dict(plain.state_dict())
    ddp = DistributedDataParallel(twin, device_ids=[0], bucket_cap_mb=0.1, first_bucket_mb=0.05)
    nb0 = ddp.reducer.num_buckets()
    assert nb0 >= 2
    assert all(k.startswith("module.") for k in ddp.state_dict())
    o1 = torch.optim.AdamW(plain.parameters(), 1e-3)
    o2 = FusedAdamW(ddp.parameters(), 1e-3)
    for step in range(5):
        x = torch.randn(32, 64, device=DEV)
        y = torch.randint(0, 10, (32,), device=DEV)
        for model, opt in ((plain, o1), (ddp, o2)):
            opt.zero_grad()
            nn.functional.cross_entropy(model(x), y).backward()
            opt.step()
    assert ddp.num_backward_syncs() == 5
    assert ddp.num_bucket_launches() == nb0 + 4 * ddp.reducer.num_buckets()  # re-planned once, after the first pass
    arena = ddp.grad_arenas()[0]
    lo, hi = arena.data_ptr(), arena.data_ptr() + arena.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in ddp.parameters())  # grads are bucket views
    for a, b in zip(plain.parameters(), ddp.parameters()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    # no_sync: two accumulated micro-batches equal one big batch
    ddp.zero_grad(set_to_none=True)
    plain.zero_grad(set_to_none=True)
    x = torch.randn(64, 64, device=DEV)
    y = torch.randint(0, 10, (64,), device=DEV)
    with ddp.no_sync():
        (nn.functional.cross_entropy(ddp(x[:32]), y[:32]) / 2).backward()
    (nn.functional.cross_entropy(ddp(x[32:]), y[32:]) / 2).backward()
    nn.functional.cross_entropy(plain(x), y).backward()
    for a, b in zip(plain.parameters(), ddp.parameters()):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-6)


def test_rccl_debug_sync_overlap_timing_and_device_identity(rccl_group):
    """DIST_DEBUG_SYNC serialises the reducer's communication-stream path (the stream is idle when
    all_reduce_async returns), the overlap accounting reports sane numbers, and the device identity
    used by the pre-init uniqueness check is a PCI bus id."""
    import re

    pg = rccl_group
    comm = pg.get_comm()
    C = type(comm)
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", C.device_bus_id(0)), C.device_bus_id(0)
    assert re.fullmatch(r"\d+\.\d+\.\d+", C.version())
    buf = torch.ones(8 << 20, device=DEV)
    n0 = comm.num_async_ops()
    assert not comm.debug_sync()
    comm.set_debug_sync(True)
    try:
        for i in range(3):
            buf.mul_(2.0)                       # work queued on the compute stream just before
            comm.all_reduce_async(buf, 4, i)
            assert comm.comm_stream_idle()      # already finished: nothing can overlap under the flag
    finally:
        comm.set_debug_sync(False)
    comm.wait_all_slots()
    assert comm.num_async_ops() == n0 + 3
    assert buf[0].item() == 8.0
    # overlap accounting
    comm.set_timing(True)
    for i in range(4):
        buf.add_(1.0)
        comm.all_reduce_async(buf, 4, i)
    comm.wait_all_slots()
    torch.cuda.synchronize()
    busy, exposed, n = comm.read_timing()
    comm.set_timing(False)
    assert n == 4 and 0.0 < busy < 1000.0 and 0.0 <= exposed < 1000.0
    assert comm.read_timing() == [0.0, 0.0, 0.0]


def test_comm_stream_waits_on_a_busy_compute_stream(rccl_group):
    """The fence the communication stream waits on must work when the compute stream has NOT reached it
    yet (long kernels queued ahead), for bucket-sized buffers, repeatedly on one slot.  A wait on an
    already-completed event proves nothing."""
    comm = rccl_group.get_comm()
    a = torch.randn(4096, 4096, device=DEV)
    for mb in (1, 4, 16, 64, 128):
        buf = torch.ones((mb << 20) // 2, device=DEV, dtype=torch.bfloat16)
        for _ in range(8):
            a = (a @ a).clamp_(-1, 1)          # ~10 ms of work ahead of the fence
        buf.add_(1.0)
        comm.all_reduce_async(buf, 4, 0)
        comm.wait_all_slots()
        buf.add_(1.0)                          # consumer on the compute stream
        torch.cuda.synchronize()
        assert buf[0].item() == 3.0 and buf[-1].item() == 3.0


def test_full_width_gpt2_step_under_the_reducer(rccl_group):
    """GPT-2 small width (768 / 12 heads / 1024 tokens, 2 layers) with the default bucket sizes: the
    77 MB tied embedding gradient goes out as sliced all-reduces behind a backward pass that is still
    running, and the wrapped model trains exactly like the plain one."""
    from distributed_pytorch_amd.models import GPT2, GPT2Config
    from distributed_pytorch_amd.ops import FusedAdamW
    from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

    cfg = GPT2Config.small(block_size=1024)
    cfg.n_layer = 2
    torch.manual_seed(0)
    plain = GPT2(cfg).to(DEV).bfloat16()
    twin = GPT2(cfg).to(DEV).bfloat16()
    twin.load_state_dict(plain.state_dict())
    ddp = DistributedDataParallel(twin)
    assert max(ddp.reducer.slices_per_bucket()) >= 2
    o1 = FusedAdamW(plain.parameters(), 1e-4, master_weights=True)
    o2 = FusedAdamW(ddp.parameters(), 1e-4, master_weights=True)
    idx = torch.randint(0, cfg.vocab_size, (4, 1024), device=DEV)
    tgt = torch.randint(0, cfg.vocab_size, (4, 1024), device=DEV)
    for step in range(3):
        losses = []
        for model, opt in ((plain, o1), (ddp, o2)):
            opt.zero_grad(set_to_none=True)
            loss = model(idx, tgt)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        assert losses[0] == losses[1]
    for (n, a), b in zip(plain.named_parameters(), ddp.module.parameters()):
        assert torch.equal(a, b), n


class _Scrambled(nn.Module):
    def __init__(self):
        super().__init__()
        self.last = nn.Linear(256, 10)
        self.first = nn.Linear(64, 256)
        self.mid = nn.Linear(256, 256)

    def forward(self, x):
        return self.last(torch.relu(self.mid(torch.relu(self.first(x)))))


def test_bucket_rebuild_and_packed_parameter_broadcast_on_gpu(rccl_group):
    from distributed_pytorch_amd.ops import FusedAdamW
    from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

    torch.manual_seed(0)
    plain, twin = _Scrambled().to(DEV), _Scrambled().to(DEV)
    twin.load_state_dict(plain.state_dict())
    ddp = DistributedDataParallel(twin, bucket_cap_mb=0.1, first_bucket_mb=0.05, max_collective_mb=0.1)
    nb = ddp.reducer.num_buckets()
    before = [list(b) for b in ddp._bucket_params]
    # initial parameter sync: one pack launch, one broadcast, one unpack launch per bucket
    n0 = ddp.reducer.num_pack_launches()
    ddp.reducer.broadcast_parameters(0)
    assert ddp.reducer.num_pack_launches() - n0 == 2 * nb
    for a, b in zip(plain.parameters(), ddp.module.parameters()):
        assert torch.equal(a, b)
    o1, o2 = torch.optim.AdamW(plain.parameters(), 1e-3), FusedAdamW(ddp.parameters(), 1e-3)
    for step in range(4):
        x = torch.randn(32, 64, device=DEV)
        y = torch.randint(0, 10, (32,), device=DEV)
        for model, opt in ((plain, o1), (ddp, o2)):
            opt.zero_grad(set_to_none=(step % 2 == 0))   # both zeroing styles cross the rebuild
            nn.functional.cross_entropy(model(x), y).backward()
            opt.step()
    assert ddp.buckets_rebuilt and [list(b) for b in ddp._bucket_params] != before
    flat = [i for b in ddp._bucket_params for i in b]
    assert flat == [int(i) for i in ddp.reducer.ready_order()]
    sizes, total = [], 0
    for b in ddp._bucket_params:
        total += len(b)
        sizes.append(total)
    assert [int(i) for i in ddp.reducer.launch_positions()] == sizes
    assert max(ddp.reducer.slices_per_bucket()) > 1
    arena = ddp.grad_arenas()[0]
    lo, hi = arena.data_ptr(), arena.data_ptr() + arena.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in ddp.parameters())
    for a, b in zip(plain.parameters(), ddp.parameters()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def test_gpt2_with_fused_nodes_under_the_reducer_matches_plain_training(rccl_group):
    """GPT-2 (tied embedding / lm-head weight, fused MLP, projection + attention and lm-head + loss nodes)
    wrapped in the data-parallel module: bucket hooks fire for every parameter, gradients live in the
    arena, and five steps give exactly the weights of unwrapped training."""
    from distributed_pytorch_amd.models import GPT2, GPT2Config
    from distributed_pytorch_amd.ops import FusedAdamW
    from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

    cfg = GPT2Config(block_size=256, vocab_size=1000, n_layer=2, n_head=2, n_embd=128)
    torch.manual_seed(0)
    plain = GPT2(cfg).to(DEV).bfloat16()
    twin = GPT2(cfg).to(DEV).bfloat16()
    twin.load_state_dict(plain.state_dict())
    ddp = DistributedDataParallel(twin, device_ids=[0], bucket_cap_mb=0.25, first_bucket_mb=0.05)
    nb0 = ddp.reducer.num_buckets()
    assert nb0 >= 3
    o1 = FusedAdamW(plain.parameters(), 1e-3, master_weights=True)
    o2 = FusedAdamW(ddp.parameters(), 1e-3, master_weights=True)
    g = torch.Generator().manual_seed(1)
    losses = ([], [])
    for step in range(5):
        idx = torch.randint(0, 1000, (4, 256), generator=g).to(DEV)
        tgt = torch.randint(0, 1000, (4, 256), generator=g).to(DEV)
        for k, (model, opt) in enumerate(((plain, o1), (ddp, o2))):
            opt.zero_grad(set_to_none=True)
            loss = model(idx, tgt)
            loss.backward()
            opt.step()
            losses[k].append(loss.item())
    assert ddp.num_backward_syncs() == 5
    assert ddp.num_bucket_launches() == nb0 + 4 * ddp.reducer.num_buckets()
    # the fused nodes hand several gradients over out of registration order: the layout follows them
    assert ddp.buckets_rebuilt and [i for b in ddp._bucket_params for i in b] == [int(i) for i in ddp.reducer.ready_order()]
    assert losses[0] == losses[1], losses
    for (n, a), b in zip(plain.named_parameters(), ddp.module.parameters()):
        assert torch.equal(a, b), n


def test_ddp_bf16_params_fp32_reduce(rccl_group):
    from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

    torch.manual_seed(1)
    m = nn.Sequential(nn.Linear(128, 128), nn.GELU(), nn.Linear(128, 8)).to(DEV).bfloat16()
    twin = nn.Sequential(nn.Linear(128, 128), nn.GELU(), nn.Linear(128, 8)).to(DEV).bfloat16()
    twin.load_state_dict(m.state_dict())
    ddp = DistributedDataParallel(twin, reduce_dtype="fp32")
    assert ddp.grad_arenas()[0].dtype == torch.float32
    x = torch.randn(16, 128, device=DEV).bfloat16()
    m(x).float().pow(2).mean().backward()
    ddp(x).float().pow(2).mean().backward()
    for a, b in zip(m.parameters(), ddp.parameters()):
        assert b.grad.dtype == torch.bfloat16
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-2, atol=1e-3)


def _train_dummy(device, native, hidden, steps=6, batch=64, seed=0):
    from distributed_pytorch_amd.models.dummy import DummyDataset, DummyModel
    from distributed_pytorch_amd.ops import FusedAdamW, SoftmaxCrossEntropy

    torch.manual_seed(seed)
    model = DummyModel(1, hidden, 4, native=native)
    ds = DummyDataset(batch * steps, 4)
    model.to(device)
    if native:
        opt, crit = FusedAdamW(model.parameters(), 1e-4), SoftmaxCrossEntropy()
    else:
        opt, crit = torch.optim.AdamW(model.parameters(), 1e-4), nn.CrossEntropyLoss()
    losses, accs = [], []
    for s in range(steps):
        x = ds.data[s * batch:(s + 1) * batch].to(device) / (batch * steps)  # keep activations O(1)
        y = ds.labels[s * batch:(s + 1) * batch].to(device)
        out = model(x)
        loss = crit(out, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        correct = crit.last_correct if native else out.argmax(1).eq(y)
        accs.append(correct.float().mean().item())
    return losses, accs, {k: v.detach().cpu() for k, v in model.state_dict().items()}


def test_min_ddp_hidden4096_native_matches_cpu_oracle():
    """The min_DDP-equivalent model at hidden_dim 4096: HIP linear (K=1 / N=4 kernels), fused
    cross-entropy + accuracy and fused AdamW track the stock CPU implementation."""
    l_cpu, a_cpu, sd_cpu = _train_dummy(torch.device("cpu"), native=False, hidden=4096)
    l_gpu, a_gpu, sd_gpu = _train_dummy(torch.device(DEV), native=True, hidden=4096)
    torch.testing.assert_close(torch.tensor(l_gpu), torch.tensor(l_cpu), rtol=1e-4, atol=1e-5)
    assert a_gpu == a_cpu
    for k in sd_cpu:
        torch.testing.assert_close(sd_gpu[k], sd_cpu[k], rtol=1e-4, atol=1e-6)


def test_native_extension_is_loaded_and_used():
    from distributed_pytorch_amd import _ext
    from distributed_pytorch_amd.ops.linear import native_route

    assert _ext.have_C()
    assert native_route(64, 4096, 1, torch.float32) == "outer"
    assert native_route(64, 4, 4096, torch.float32) == "skinny"
    assert native_route(256, 8192, 8192, torch.bfloat16) == "mfma"
    assert os.path.basename(_ext.C().__file__).startswith("_C")


def test_device_prefetcher_delivers_exact_batches_on_device():
    from torch.utils.data import DataLoader, TensorDataset

    from distributed_pytorch_amd.utils.data import DevicePrefetcher

    torch.manual_seed(0)
    xs, ys = torch.randn(1000, 33), torch.randint(0, 10, (1000,))
    loader = DataLoader(TensorDataset(xs, ys), batch_size=96, shuffle=False)  # last batch is short: staging is re-sized
    dev = torch.device("cuda:0")
    seen = 0
    for depth in (1, 2, 3):
        for i, (x, y) in enumerate(DevicePrefetcher(loader, dev, depth=depth)):
            assert x.device == dev and y.device == dev
            lo = i * 96
            # consume on the compute stream, as a training step would
            assert torch.equal((x * 1).cpu(), xs[lo:lo + 96]) and torch.equal(y.cpu(), ys[lo:lo + 96])
            seen += x.shape[0]
    assert seen == 3000


def _run_example_gpu(args, env_extra, timeout=300):
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    env.update(env_extra)
    env["PYTHONPATH"] = str(root) + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, str(root / "min_DDP.py"), *args], cwd=str(root), env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_min_ddp_two_ranks_on_one_gpu_match_reference_goldens():
    """Two processes share the single GPU (gloo carries the collectives, since RCCL refuses two
    ranks on one device): the native kernels, the C++ reducer with device-resident buckets, the
    launcher and the metric collectives run in a real multi-process setting on the GPU and must
    print the numbers measured with the reference code on CPU."""
    p = _run_example_gpu(["--seed", "0", "--backend", "gloo"], {"DIST_WORLD_SIZE": "2", "HIP_VISIBLE_DEVICES": "0", "DIST_ALLOW_GPU_SHARING": "1"})
    assert p.returncode == 0, p.stdout[-3000:]
    assert "Device: cuda:0" in p.stdout
    lines = [l for l in p.stdout.splitlines() if l.startswith("Finish iteration")]
    assert len(lines) == 4
    assert lines[0] == "Finish iteration 0 - acc: 0.1875 (3/16) - loss: 6.2573"
    assert lines[1] == "Finish iteration 1 - acc: 0.3750 (6/16) - loss: 14.9106"


def test_bench_two_ranks_on_one_gpu_gloo_ddp_contract():
    """bench.py self-spawning 2 ranks on the one GPU (gloo; a plumbing configuration, slow by
    nature): launcher, store, DDP wrapper with device-resident buckets and the JSON contract."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    env["PYTHONPATH"] = str(root) + os.pathsep + env.get("PYTHONPATH", "")
    env["HIP_VISIBLE_DEVICES"] = "0"
    env["DIST_ALLOW_GPU_SHARING"] = "1"
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--model", "dummy", "--batch-size", "512",
                        "--steps", "2", "--warmup", "1"], cwd=str(root), env=env, timeout=300, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 1024 and r["config"]["parallelism"] == "dp2"
    assert r["steps"] == 2 and r["warmup"] == 1 and r["value"] > 0


def test_full_size_gpt2_two_ranks_on_one_gpu_over_gloo():
    """Two ranks sharing the GPU, GPT-2 small at full width under the reducer with the host-side (gloo)
    data plane: the backward leaves a long backlog on the compute stream when the first bucket is reduced,
    which used to stall the group's staging streams for good (the transport now drains the stream first)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    env["PYTHONPATH"] = str(root) + os.pathsep + env.get("PYTHONPATH", "")
    env["HIP_VISIBLE_DEVICES"] = "0"
    env["DIST_ALLOW_GPU_SHARING"] = "1"
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch-size", "4", "--steps", "2",
                        "--warmup", "1"], cwd=str(root), env=env, timeout=110, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["config"]["model"] == "gpt2-small" and r["config"]["comm"] == "gloo"
    assert r["replicas_in_sync"] is True
    assert r["final_loss"] < r["first_loss"]


def test_sharded_optimizer_two_ranks_on_one_gpu_equals_full_optimizer():
    """bf16 GPT-2 under the reducer on two ranks sharing the GPU (gloo): with the optimizer state sharded by
    parameter ownership (flat device buffers the owned parameters live in, fp32 master weights on the owner only,
    one broadcast per owner) the run ends on the same loss, digit for digit, as with a full optimizer per rank."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    env["PYTHONPATH"] = str(root) + os.pathsep + env.get("PYTHONPATH", "")
    env["HIP_VISIBLE_DEVICES"] = "0"
    env["DIST_ALLOW_GPU_SHARING"] = "1"
    runs = []
    for extra in ([], ["--shard-optimizer"]):
        p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch-size", "2", "--seq-len", "256",
                            "--steps", "4", "--warmup", "2"] + extra, cwd=str(root), env=env, timeout=110, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout[-3000:]
        runs.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1]))
    full, sharded = runs
    assert sharded["config"]["optimizer"].startswith("ShardedOptimizer(") and not full["config"]["optimizer"].startswith("Sharded")
    assert sharded["first_loss"] == full["first_loss"] and sharded["final_loss"] == full["final_loss"]
    assert sharded["final_loss"] < sharded["first_loss"]
    assert sharded["replicas_in_sync"] is True and full["replicas_in_sync"] is True
    assert sharded["peak_mem_gb"] < full["peak_mem_gb"]  # half of the master weights and moments per rank


def test_rccl_group_through_torch_distributed_run_store():
    """The rendezvous path of the multi-GPU benchmark (torch.distributed.run hosts the store on
    MASTER_PORT, the ncclUniqueId travels through it), at the world size one GPU allows."""
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "rccl_torchrun_probe.py")]
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "rccl over the launcher store ok: world 1" in out.stdout


def test_bench_bus_bandwidth_probe_runs_on_the_reducer_path(rccl_group):
    """The post-run all-reduce bandwidth measurement of bench.py (only taken with N > 1): same calls at
    world size 1, where the formula's 2(N-1)/N factor makes the result zero."""
    import importlib.util

    import distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n0 = rccl_group.get_comm().num_async_ops()
    out = bench._allreduce_busbw(rccl_group, dist, torch.device("cuda", 0), 1, sizes_mb=(1, 4), iters=3)
    assert out == {"1MiB": 0.0, "4MiB": 0.0}
    assert rccl_group.get_comm().num_async_ops() == n0 + 2 * (2 + 3)


def _reducer_stream_sequence(comm, reps=2):
    """The reducer's stream traffic on four 4 MiB buckets behind a backlog of matmuls: fill on the compute stream,
    all-reduce on the communication stream (which therefore waits for an event the compute stream has not
    reached yet), compute stream waits for the completion events, host consumes."""
    x = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    bufs = [torch.zeros(1 << 20, device=DEV) for _ in range(4)]
    for rep in range(reps):
        for i, buf in enumerate(bufs):
            for _ in range(4):
                x = (x @ x).clamp_(-1, 1)           # keeps the compute stream busy ahead of the fence
            buf.add_(1.0)
            comm.all_reduce_async(buf, 4, i)
        comm.wait_all_slots()
        assert [b.sum().item() for b in bufs] == [float(rep + 1) * (1 << 20)] * 4


def test_eight_groups_after_gpu_work_share_one_probed_stream():
    """Communicators created one after another in a process that has long been using the GPU (this test file runs
    behind the kernel tests): every one gets the SAME long-lived communication stream, chosen once by the bounded
    independence probe, and the reducer's stream sequence runs on each of them."""
    from distributed_pytorch_amd.parallel import process_group as pg
    from distributed_pytorch_amd.parallel.store import Store

    (torch.randn(1024, 1024, device=DEV) @ torch.randn(1024, 1024, device=DEV)).sum().item()   # GPU work first
    handles = []
    for k in range(8):
        store = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
        pg.init_process_group(0, 1, "rccl", store=store)
        try:
            comm = pg.get_comm()
            info = pg.comm_stream_info()
            assert info["independent"], info            # the probe found a stream that overlaps the compute stream
            assert info["priority"] == os.environ.get("DIST_COMM_STREAM", "normal")
            handles.append(comm.comm_stream_handle())
            _reducer_stream_sequence(comm)
        finally:
            pg.destroy_process_group()
    assert len(set(handles)) == 1, handles


def test_stream_independence_probe_tells_shared_from_separate_queues():
    """The probe behind the stream choice: a kernel on the compute stream is held up by a spin kernel on the SAME
    stream, and is not by one on the chosen communication stream."""
    from distributed_pytorch_amd import _ext
    from distributed_pytorch_amd.parallel import process_group as pg
    from distributed_pytorch_amd.parallel.store import Store

    C = _ext.C()
    cur = torch.cuda.current_stream().cuda_stream
    same = C.probe_stream_independence(0, cur, cur, 3.0)
    assert same > 2.0, same                                # behind the 3 ms spin on its own stream
    store = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
    pg.init_process_group(0, 1, "rccl", store=store)
    try:
        other = C.probe_stream_independence(0, pg.get_comm().comm_stream_handle(), cur, 3.0)
    finally:
        pg.destroy_process_group()
    assert other < 1.5, other


def _run_bench_subprocess(extra_env, *extra_args):
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, DIST_QUIET_INIT="1", **extra_env)
    cmd = [sys.executable, str(root / "bench.py"), "--force-ddp", "--batch-size", "8", "--steps", "2", "--warmup", "2", *extra_args]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_under_the_reducer_in_a_fresh_process():
    """The production order -- process group first, model afterwards, in a fresh process -- on the full-width
    GPT-2 under the reducer, with the default (normal priority) and the high-priority communication stream."""
    line = _run_bench_subprocess({})
    assert line["config"]["comm"] == "rccl" and line["config"]["comm_stream"]["priority"] == "normal"
    assert line["config"]["comm_stream"]["independent"] is True
    assert line["value"] > 0 and line["collectives_per_step"] >= 2
    line = _run_bench_subprocess({"DIST_COMM_STREAM": "high"})
    assert line["config"]["comm_stream"]["independent"] is True   # whichever priority the probe settled on
    assert line["value"] > 0


def test_weight_gradients_land_in_the_bucket_and_optimizer_waits_per_bucket(rccl_group):
    """Operators with their own weight-gradient kernels write straight into the parameter's bucket slice (no pack
    copy for those tensors), and ``FusedAdamW(overlap_grad_reduce=True)`` takes over the waits for the all-reduces
    bucket by bucket; five steps still give exactly the weights of unwrapped training."""
    from distributed_pytorch_amd.models import GPT2, GPT2Config
    from distributed_pytorch_amd.ops import FusedAdamW
    from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

    cfg = GPT2Config(block_size=256, vocab_size=1024, n_layer=2, n_head=2, n_embd=128)
    torch.manual_seed(0)
    plain = GPT2(cfg).to(DEV).bfloat16()
    twin = GPT2(cfg).to(DEV).bfloat16()
    twin.load_state_dict(plain.state_dict())
    ddp = DistributedDataParallel(twin, device_ids=[0], bucket_cap_mb=0.25, first_bucket_mb=0.05)
    o1 = FusedAdamW(plain.parameters(), 1e-3, master_weights=True)
    o2 = FusedAdamW(ddp.parameters(), 1e-3, master_weights=True, overlap_grad_reduce=True)
    g = torch.Generator().manual_seed(1)
    pending = []
    for step in range(5):
        idx = torch.randint(0, 1024, (4, 256), generator=g).to(DEV)
        tgt = torch.randint(0, 1024, (4, 256), generator=g).to(DEV)
        for model, opt in ((plain, o1), (ddp, o2)):
            opt.zero_grad(set_to_none=True)
            loss = model(idx, tgt)
            loss.backward()
            if model is ddp:
                pending.append(bool(ddp.reducer.waits_pending()))
            opt.step()
            if model is ddp:
                assert not ddp.reducer.waits_pending()
    # the optimizer switched the deferral on in its first step; from the second backward on the waits are its own
    assert pending == [False, True, True, True, True], pending
    n_weights = sum(1 for n, p in ddp.module.named_parameters() if p.dim() == 2 and "wpe" not in n)
    assert ddp.reducer.num_grads_in_place() >= 4 * (n_weights - 1), (ddp.reducer.num_grads_in_place(), n_weights)
    for (n, a), b in zip(plain.named_parameters(), ddp.module.parameters()):
        assert torch.equal(a, b), n
    # a consumer that is not the optimizer asks for the gradients explicitly
    ddp.zero_grad(set_to_none=True)
    ddp(idx, tgt).backward()
    assert ddp.reducer.waits_pending()
    ddp.wait_for_gradients()
    assert not ddp.reducer.waits_pending()
    total = torch.stack([p.grad.float().norm() for p in ddp.parameters()]).sum().item()
    assert total == total and total > 0


def test_sigterm_reaches_a_rank_that_is_blocked_in_a_native_wait(tmp_path):
    """A rank sitting inside torch.cuda.synchronize() behind seconds of queued GPU work cannot run a Python signal
    handler; SIGTERM must still abort its communicator and end it within the grace period (foreign launchers and
    schedulers send SIGTERM and expect the process gone)."""
    import signal
    import subprocess
    import sys
    import time
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    script = tmp_path / "blocked_rank.py"
    script.write_text(
        "import sys, torch\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "from distributed_pytorch_amd.parallel import process_group as pg\n"
        "from distributed_pytorch_amd.parallel.store import Store\n"
        "pg.init_process_group(0, 1, 'rccl', store=Store('127.0.0.1', 0, is_host=True, bind_addr='127.0.0.1'))\n"
        "x = torch.zeros(8, device='cuda')\n"
        "for _ in range(12):\n"
        "    torch.cuda._sleep(2_000_000_000)\n"      # about a second of GPU time each
        "print('blocked', flush=True)\n"
        "torch.cuda.synchronize()\n"
        "print('woke up by itself', flush=True)\n")
    proc = subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            env=dict(os.environ, DIST_QUIET_INIT="1"))
    assert proc.stdout.readline().strip() == "blocked"
    time.sleep(0.5)
    t0 = time.time()
    proc.send_signal(signal.SIGTERM)
    try:
        rc = proc.wait(timeout=8)
    finally:
        if proc.poll() is None:
            proc.kill()
    took = time.time() - t0
    assert rc in (-signal.SIGTERM, 128 + signal.SIGTERM), (rc, proc.stderr.read()[-2000:])
    assert took < 6.0, took


def test_range_sharded_optimizer_over_rccl_equals_full_optimizer(rccl_group):
    """shard_gradients=True + ShardedOptimizer (reduce-scatter -> range update on flat fp32 state -> all-gather, all on
    the communication stream with per-bucket waits) on bf16 GPT-2 with master weights: bit-equal to FusedAdamW on the
    unwrapped model, with and without clipping; state converts to the unsharded layout and back."""
    from distributed_pytorch_amd.models import GPT2, GPT2Config
    from distributed_pytorch_amd.ops import FusedAdamW
    from distributed_pytorch_amd.parallel import ShardedOptimizer
    from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel

    cfg = GPT2Config(block_size=128, vocab_size=512, n_layer=2, n_head=2, n_embd=128)
    for clip in (None, 1.0):
        torch.manual_seed(0)
        plain = GPT2(cfg).to(DEV).bfloat16()
        twin = GPT2(cfg).to(DEV).bfloat16()
        twin.load_state_dict(plain.state_dict())
        ddp = DistributedDataParallel(twin, device_ids=[0], bucket_cap_mb=0.25, first_bucket_mb=0.05, shard_gradients=True)
        o1 = FusedAdamW(plain.parameters(), 1e-3, master_weights=True, max_grad_norm=clip)
        o2 = ShardedOptimizer(ddp.parameters(), lr=1e-3, master_weights=True, max_grad_norm=clip)
        assert o2._arena is not None and len(o2._arena["buckets"]) >= 3
        g = torch.Generator().manual_seed(1)
        for step in range(4):
            idx = torch.randint(0, 512, (4, 128), generator=g).to(DEV)
            tgt = torch.randint(0, 512, (4, 128), generator=g).to(DEV)
            for model, opt in ((plain, o1), (ddp, o2)):
                opt.zero_grad(set_to_none=True)
                model(idx, tgt).backward()
                opt.step()
        # without clipping the update is element for element the full optimizer's; with it the global norm is summed
        # range by range instead of tensor by tensor, so the clip coefficient may differ in its last bit
        same = (lambda x, y: torch.equal(x, y)) if clip is None else (lambda x, y: torch.allclose(x.float(), y.float(), rtol=2e-2, atol=1e-5))
        for (n, a), b in zip(plain.named_parameters(), ddp.module.parameters()):
            assert same(a, b), (clip, n)
        if clip is not None:
            torch.testing.assert_close(o2.last_grad_norm.reshape(()), o1.last_grad_norm.reshape(()), rtol=1e-5, atol=1e-7)
        cons = o2.consolidated_state_dict()
        ref = o1.state_dict()["state"]
        assert sorted(cons["state"]) == sorted(ref)
        for k in ref:
            for f in ("exp_avg", "exp_avg_sq", "master"):
                assert same(cons["state"][k][f].cpu(), ref[k][f].cpu()), (k, f)
        o2.load_state_dict(cons)
        assert ddp.reducer.num_collectives() >= 2 * 4 * len(o2._arena["buckets"])
