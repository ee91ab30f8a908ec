"""The drop-in API contract of the top-level ``distributed`` module without any process
group: every query / collective degrades to the single-process answer, exactly as the
reference helper does (reference distributed.py:69-101, 119-187)."""
import inspect
import os
import io
import socket
from contextlib import redirect_stdout

import pytest
import torch
import torch.nn as nn

import distributed as dist

API = ["find_free_port", "launch", "init_process_group", "is_dist_avail_and_initialized", "cleanup", "get_rank",
       "get_device", "is_primary", "get_world_size", "data_sampler", "prepare_ddp_model", "all_reduce", "reduce",
       "gather", "sync_params", "barrier", "wait_for_everyone", "print_primary"]


def test_all_eighteen_functions_exist_with_reference_signatures():
    for name in API:
        assert callable(getattr(dist, name)), name
    assert list(inspect.signature(dist.init_process_group).parameters) == ["rank", "world_size", "backend"]
    assert list(inspect.signature(dist.data_sampler).parameters) == ["dataset", "distributed", "shuffle"]
    assert list(inspect.signature(dist.prepare_ddp_model).parameters)[:2] == ["model", "device_ids"]
    assert inspect.signature(dist.all_reduce).parameters["op"].default == "sum"
    assert inspect.signature(dist.reduce).parameters["op"].default == torch.distributed.ReduceOp.SUM
    assert list(inspect.signature(dist.launch).parameters)[:2] == ["worker_fn", "args"]


def test_single_process_fallbacks():
    assert not dist.is_dist_avail_and_initialized()
    assert dist.get_rank() == 0
    assert dist.get_world_size() == 1
    assert dist.is_primary()
    t = torch.arange(4.0)
    assert dist.all_reduce(t) is t
    assert dist.all_reduce(t, "avg") is t
    assert dist.all_reduce(t, "no-such-op") is t  # the op is not even looked at in a single process
    assert dist.reduce(t) is t
    g = dist.gather(t)
    assert isinstance(g, list) and len(g) == 1 and g[0] is t
    assert dist.barrier() is None and dist.wait_for_everyone() is None
    assert dist.sync_params([t]) is None
    assert dist.cleanup() is None  # safe without a group
    assert torch.equal(t, torch.arange(4.0))


def test_get_device_type():
    d = dist.get_device()
    assert isinstance(d, torch.device)
    assert d.type == ("cuda" if torch.cuda.is_available() else "cpu")


def test_data_sampler_and_prepare_ddp_model_without_group():
    ds = list(range(10))
    assert dist.data_sampler(ds, False, True) is None
    s = dist.data_sampler(ds, True, False)  # world 1: every index, in order
    assert list(s) == list(range(10)) and len(s) == 10
    m = nn.Linear(2, 2)
    assert dist.prepare_ddp_model(m, device_ids=[0]) is m  # unwrapped: state_dict keys keep no prefix
    assert list(m.state_dict()) == ["weight", "bias"]


def test_print_primary_has_print_signature():
    buf = io.StringIO()
    with redirect_stdout(buf):
        dist.print_primary("a", 1, sep="-", end="!\n")
    assert buf.getvalue() == "a-1!\n"


def test_find_free_port_is_bindable():
    p = dist.find_free_port()
    assert 1024 < p < 65536
    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", p))


# --------------------------------------------------------------------------- launch branch selection
def test_launch_runs_in_process_for_zero_and_one_device(monkeypatch):
    calls = []
    monkeypatch.delenv("DIST_WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    dist.launch(lambda rank, world, *a: calls.append((rank, world, a)), "x", 3)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    dist.launch(lambda rank, world, *a: calls.append((rank, world, a)))
    # CPU passes world_size 0, one GPU passes 1 (reference distributed.py:54-58)
    assert calls == [(0, 0, ("x", 3)), (0, 1, ())]


def test_launch_refuses_unpinned_multi_gpu(monkeypatch):
    monkeypatch.delenv("DIST_WORLD_SIZE", raising=False)
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(ValueError, match="GPUs not specified"):
        dist.launch(lambda *a: None)


def test_child_device_env_binding_modes(monkeypatch):
    from distributed_pytorch_amd.parallel.launcher import child_device_env, visible_device_ids

    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    ids = ["2", "3", "5", "7"]
    e = child_device_env(ids, 2, "rotate")
    assert e["HIP_VISIBLE_DEVICES"] == "5,7,2,3" and e["DIST_LOCAL_DEVICE"] == "0" and e["CUDA_VISIBLE_DEVICES"] is None
    assert child_device_env(ids, 1, "exclusive")["HIP_VISIBLE_DEVICES"] == "3"
    assert child_device_env(ids, 3, "index") == {"DIST_LOCAL_DEVICE": "3"}
    with pytest.raises(ValueError):
        child_device_env(ids, 0, "bogus")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,6")
    assert visible_device_ids(2) == ["4", "6"]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "1,0")
    assert visible_device_ids(2) == ["1", "0"]
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES")
    assert visible_device_ids(2) is None


def test_launcher_never_disables_p2p():
    import distributed_pytorch_amd.parallel.launcher as L

    src = inspect.getsource(L.launch) + inspect.getsource(L.spawn)
    assert 'os.environ["NCCL_P2P_DISABLE"]' not in src and "NCCL_P2P_DISABLE=" not in src


def test_cpu_slices_partition_affinity():
    import os

    from distributed_pytorch_amd.parallel.launcher import cpu_slices

    cpus = sorted(os.sched_getaffinity(0))
    sl = cpu_slices(2)
    assert len(sl) == 2 and not set(sl[0]) & set(sl[1])
    assert set(sl[0]) | set(sl[1]) <= set(cpus)


def test_device_prefetcher_cpu_passthrough_preserves_batches_and_structure():
    from torch.utils.data import DataLoader, TensorDataset

    from distributed_pytorch_amd.utils.data import DevicePrefetcher

    ds = TensorDataset(torch.arange(10).float().unsqueeze(1), torch.arange(10))
    loader = DataLoader(ds, batch_size=4, shuffle=False)
    pf = DevicePrefetcher(loader, torch.device("cpu"))
    assert len(pf) == 3
    got = list(pf)
    ref = list(loader)
    assert len(got) == len(ref)
    for (gx, gy), (rx, ry) in zip(got, ref):
        assert torch.equal(gx, rx) and torch.equal(gy, ry)
    # nested containers keep their structure
    nested = [{"a": torch.ones(2), "b": (torch.zeros(1), 3)}]
    out = list(DevicePrefetcher(nested, "cpu"))
    assert isinstance(out[0], dict) and isinstance(out[0]["b"], tuple) and out[0]["b"][1] == 3


def test_launcher_refuses_more_ranks_than_gpus_unless_sharing_is_requested(monkeypatch):
    """World size above the visible GPU count is an error (RCCL cannot place two ranks on one
    device); DIST_ALLOW_GPU_SHARING=1 maps ranks round-robin for gloo plumbing runs."""
    from distributed_pytorch_amd.parallel import launcher

    env = launcher.child_device_env(["3", "5"], 1, "rotate")
    assert env["HIP_VISIBLE_DEVICES"] == "5,3" and env["DIST_LOCAL_DEVICE"] == "0"
    assert launcher.child_device_env(["3", "5"], 0, "exclusive")["HIP_VISIBLE_DEVICES"] == "3"
    assert launcher.child_device_env(["3", "5"], 1, "index") == {"DIST_LOCAL_DEVICE": "1"}
    with pytest.raises(ValueError):
        launcher.child_device_env(["0"], 0, "bogus")

    calls = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    monkeypatch.delenv("DIST_ALLOW_GPU_SHARING", raising=False)
    with pytest.raises(ValueError, match="exceeds"):
        launcher.launch(lambda *a: None, nprocs=2)

    def fake_spawn(fn, args, nprocs, env_per_rank, pin_cpus, join, monitor=None, cpu_sets=None):
        calls["env"] = env_per_rank

    monkeypatch.setattr(launcher, "spawn", fake_spawn)
    monkeypatch.setenv("DIST_ALLOW_GPU_SHARING", "1")
    launcher.launch(lambda *a: None, nprocs=3)
    assert [e["HIP_VISIBLE_DEVICES"] for e in calls["env"]] == ["0", "0", "0"]
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["DIST_STORE_HOSTED"] == "1" for e in calls["env"])


def test_profile_summaries_parse(tmp_path):
    from distributed_pytorch_amd.bench import profile

    # counter CSV in rocprofv3's layout -> per-kernel averages and derived MFMA utilisation
    p = tmp_path / "c.csv"
    hdr = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"
    rows = []
    for d in (1, 2):
        for name, val in (("GRBM_GUI_ACTIVE", 8 * 1000.0), ("SQ_VALU_MFMA_BUSY_CYCLES", 512000.0)):
            rows.append(f'{d},{d},"Agent 1",1,1,1,64,1,"dpa::k(int)",256,0,0,64,0,32,{name},{val},{1000 * d},{1000 * d + 500}\n')
    rows.append('3,3,"Agent 1",1,1,1,64,2,"other_kernel",256,0,0,64,0,32,GRBM_GUI_ACTIVE,1.0,0,10\n')
    p.write_text(hdr + "".join(rows))
    out = profile.pmc_summary(str(p), "dpa::")
    assert len(out) == 1 and out[0]["dispatches"] == 2 and out[0]["avg_us"] == 0.5
    assert out[0]["mfma_busy_pct"] == 50.0  # 512000 / (1000 cycles * 1024 SIMDs)


def test_split_k_heuristic_fills_the_last_round_of_tiles():
    """Host-side planning of the bf16 GEMM: few tiles with a long contraction split to fill the chip,
    many tiles split only when the tile count sits just above a multiple of the 256 CUs."""
    from distributed_pytorch_amd import _ext

    if not _ext.have_C():
        pytest.skip("extension not built")
    split = _ext.C().gemm_bf16_split_k
    assert split(768, 3072, 65536, False, False) > 1          # 36 tiles, 1024 k-steps
    assert split(50304, 768, 262144, False, False) == 3       # 591 tiles: 3 rounds -> 7 thirds
    assert split(8192, 8192, 8192, False, False) == 1         # 1024 tiles = 4 full rounds
    assert split(50304, 768, 4096, False, False) == 1         # contraction too short to amortise the reduce
    assert split(65536, 3072, 768, True, True) == 1           # forward shape: many tiles, short K


def test_linear_module_impl_switch_keeps_parameters_and_cpu_path():
    import torch
    from distributed_pytorch_amd.ops.linear import Linear

    torch.manual_seed(0)
    a, b = Linear(16, 8, act="relu"), Linear(16, 8, act="relu", impl="library")
    b.load_state_dict(a.state_dict())
    x = torch.randn(4, 16)
    assert torch.equal(a(x), b(x))
    assert list(a.state_dict()) == ["weight", "bias"] and "impl=library" in repr(b)
    with pytest.raises(ValueError):
        Linear(4, 4, impl="other")


def test_every_binding_mode_gives_each_rank_its_own_gpu(monkeypatch):
    """For each DIST_GPU_BINDING mode, the (visible list, local index) pair a rank ends up with names a
    different physical GPU per rank, and in rotate mode every peer stays visible."""
    from distributed_pytorch_amd.parallel.launcher import child_device_env

    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    ids = [str(i) for i in range(8)]
    for mode in ("rotate", "exclusive", "index"):
        physical = []
        for r in range(8):
            e = child_device_env(ids, r, mode)
            visible = e.get("HIP_VISIBLE_DEVICES", ",".join(ids)).split(",")
            physical.append(visible[int(e["DIST_LOCAL_DEVICE"])])
            if mode == "rotate":
                assert sorted(visible) == ids
            if mode == "exclusive":
                assert visible == [str(r)]
        assert physical == ids, mode


def test_local_device_index_paths(monkeypatch):
    """DIST_LOCAL_DEVICE (our launcher) wins, then LOCAL_RANK (torch.distributed.run), then the rank."""
    import torch

    from distributed_pytorch_amd.parallel import process_group as pg

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("DIST_LOCAL_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert pg.local_device_index(5) == 5
    assert pg.local_device_index(11) == 3            # more ranks than devices wraps (gloo sharing)
    monkeypatch.setenv("LOCAL_RANK", "6")
    assert pg.local_device_index(0) == 6             # torchrun: every rank sees all GPUs, LOCAL_RANK picks
    monkeypatch.setenv("DIST_LOCAL_DEVICE", "0")
    assert pg.local_device_index(6) == 0             # our launcher rotated the visible list


def test_device_uniqueness_check():
    from distributed_pytorch_amd.parallel.process_group import check_device_map, format_device_map

    ok = [(0, "node", "0000:05:00.0"), (1, "node", "0000:15:00.0"), (2, "other", "0000:05:00.0")]
    assert check_device_map(ok) is None              # same bus id on another host is a different GPU
    bad = ok + [(3, "node", "0000:15:00.0")]
    msg = check_device_map(bad)
    assert msg is not None and "ranks 1 and 3" in msg and "0000:15:00.0" in msg
    assert format_device_map(ok).startswith("[rccl] rank->GPU (PCI bus id): 0->0000:05:00.0, 1->0000:15:00.0")


def test_guarded_call_deadline(capsys):
    """Communicator creation runs under a deadline: a call that returns in time passes its value (or its
    exception) through, one that does not gets a line naming what was waited for and the exit hook."""
    import threading
    import time

    from distributed_pytorch_amd.parallel.process_group import guarded_call

    assert guarded_call(lambda: 41 + 1, 5.0, "quick call") == 42
    with pytest.raises(ValueError):
        guarded_call(lambda: (_ for _ in ()).throw(ValueError("boom")), 5.0, "failing call")
    fired = threading.Event()
    guarded_call(lambda: time.sleep(0.6), 0.2, "ncclCommInitRank (rank 0 of 8)", on_timeout=fired.set)
    assert fired.is_set()
    err = capsys.readouterr().err
    assert "ncclCommInitRank (rank 0 of 8) did not return within 0 s" in err and "HSA_ENABLE_IPC_MODE_LEGACY" in err
    fired.clear()
    guarded_call(lambda: None, 0.1, "returns at once", on_timeout=fired.set)
    time.sleep(0.3)
    assert not fired.is_set()  # the timer is disarmed when the call returns


def test_library_banner_is_kept_off_stdout():
    """While a communicator is created, C-level and Python-level writes to stdout land on stderr (RCCL prints a
    version banner there), and stdout works again afterwards: the benchmark's stdout stays one JSON line."""
    import subprocess
    import sys
    from pathlib import Path

    code = (
        "import ctypes\n"
        "from distributed_pytorch_amd.parallel.process_group import _LibraryStdoutToStderr\n"
        "libc = ctypes.CDLL(None)\n"
        "print('before')\n"
        "with _LibraryStdoutToStderr():\n"
        "    libc.printf(b'banner from C\\n')\n"
        "    print('inside python')\n"
        "print('after')\n"
    )
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(Path(__file__).resolve().parent.parent), timeout=120)
    assert p.returncode == 0, p.stderr[-1000:]
    assert p.stdout == "before\nafter\n"
    assert "banner from C" in p.stderr and "inside python" in p.stderr


def test_launcher_leaves_ipc_mode_alone(monkeypatch):
    """Runtime variables that change how peer mappings are opened reach the workers as exported,
    never invented by the launcher."""
    import distributed_pytorch_amd.parallel.launcher as L

    src = inspect.getsource(L.launch)
    assert '"HSA_ENABLE_IPC_MODE_LEGACY":' not in src


def test_cross_entropy_target_check_is_opt_in():
    import torch

    from distributed_pytorch_amd.ops import SoftmaxCrossEntropy, softmax_cross_entropy

    logits = torch.randn(4, 5)
    bad = torch.tensor([0, 5, 1, 2])
    with pytest.raises(IndexError, match="Target 5 is out of bounds"):
        softmax_cross_entropy(logits, bad, check_targets=True)
    with pytest.raises(IndexError):
        SoftmaxCrossEntropy(check_targets=True)(logits, torch.tensor([0, -3, 1, 2]))
    ok = torch.tensor([0, 4, -100, 2])           # ignore_index is not an error
    ref = torch.nn.functional.cross_entropy(logits, ok)
    torch.testing.assert_close(softmax_cross_entropy(logits, ok, check_targets=True).loss, ref)


def test_numa_local_cpu_slices_from_sysfs(tmp_path):
    """CPU affinity follows the GPU's NUMA node: a fake sysfs with one CPU node, four GPUs on two NUMA
    nodes; ranks on a node split that node's allowed CPUs, unknown topology falls back (None)."""
    from distributed_pytorch_amd.parallel.launcher import gpu_numa_nodes, numa_cpu_slices, topology_cpu_slices, _parse_cpulist

    root = tmp_path
    nodes = root / "sys/class/kfd/kfd/topology/nodes"
    layout = [(0, 0, None), (1, 64, (0, 0x05, 0)), (2, 64, (0, 0x15, 0)), (3, 64, (0, 0x85, 1)), (4, 64, (0, 0x95, 1))]
    for nid, simd, pci in layout:
        d = nodes / str(nid)
        d.mkdir(parents=True)
        loc = 0 if pci is None else (pci[1] << 8)
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\ndomain 0\nlocation_id {loc}\n")
        if pci is not None:
            p = root / f"sys/bus/pci/devices/0000:{pci[1]:02x}:00.0"
            p.mkdir(parents=True)
            (p / "numa_node").write_text(f"{pci[2]}\n")
    for n, cpus in ((0, "0-3"), (1, "4-7")):
        p = root / f"sys/devices/system/node/node{n}"
        p.mkdir(parents=True)
        (p / "cpulist").write_text(cpus + "\n")
    assert gpu_numa_nodes(str(root)) == [0, 0, 1, 1]
    assert _parse_cpulist("0-2,5,7-8") == [0, 1, 2, 5, 7, 8]
    s = numa_cpu_slices([0, 1, 2, 3], [0, 0, 1, 1], {0: [0, 1, 2, 3], 1: [4, 5, 6, 7]}, range(8))
    assert s == [[0, 1], [2, 3], [4, 5], [6, 7]]
    # rotated rank -> GPU assignment, restricted affinity mask
    s = numa_cpu_slices([3, 0], [0, 0, 1, 1], {0: [0, 1, 2, 3], 1: [4, 5, 6, 7]}, [1, 2, 3, 6])
    assert s == [[6], [1, 2, 3]]
    assert numa_cpu_slices([0, 5], [0, 0, 1, 1], {0: [0], 1: [4]}, range(8)) is None      # GPU index without topology
    assert numa_cpu_slices([0, 1], [0, 0], {0: [0]}, range(8)) is None                    # fewer CPUs than ranks
    assert gpu_numa_nodes(str(tmp_path / "nowhere")) == []
    import os
    if set(range(8)) <= os.sched_getaffinity(0):
        assert topology_cpu_slices(["2", "0"], str(root)) == [[4, 5, 6, 7], [0, 1, 2, 3]]
    assert topology_cpu_slices(["GPU-abc"], str(root)) is None
