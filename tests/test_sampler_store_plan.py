"""CPU unit tests: sampler index sequences against torch's DistributedSampler, the native
TCP store, and the bucket planner."""
import threading
import time

import pytest
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from distributed_pytorch_amd import _ext
from distributed_pytorch_amd.parallel.sampler import ShardedSampler
from distributed_pytorch_amd.parallel.store import Store


# ----------------------------------------------------------------------------- sampler
@pytest.mark.parametrize("n", [1, 7, 32, 33, 100])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("drop_last", [False, True])
def test_sampler_matches_torch_distributed_sampler(n, world, shuffle, drop_last):
    ds = list(range(n))
    for rank in range(world):
        ours = ShardedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=3, drop_last=drop_last)
        ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=3, drop_last=drop_last)
        for epoch in (0, 1, 5):
            ours.set_epoch(epoch)
            ref.set_epoch(epoch)
            assert list(ours) == list(ref)
            assert len(ours) == len(ref)


def test_sampler_shards_are_rank_strided_and_work_with_dataloader():
    ds = list(range(32))
    r0 = ShardedSampler(ds, num_replicas=2, rank=0, shuffle=False)
    r1 = ShardedSampler(ds, num_replicas=2, rank=1, shuffle=False)
    assert list(r0) == list(range(0, 32, 2)) and list(r1) == list(range(1, 32, 2))
    batches = [b.tolist() for b in DataLoader(torch.arange(32), batch_size=8, sampler=r1)]
    assert batches[0] == [1, 3, 5, 7, 9, 11, 13, 15] and len(batches) == 2
    with pytest.raises(ValueError):
        ShardedSampler(ds, num_replicas=2, rank=2)


def test_sampler_padding_repeats_head():
    s = ShardedSampler(list(range(5)), num_replicas=4, rank=3, shuffle=False)
    assert s.global_indices() == [0, 1, 2, 3, 4, 0, 1, 2] and list(s) == [3, 2]


# ----------------------------------------------------------------------------- store
def test_store_set_get_add_check_delete_compare():
    srv = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
    cli = Store("127.0.0.1", srv.port, timeout=5)
    srv.set("k", b"\x00\x01binary\xff")
    assert cli.get("k") == b"\x00\x01binary\xff"
    cli.set("s", "text")
    assert srv.get("s") == b"text"
    assert cli.add("n", 5) == 5 and srv.add("n", -2) == 3 and cli.get("n") == b"3"
    assert cli.check(["k", "n"]) and not cli.check(["k", "missing"])
    assert cli.num_keys() == 3
    assert cli.compare_set("c", "", "v1") == b"v1"
    assert cli.compare_set("c", "nope", "v2") == b"v1"
    assert cli.compare_set("c", "v1", "v2") == b"v2"
    cli.append("a", b"ab")
    cli.append("a", b"cd")
    assert cli.get("a") == b"abcd"
    assert cli.delete_key("k") and not cli.delete_key("k")
    big = bytes(range(256)) * 4096  # 1 MiB value crosses many recv() calls
    cli.set("big", big)
    assert srv.get("big") == big
    cli.close()
    srv.close()


def test_store_blocking_get_wait_timeout_and_barrier():
    srv = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
    cli = Store("127.0.0.1", srv.port, timeout=5)
    threading.Timer(0.2, lambda: Store("127.0.0.1", srv.port).set("late", b"1")).start()
    t0 = time.time()
    assert cli.get("late") == b"1"
    assert 0.1 < time.time() - t0 < 4.5  # blocked until the late writer came, well inside the 5 s timeout
    with pytest.raises(RuntimeError, match="timed out"):
        cli.get("never", timeout=0.2)
    assert cli.add("after-timeout", 1) == 1  # the client reconnects after an expired wait
    with pytest.raises(RuntimeError, match="timed out"):
        cli.wait(["late", "never"], timeout=0.2)
    # reusable counter barrier across 4 threads, 3 rounds
    order = []

    def worker(i):
        s = Store("127.0.0.1", srv.port, timeout=10)
        for r in range(3):
            time.sleep(0.01 * i)
            s.barrier("b", 4)
            order.append((r, i))
        s.close()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert [r for r, _ in order] == sorted(r for r, _ in order)  # nobody enters round r+1 before all left round r
    cli.close()
    srv.close()


def test_store_port_zero_gives_distinct_ports_and_port_is_released():
    a = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
    b = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
    assert a.port != b.port
    port = a.port
    a.close()
    c = Store("127.0.0.1", port, is_host=True, bind_addr="127.0.0.1")  # rebinding right away works
    assert c.port == port
    b.close()
    c.close()


def test_torch_store_adapter_bootstraps_gloo_single_rank():
    import torch.distributed as tdist

    from distributed_pytorch_amd.parallel import process_group as pg

    srv = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
    pg.init_process_group(0, 1, "gloo", store=srv)
    try:
        assert pg.is_initialized() and pg.get_world_size() == 1 and pg.get_backend() == "gloo"
        assert tdist.is_initialized()
        t = torch.ones(3)
        assert pg.all_reduce(t) is t and pg.gather(t)[0] is t
    finally:
        pg.destroy_process_group()
    assert not pg.is_initialized() and not tdist.is_initialized()


# ----------------------------------------------------------------------------- bucket planner
def _plan(sizes, first, cap, align=128, elem=4, keys=None):
    h = _ext.host()
    metas = [h.ParamMeta(n, elem, (keys[i] if keys else 0), True) for i, n in enumerate(sizes)]
    return h.plan_buckets(metas, h.PlanOptions(first, cap, align, True))


def test_plan_reverse_order_first_bucket_small_then_cap():
    sizes = [1000] * 10  # 4000 B each, padded to 4096
    p = _plan(sizes, first=8192, cap=16384)
    assert [list(b.param_indices) for b in p.buckets] == [[9, 8], [7, 6, 5, 4], [3, 2, 1, 0]]
    assert all(o % 32 == 0 for b in p.buckets for o in b.offsets)  # 128-byte aligned slices
    assert [b.arena_offset for b in p.buckets] == [0, 2048, 2048 + 4096]
    assert p.arena_numel == [(0, 2048 + 4096 + 4096)]


def test_plan_covers_every_param_once_and_splits_dtypes():
    sizes = [3, 70000, 1, 513, 100000, 7]
    keys = [0, 0, 1, 1, 0, 1]
    p = _plan(sizes, first=1 << 20, cap=1 << 20, keys=keys)
    seen = sorted(i for b in p.buckets for i in b.param_indices)
    assert seen == list(range(6))
    for b in p.buckets:
        assert {keys[i] for i in b.param_indices} == {b.group_key}
        for i, o in zip(b.param_indices, b.offsets):
            assert o + sizes[i] <= b.numel
        spans = sorted((o, o + sizes[i]) for i, o in zip(b.param_indices, b.offsets))
        assert all(spans[k][1] <= spans[k + 1][0] for k in range(len(spans) - 1))  # no overlap


def test_plan_skips_frozen_params_and_replans_from_ready_order():
    h = _ext.host()
    metas = [h.ParamMeta(100, 4, 0, True), h.ParamMeta(100, 4, 0, False), h.ParamMeta(100, 4, 0, True)]
    p = h.plan_buckets(metas, h.PlanOptions(1 << 20, 1 << 20, 128, True))
    assert [list(b.param_indices) for b in p.buckets] == [[2, 0]]
    q = h.plan_buckets_from_order(metas, [0, 2], h.PlanOptions(1 << 20, 1 << 20, 128, True))
    assert [list(b.param_indices) for b in q.buckets] == [[0, 2]]
    with pytest.raises(Exception):
        h.plan_buckets(metas, h.PlanOptions(0, 1, 128, True))
