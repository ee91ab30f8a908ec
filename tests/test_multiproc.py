"""Multi-process CPU tier: own launcher + own TCP store + gloo data plane, world size 2-3.
Covers the collective semantics of the facade, gradient equality of the native reducer
against torch's DDP, the min_DDP golden numbers, checkpoint layout, and failure propagation."""
import json
import os
import re
import signal
import subprocess
import sys
import time
from pathlib import Path

import pytest

import distributed as dist
from distributed_pytorch_amd.parallel.launcher import ProcessExitedException, ProcessRaisedException

sys.path.insert(0, str(Path(__file__).resolve().parent))
import mp_workers  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent


def _read(out_dir, world):
    return [json.loads((Path(out_dir) / f"rank{r}.json").read_text()) for r in range(world)]


def test_collectives_world2(tmp_path):
    dist.launch(mp_workers.collectives_worker, str(tmp_path), nprocs=2)
    r0, r1 = _read(tmp_path, 2)
    assert (r0["rank"], r1["rank"]) == (0, 1) and r0["world"] == r1["world"] == 2
    assert r0["primary"] and not r1["primary"] and r0["initialised"] and r0["device"] == "cpu"
    # reduce: SUM onto rank 0, same tensor object returned everywhere
    assert r0["reduce"] == [3.0, 30.0] and r0["reduce_same_object"] and r1["reduce_same_object"]
    assert r0["reduce_max"] == [3.0]
    assert r0["all_reduce_sum"] == r1["all_reduce_sum"] == [3.0]
    assert r0["all_reduce_avg"] == r1["all_reduce_avg"] == [1.5]
    assert r0["bad_op"] == r1["bad_op"] == '"max" is an invalid reduce operation!'
    # gather: rank order, real data on the primary only, zeros elsewhere; bool works
    assert r0["gather"] == [[0, 0], [1, 2]] and r1["gather"] == [[0, 0], [0, 0]]
    assert r0["gather_bool"] == [[True, True], [False, True]] and r0["gather_bool_dtype"] == "torch.bool"
    assert r1["gather_bool"] == [[False, False], [False, False]]
    assert r0["sync_params"] == r1["sync_params"] == [0.0, 0.0, 0.0]
    assert r0["sampler"] == [0, 2, 4, 6, 8] and r1["sampler"] == [1, 3, 5, 7, 9]
    # all_gather: every rank gets every rank's tensor; reduce_scatter: rank r gets rows r*n .. of the reduced tensor
    assert r0["all_gather"] == r1["all_gather"] == [[0, 10], [1, 11]]
    assert r0["reduce_scatter_sum"] == [0.0, 3.0] and r1["reduce_scatter_sum"] == [6.0, 9.0]  # arange(4) * (1 + 2)
    assert r0["reduce_scatter_avg"] == [0.0, 1.5] and r1["reduce_scatter_avg"] == [3.0, 4.5]
    assert r0["reduce_scatter_input_kept"] and r1["reduce_scatter_input_kept"]
    assert "not divisible by the world size 2" in r0["reduce_scatter_bad"]


def test_collectives_world3_sampler_padding(tmp_path):
    dist.launch(mp_workers.collectives_worker, str(tmp_path), nprocs=3)
    rs = _read(tmp_path, 3)
    assert rs[0]["reduce"] == [6.0, 60.0]
    assert rs[0]["gather"] == [[0, 0], [1, 2], [2, 4]]
    assert [r["reduce_scatter_sum"] for r in rs] == [[0.0, 6.0], [12.0, 18.0], [24.0, 30.0]]  # arange(6) * (1 + 2 + 3)
    assert [r["sampler"] for r in rs] == [[0, 3, 6, 9], [1, 4, 7, 0], [2, 5, 8, 1]]  # padded with the head


def test_native_reducer_matches_torch_ddp(tmp_path):
    dist.launch(mp_workers.ddp_vs_torch_worker, str(tmp_path), nprocs=2)
    r0, r1 = _read(tmp_path, 2)
    for r in (r0, r1):
        assert r["n_buckets"] >= 3
        assert r["keys"][0].startswith("module.")
        assert r["max_param_diff"] < 1e-6
        assert r["bn_running_mean_diff"] < 1e-6
        assert r["in_sync"]
        assert r["unused_grad_is_zero"]
    assert r0["checksum"] == r1["checksum"]
    # c.bias grad of sum(): 4 rows -> 4 per output locally; accumulated twice, then averaged over ranks
    assert r0["no_sync_local_grad"] == [4.0, 4.0, 4.0]
    assert r0["no_sync_final_grad"] == r1["no_sync_final_grad"] == [8.0, 8.0, 8.0]


def test_launch_forwards_extra_args_and_environment(tmp_path):
    dist.launch(mp_workers.extra_args_worker, str(tmp_path), "alpha", 42, nprocs=2)
    r0, r1 = _read(tmp_path, 2)
    assert (r0["a"], r0["b"], r1["a"], r1["b"]) == ("alpha", 42, "alpha", 42)
    assert (r0["rank"], r1["rank"], r0["world"]) == (0, 1, 2)
    assert (r0["env_rank"], r1["env_rank"], r0["env_world"]) == ("0", "1", "2")
    assert r0["master_port"] == r1["master_port"] and r0["hosted"] == "1"
    assert not set(r0["affinity"]) & set(r1["affinity"])  # disjoint CPU slices


def test_worker_exception_propagates_and_siblings_die(tmp_path):
    t0 = time.time()
    with pytest.raises(ProcessRaisedException) as ei:
        dist.launch(mp_workers.crash_worker, str(tmp_path), nprocs=2)
    assert time.time() - t0 < 40
    assert ei.value.rank == 1 and "boom from rank 1" in str(ei.value) and "Traceback" in str(ei.value)
    for r in range(2):
        pid = int((tmp_path / f"pid{r}").read_text())
        deadline = time.time() + 15  # the teardown is asynchronous to the exception on a loaded machine
        while True:
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break  # no orphans
            assert time.time() < deadline, f"rank {r} (pid {pid}) is still alive"
            time.sleep(0.1)


def test_worker_hard_exit_is_reported(tmp_path):
    with pytest.raises(ProcessExitedException) as ei:
        dist.launch(mp_workers.exit_code_worker, str(tmp_path), nprocs=2)
    assert ei.value.rank == 0 and ei.value.exit_code == 7


def test_checkpoint_keeps_module_prefix_layout(tmp_path):
    dist.launch(mp_workers.checkpoint_worker, str(tmp_path), nprocs=2)
    for r in _read(tmp_path, 2):
        assert r["restored"] and r["step"] == 5 and r["note"] == "hi"
        assert r["keys"] == ["module.weight", "module.bias"]
    # a multi-rank checkpoint loads into an unwrapped single-process model
    import torch

    from distributed_pytorch_amd.utils.checkpoint import load_checkpoint

    m = torch.nn.Linear(4, 2)
    meta = load_checkpoint(str(tmp_path / "ckpt.pt"), m)
    assert meta["world_size"] == 2 and list(m.state_dict()) == ["weight", "bias"]


def _run_example(args, env_extra, timeout=180):
    env = dict(os.environ)
    env.update(env_extra)
    env["PYTHONPATH"] = str(ROOT) + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, str(ROOT / "min_DDP.py"), *args], cwd=str(ROOT), env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_min_ddp_world2_matches_reference_goldens():
    """Numbers measured with the reference code (2 ranks, gloo, torch.manual_seed(0), defaults):
    iteration 0 -> loss 6.2573 (SUM over ranks), acc 0.1875 (3/16); iteration 1 -> 14.9106, 0.3750 (6/16)."""
    p = _run_example(["--seed", "0"], {"DIST_WORLD_SIZE": "2"})
    assert p.returncode == 0, p.stdout[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("Finish iteration")]
    assert len(lines) == 4  # 2 iterations per epoch at world 2, 2 epochs
    assert lines[0] == "Finish iteration 0 - acc: 0.1875 (3/16) - loss: 6.2573"
    assert lines[1] == "Finish iteration 1 - acc: 0.3750 (6/16) - loss: 14.9106"
    assert p.stdout.count("Run epochs") == 2  # un-gated print on every rank, as in the reference
    # shard membership is visible: rank 0 sees even inputs, rank 1 odd ones
    assert re.search(r"Input: \ttensor\(\[ ?0,  ?2,  ?4", p.stdout) and re.search(r"Input: \ttensor\(\[ ?1,  ?3,  ?5", p.stdout)


def test_min_ddp_single_process_and_stock_path_agree():
    a = _run_example(["--seed", "0", "--epochs", "1"], {"DIST_WORLD_SIZE": ""})
    b = _run_example(["--seed", "0", "--epochs", "1", "--no-native"], {"DIST_WORLD_SIZE": ""})
    assert a.returncode == 0 and b.returncode == 0, a.stdout[-1500:] + b.stdout[-1500:]
    fa = [l for l in a.stdout.splitlines() if l.startswith("Finish iteration")]
    fb = [l for l in b.stdout.splitlines() if l.startswith("Finish iteration")]
    assert len(fa) == 4 and fa == fb  # 4 iterations per epoch at world 1; native CPU path == stock torch


def test_min_ddp_quiet_mode_world2():
    p = _run_example(["--seed", "0", "--quiet"], {"DIST_WORLD_SIZE": "2"})
    assert p.returncode == 0, p.stdout[-2000:]
    assert "Finish iteration" not in p.stdout
    assert p.stdout.count("Epoch summary") == 2 and "(9/32)" in p.stdout  # 3 + 6 correct of 32 in epoch 1


def test_sigterm_on_launcher_kills_workers(tmp_path):
    """Interrupting the parent must not leave workers (or the rendezvous port) behind."""
    script = tmp_path / "run.py"
    script.write_text(
        "import sys, time, os\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})\n"
        "import distributed as dist\n"
        "def sleepy(rank, world, out):\n"
        "    open(os.path.join(out, f'pid{rank}'), 'w').write(str(os.getpid()))\n"
        "    time.sleep(120)\n"
        "if __name__ == '__main__':\n"
        f"    dist.launch(sleepy, {str(tmp_path)!r}, nprocs=2)\n")
    proc = subprocess.Popen([sys.executable, str(script)], cwd=str(tmp_path))
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / f"pid{r}").exists() for r in range(2)):
        time.sleep(0.1)
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in range(2)]
    proc.send_signal(signal.SIGTERM)
    proc.wait(timeout=30)
    for pid in pids:
        deadline = time.time() + 15
        while True:
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            assert time.time() < deadline, f"worker {pid} outlived the launcher"
            time.sleep(0.1)


def test_bench_under_torch_distributed_run_two_ranks():
    """The launch line of the scaling benchmark: torch.distributed.run owns MASTER_PORT with its own
    store, the ranks rendezvous through it, rank 0 prints exactly one JSON line."""
    import json
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "dummy",
           "--batch-size", "256", "--backend", "gloo"]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["steps"] == 2 and rec["warmup"] == 1 and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["global_batch"] == 512 and rec["config"]["comm"] == "gloo" and rec["value"] > 0
    assert rec["replicas_in_sync"] is True  # parameter checksums of the ranks compared after the timed steps


def test_example_training_script_two_ranks_checkpoint_and_resume(tmp_path):
    """examples/train_gpt2.py end to end on two gloo ranks: sampler, prefetcher, DDP, fused optimizer,
    logging, checkpoint; a second invocation resumes from the saved step."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ckpt = str(tmp_path / "gpt2.pt")
    base = [sys.executable, os.path.join(root, "examples", "train_gpt2.py"), "--n-layer", "2", "--n-embd", "64", "--n-head", "2",
            "--seq-len", "32", "--vocab-size", "512", "--batch-size", "8", "--data-size", "256", "--log-every", "4", "--ckpt", ckpt]
    env = dict(os.environ, DIST_WORLD_SIZE="2", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    first = subprocess.run(base + ["--steps", "8"], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert first.returncode == 0, first.stderr[-2000:]
    recs = [json.loads(ln) for ln in first.stdout.splitlines() if ln.startswith("{")]
    assert [r["step"] for r in recs] == [4, 8] and all(r["loss"] > 0 for r in recs)
    assert "saved" in first.stdout and os.path.exists(ckpt)
    second = subprocess.run(base + ["--steps", "12", "--resume"], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert second.returncode == 0, second.stderr[-2000:]
    assert "resumed from" in second.stdout and "at step 8" in second.stdout
    recs2 = [json.loads(ln) for ln in second.stdout.splitlines() if ln.startswith("{")]
    assert [r["step"] for r in recs2] == [12]
    # the checkpoint of the full optimizer continues under the sharded one, whose checkpoint (unsharded layout,
    # assembled from both ranks' shards) continues under the full one again
    third = subprocess.run(base + ["--steps", "16", "--resume", "--shard-optimizer"], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert third.returncode == 0, third.stderr[-2000:]
    assert "at step 12" in third.stdout and "saved" in third.stdout
    import torch

    saved = torch.load(ckpt, weights_only=True)
    n_params = len([k for k in saved["model"] if not k.endswith("lm_head.weight")])
    assert saved["step"] == 16 and len(saved["optimizer"]["state"]) == n_params  # state of every parameter, from both shards
    fourth = subprocess.run(base + ["--steps", "20", "--resume"], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert fourth.returncode == 0, fourth.stderr[-2000:]
    recs4 = [json.loads(ln) for ln in fourth.stdout.splitlines() if ln.startswith("{")]
    assert "at step 16" in fourth.stdout and [r["step"] for r in recs4] == [20]
    assert recs4[-1]["loss"] < recs[0]["loss"]


def test_bucket_rebuild_slicing_and_positional_args(tmp_path):
    dist.launch(mp_workers.ddp_rebuild_worker, str(tmp_path), nprocs=2)
    r0, r1 = _read(tmp_path, 2)
    for r in (r0, r1):
        assert r["cap_bytes"] == int(0.001 * (1 << 20))          # bucket_cap_mb arrived positionally
        assert r["rebuilt"] and r["buckets_after"] != r["buckets_before"]
        flat = [i for b in r["buckets_after"] for i in b]
        assert flat == r["ready_first_iteration"]                 # bucket k holds the k-th group of ready gradients
        # ... so every bucket goes out the moment its last gradient arrives
        sizes, total = [], 0
        for b in r["buckets_after"]:
            total += len(b)
            sizes.append(total)
        assert r["launch_positions"] == sizes
        assert max(r["slices"]) > 1                               # the 16x64 layer exceeds the collective cap
        assert r["grads_are_views"]
        assert r["max_param_diff"] < 1e-6
        assert r["skipped_got"] == pytest.approx(r["skipped_expected"], rel=1e-6)
    assert r0["buckets_after"] == r1["buckets_after"]


def test_hung_rank_is_named_and_job_is_killed(tmp_path, monkeypatch):
    from distributed_pytorch_amd.parallel.launcher import ProcessHungException

    monkeypatch.setenv("DIST_HANG_TIMEOUT_S", "3")
    monkeypatch.setenv("DIST_HEARTBEAT_S", "0.2")
    t0 = time.time()
    with pytest.raises(ProcessHungException) as ei:
        dist.launch(mp_workers.hang_worker, str(tmp_path), nprocs=2)
    took = time.time() - t0
    assert ei.value.ranks == [1]
    assert "rank 1 (collective #1)" in str(ei.value) and "#2" in str(ei.value)
    assert took < 60, took      # the sleeping rank would have held the job for 120 s


def test_hang_monitor_decisions():
    from distributed_pytorch_amd.parallel.heartbeat import HangMonitor

    now = [0.0]
    mon = HangMonitor(None, 3, timeout_s=10, stall_s=0, clock=lambda: now[0])
    assert mon.observe({0: (1, "run"), 1: (1, "run")}) is None                      # rank 2 has not joined yet
    assert mon.observe({0: (5, "run"), 1: (5, "run"), 2: (5, "run")}) is None
    now[0] = 100.0
    assert mon.observe({0: (5, "run"), 1: (5, "run"), 2: (5, "run")}) is None       # equal counters: idle, not hung
    assert mon.observe({0: (6, "run"), 1: (5, "run"), 2: (6, "run")}) is not None   # 1 has been at #5 for 100 s
    mon2 = HangMonitor(None, 2, timeout_s=10, stall_s=0, clock=lambda: now[0])
    assert mon2.observe({0: (1, "run"), 1: (1, "run")}) is None
    now[0] += 5
    assert mon2.observe({0: (2, "run"), 1: (1, "run")}) is None                     # behind, but only for 5 s
    now[0] += 6
    v = mon2.observe({0: (2, "run"), 1: (1, "run")})
    assert v is not None and v.ranks == [1] and v.kind == "lag"
    mon3 = HangMonitor(None, 2, timeout_s=10, stall_s=0, clock=lambda: now[0])
    mon3.observe({0: (3, "run"), 1: (9, "done")})
    now[0] += 100
    assert mon3.observe({0: (3, "run"), 1: (9, "done")}) is None                    # a finished rank is not waited for
    mon4 = HangMonitor(None, 2, timeout_s=0, stall_s=20, clock=lambda: now[0])
    mon4.observe({0: (4, "run"), 1: (4, "run")})
    now[0] += 21
    v = mon4.observe({0: (4, "run"), 1: (4, "run")})
    assert v is not None and v.kind == "stall" and v.ranks == [0, 1]


def test_find_unused_parameters_keeps_globally_unused_grads_none(tmp_path):
    dist.launch(mp_workers.ddp_unused_worker, str(tmp_path), nprocs=2)
    r0, r1 = _read(tmp_path, 2)
    for r in (r0, r1):
        assert r["never_grad_is_none"] and r["never_untouched"] and r["in_sync"]
        assert r["globally_unused"] == [4, 5]                      # never.weight, never.bias
        # rank 0 contributes ones * 2 rows = 2 per output, rank 1 zeros: average 1
        assert r["rank0_only_bias_grad"] == [1.0, 1.0, 1.0, 1.0]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_matches_full_optimizer(tmp_path, world):
    dist.launch(mp_workers.sharded_optimizer_worker, str(tmp_path), nprocs=world)
    rs = _read(tmp_path, world)
    for r in rs:
        assert r["owner"] == rs[0]["owner"] and sorted(set(r["owner"])) == list(range(world))
        assert max(r["load"]) < 0.75 * sum(r["load"])              # no rank owns (nearly) everything
        assert False in r["rehomed"] and True in r["rehomed"]      # the channels-last weight is copied, the rest are flat views
        assert r["state_entries"] == r["owned"] < 6                # optimizer state exists for owned parameters only
        assert r["diff_after_3"] == 0.0 and r["diff_after_5"] == 0.0 and r["diff_after_resume"] == 0.0
        assert r["grad_norm_equal"] and r["in_sync"]
        assert r["cons_keys_equal"] and r["cons_state_diff"] == 0.0 and r["cons_groups_equal"]
        assert r["lr_restored"] == [5e-3, 5e-3]
    assert sum(r["owned"] for r in rs) == 6


def test_ddp_keyword_and_method_surface_world2(tmp_path):
    dist.launch(mp_workers.ddp_surface_worker, str(tmp_path), nprocs=2)
    for r in _read(tmp_path, 2):
        assert r["warned"] == ["mixed_precision", "static_graph"] or r["warned"] == [], r["warned"]  # warned once per process
        assert r["in_sync_after_training"] and r["in_sync_after_compressed_step"]
        assert r["bucket_dtype_after_hook"] == "torch.float16" and r["bucket_dtype_restored"] == "torch.float32"
        assert r["custom_hook"] and "register_comm_hook" in r["custom_hook"]
        assert r["hashes_say_in_sync"] is False and r["diverged"] == ["0.weight"]
        assert r["join_even"] is True
    assert _read(tmp_path, 2)[0]["warned"] == ["mixed_precision", "static_graph"]


@pytest.mark.parametrize("world,clip", [(2, None), (3, None), (2, 0.5)])
def test_range_sharded_optimizer_on_reduce_scatter_matches_full_optimizer(tmp_path, world, clip):
    dist.launch(mp_workers.sharded_arena_worker, str(tmp_path), clip, nprocs=world)
    # Two ranks, no clipping: the update is the full optimizer's, element for element -- bit-equal.  With clipping the
    # global norm is summed range by range and over the ranks instead of parameter by parameter: equal to the last
    # bits.  Three ranks: the padded buckets are cut into other ring segments than the plain wrapper's, so the
    # three-term sums round differently here and there, and Adam's first steps (update ~ sign of the gradient)
    # turn a last-bit difference of a near-zero gradient into a visible one.
    tol = 0.0 if (clip is None and world == 2) else (1e-6 if world == 2 else 2e-4)
    for r in _read(tmp_path, world):
        assert r["arena"] and r["n_buckets"] >= 2 and r["params_are_views"]
        # fp32 moments exist for one range of every bucket: the model over the ranks, plus padding
        assert r["model_numel"] / world <= r["state_numel"] <= r["model_numel"] / world + 256 * r["n_buckets"]
        assert r["diff_after_3"] <= tol and r["diff_after_5"] <= tol and r["diff_after_resume"] <= tol
        assert r["grad_norm_diff"] < 1e-6 and r["in_sync"]
        assert r["cons_keys_equal"] and r["cons_state_diff"] <= tol
        assert r["lr_restored"] == [5e-3]
        # per step: one reduce-scatter and one all-gather per bucket (the plain wrapper: one all-reduce per bucket)
        assert r["collectives_per_step"] == 2 * r["n_buckets"]
