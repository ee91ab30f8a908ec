"""Numerics of every HIP kernel against plain PyTorch references (fp32 / fp64 math on the
same inputs).  Shape grids include the reference model's degenerate layers (K = 1, N = 4),
sizes that are not multiples of any tile, single rows, and extreme logits."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def C():
    from distributed_pytorch_amd import _ext

    return _ext.C()


def tol(dtype):
    return dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)


# ----------------------------------------------------------------------------- pack
@pytest.mark.parametrize("src_dt,dst_dt", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                            (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
def test_multi_copy_scale(src_dt, dst_dt):
    torch.manual_seed(0)
    sizes = [1, 7, 8, 63, 4096, 8192, 8193, 100003] + [33] * 40  # > 32 tensors: several launches
    srcs = [torch.randn(n, device=DEV).to(src_dt) for n in sizes]
    flat = torch.full((sum((n + 31) // 32 * 32 for n in sizes),), 7.0, device=DEV, dtype=dst_dt)
    dsts, off = [], 0
    for n in sizes:
        dsts.append(flat.narrow(0, off, n))
        off += (n + 31) // 32 * 32
    C().multi_copy_scale(srcs, dsts, 0.25)
    for s, d in zip(srcs, dsts):
        ref = (s.float() * 0.25).to(dst_dt)
        assert torch.equal(d, ref)
    # padding between tensors is untouched
    assert flat[sizes[0]].item() == 7.0


def test_multi_copy_unaligned_views():
    base = torch.randn(10001, device=DEV)
    out = torch.zeros(10001, device=DEV)
    C().multi_copy_scale([base[1:5000], base[5003:]], [out[3:5002], out[5003:]], 2.0)
    assert torch.equal(out[3:5002], base[1:5000] * 2)
    assert torch.equal(out[5003:], base[5003:] * 2)
    assert out[5002].item() == 0.0 and out[:3].abs().sum().item() == 0.0


def test_scale_inplace():
    for dt in (torch.float32, torch.bfloat16):
        t = torch.randn(100001, device=DEV).to(dt)
        ref = (t.float() * 0.125).to(dt)
        C().scale_inplace(t, 0.125)
        assert torch.equal(t, ref)


# ----------------------------------------------------------------------------- AdamW
@pytest.mark.parametrize("wd", [0.01, 0.0])
def test_fused_adamw_matches_torch_fp32(wd):
    from distributed_pytorch_amd.ops import FusedAdamW

    torch.manual_seed(1)
    shapes = [(4096, 1), (4096,), (4, 4096), (4,), (257, 33), (1,)] + [(17,)] * 30
    ref_p = [torch.randn(s, dtype=torch.float64) for s in shapes]
    my_p = [p.clone().float().to(DEV).requires_grad_() for p in ref_p]
    ref_p = [p.requires_grad_() for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=1e-2, weight_decay=wd)
    mine = FusedAdamW(my_p, lr=1e-2, weight_decay=wd)
    for step in range(12):
        for rp, mp in zip(ref_p, my_p):
            g = torch.randn(rp.shape, dtype=torch.float64)
            rp.grad = g
            mp.grad = g.float().to(DEV)
        ref.step()
        mine.step()
    for rp, mp in zip(ref_p, my_p):
        torch.testing.assert_close(mp.detach().cpu().double(), rp.detach(), rtol=2e-5, atol=2e-6)


def test_fused_adamw_bf16_master_weights_and_lr_change():
    from distributed_pytorch_amd.ops import FusedAdamW

    torch.manual_seed(2)
    w0 = torch.randn(1000, 64)
    ref_p = w0.clone().double().requires_grad_()
    my_p = w0.clone().to(DEV).bfloat16().requires_grad_()
    ref_p.data = my_p.detach().cpu().double()  # same starting point after bf16 rounding
    ref = torch.optim.AdamW([ref_p], lr=1e-3)
    mine = FusedAdamW([my_p], lr=1e-3, master_weights=True)
    for step in range(8):
        if step == 4:  # learning-rate schedule must reach the device-side hyper-parameters
            for o in (ref, mine):
                o.param_groups[0]["lr"] = 5e-4
        g = torch.randn(1000, 64).bfloat16()
        ref_p.grad = g.double()
        my_p.grad = g.to(DEV)
        ref.step()
        mine.step()
    master = mine.state[my_p]["master"]
    torch.testing.assert_close(master.cpu().double(), ref_p.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(my_p.detach().cpu().double(), ref_p.detach(), rtol=1e-2, atol=1e-2)


def test_fused_adamw_clipping():
    from distributed_pytorch_amd.ops import FusedAdamW

    torch.manual_seed(3)
    ref_p = [torch.randn(300, 7, dtype=torch.float64).requires_grad_(), torch.randn(50, dtype=torch.float64).requires_grad_()]
    my_p = [p.detach().clone().float().to(DEV).requires_grad_() for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=1e-2)
    mine = FusedAdamW(my_p, lr=1e-2, max_grad_norm=1.0)
    for _ in range(3):
        for rp, mp in zip(ref_p, my_p):
            g = torch.randn(rp.shape, dtype=torch.float64) * 3
            rp.grad = g
            mp.grad = g.float().to(DEV)
        n = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
        ref.step()
        mine.step()
        torch.testing.assert_close(mine.last_grad_norm.cpu().double().squeeze(), n, rtol=1e-5, atol=1e-6)
    for rp, mp in zip(ref_p, my_p):
        torch.testing.assert_close(mp.detach().cpu().double(), rp.detach(), rtol=2e-5, atol=2e-6)


def test_fused_adamw_late_parameter_restore_and_graph_replay_bookkeeping():
    """Per-parameter step counts: a parameter that gets its first gradient after others have stepped, a
    state restored at different steps, and updates replayed from a captured graph all see the bias
    correction torch's per-parameter counters would give."""
    from distributed_pytorch_amd.ops import FusedAdamW

    torch.manual_seed(9)
    ref_p = [torch.randn(64, dtype=torch.float64).requires_grad_(), torch.randn(33, dtype=torch.float64).requires_grad_()]
    my_p = [p.detach().float().to(DEV).requires_grad_() for p in ref_p]
    ref, mine = torch.optim.AdamW(ref_p, lr=1e-2), FusedAdamW(my_p, lr=1e-2)
    for step in range(5):
        for k, (rp, mp) in enumerate(zip(ref_p, my_p)):
            if k == 1 and step < 2:       # the second parameter joins at step 2
                rp.grad, mp.grad = None, None
                continue
            g = torch.randn(rp.shape, dtype=torch.float64)
            rp.grad, mp.grad = g, g.float().to(DEV)
        ref.step()
        mine.step()
    for rp, mp in zip(ref_p, my_p):
        torch.testing.assert_close(mp.detach().cpu().double(), rp.detach(), rtol=2e-5, atol=2e-6)
    sd = mine.state_dict()
    assert [int(sd["state"][i]["step"]) for i in (0, 1)] == [5, 3]
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    # restore into a fresh optimizer and keep going: both cohorts continue from their own step
    mine2 = FusedAdamW(my_p, lr=1e-2)
    mine2.load_state_dict(sd)
    for step in range(2):
        for rp, mp in zip(ref_p, my_p):
            g = torch.randn(rp.shape, dtype=torch.float64)
            rp.grad, mp.grad = g, g.float().to(DEV)
        ref.step()
        mine2.step()
    for rp, mp in zip(ref_p, my_p):
        torch.testing.assert_close(mp.detach().cpu().double(), rp.detach(), rtol=2e-5, atol=2e-6)
    assert [int(mine2.state_dict()["state"][i]["step"]) for i in (0, 1)] == [7, 5]
    # graph-replay bookkeeping: hyper-parameters pushed without step(), host mirrors advanced
    mine2.param_groups[0]["lr"] = 5e-3
    mine2.sync_hparams()
    assert all(abs(float(hp[0]) - 5e-3) < 1e-9 for hp in mine2._hparams.values())
    mine2.note_external_steps(3)
    assert [int(mine2.state_dict()["state"][i]["step"]) for i in (0, 1)] == [10, 8]


def test_grad_norm_is_fixed_order_and_exact():
    """Sum of squares over many tensors (several dtypes, sizes beyond one table): two stages with a
    fixed summation order, so repeated runs are bit-identical, and the value matches fp64."""
    Cx = C()
    torch.manual_seed(11)
    ts = [torch.randn(n, device=DEV).to(dt) for n, dt in
          [(1, torch.float32), (8191, torch.bfloat16), (300_000, torch.float32), (70_001, torch.bfloat16)] + [(1000 + i, torch.float32) for i in range(40)]]
    outs = []
    for _ in range(3):
        out = torch.full((1,), 123.0, device=DEV)   # stale content must be overwritten, not added to
        Cx.grad_sumsq(ts, out)
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    ref = sum(float(t.double().pow(2).sum()) for t in ts)
    assert abs(outs[0].item() - ref) / ref < 1e-5


def test_fused_adamw_channels_last_parameters():
    """Convolution weights kept in channels-last memory format (the ResNet path): parameter, gradient and
    moments are walked in storage order, with gradients arriving in either memory format."""
    from distributed_pytorch_amd.ops import FusedAdamW

    torch.manual_seed(5)
    ref_p = [torch.randn(16, 8, 3, 3, dtype=torch.float64).requires_grad_(), torch.randn(16, dtype=torch.float64).requires_grad_()]
    my_p = [ref_p[0].detach().float().to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(),
            ref_p[1].detach().float().to(DEV).requires_grad_()]
    assert not my_p[0].is_contiguous()
    ref = torch.optim.AdamW(ref_p, lr=1e-2, weight_decay=0.1)
    mine = FusedAdamW(my_p, lr=1e-2, weight_decay=0.1, max_grad_norm=5.0)
    for step in range(4):
        for rp, mp in zip(ref_p, my_p):
            g = torch.randn(rp.shape, dtype=torch.float64)
            rp.grad = g
            gd = g.float().to(DEV)
            # even steps: gradient in the parameter's memory format; odd steps: plain contiguous
            mp.grad = gd.contiguous(memory_format=torch.channels_last) if (gd.dim() == 4 and step % 2 == 0) else gd
        torch.nn.utils.clip_grad_norm_(ref_p, 5.0)
        ref.step()
        mine.step()
    assert not my_p[0].is_contiguous() and my_p[0].is_contiguous(memory_format=torch.channels_last)
    for rp, mp in zip(ref_p, my_p):
        torch.testing.assert_close(mp.detach().cpu().double(), rp.detach(), rtol=2e-5, atol=2e-6)


# ----------------------------------------------------------------------------- fused Linear(1,H) -> Linear(H,N)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H,N,act", [(1000, 300, 4, None), (65, 4096, 4, "relu"), (4096, 513, 3, None), (257, 256, 10, "relu"),
                                       (64, 64, 16, None), (1, 7, 1, "relu")])
def test_rank1_mlp_matches_fp64_reference(dtype, M, H, N, act):
    """Both directions of the fused two-layer node against an fp64 evaluation of the same parameters,
    including hidden sizes that are not a multiple of the workgroup, N below the register tile and ReLU."""
    from distributed_pytorch_amd.ops.rank1_mlp import rank1_mlp

    torch.manual_seed(M + H + N)
    x = torch.rand(M, 1, device=DEV).to(dtype)
    w1 = (torch.randn(H, 1, device=DEV) * 1.5).to(dtype).requires_grad_()
    b1 = torch.randn(H, device=DEV).to(dtype).requires_grad_()
    w2 = (torch.randn(N, H, device=DEV) / H ** 0.5).to(dtype).requires_grad_()
    b2 = torch.randn(N, device=DEV).to(dtype).requires_grad_()
    y = rank1_mlp(x, w1, b1, w2, b2, act)
    dy = torch.randn(M, N, device=DEV).to(dtype)
    y.backward(dy)
    ref_in = [t.detach().double().requires_grad_() for t in (w1, b1, w2, b2)]
    h = x.double() @ ref_in[0].t() + ref_in[1]
    if act == "relu":
        h = torch.relu(h)
    yr = h @ ref_in[2].t() + ref_in[3]
    yr.backward(dy.double())
    t = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(y.double(), yr, **t)
    for got, ref, name in zip((w1, b1, w2, b2), ref_in, ("dw1", "db1", "dW2", "db2")):
        scale = max(1.0, float(ref.grad.abs().max()))
        assert got.grad.shape == ref.grad.shape, name
        err = float((got.grad.double() - ref.grad).abs().max()) / scale
        assert err < (2e-2 if dtype == torch.bfloat16 else 1e-4), (name, err)


def test_rank1_mlp_without_biases_and_determinism():
    from distributed_pytorch_amd.ops.rank1_mlp import rank1_mlp

    torch.manual_seed(0)
    x = torch.rand(5000, 1, device=DEV)
    w1 = torch.randn(1024, 1, device=DEV, requires_grad=True)
    w2 = torch.randn(4, 1024, device=DEV, requires_grad=True)
    grads = []
    for _ in range(2):
        w1.grad = w2.grad = None
        y = rank1_mlp(x, w1, None, w2, None)
        y.square().sum().backward()
        grads.append((w1.grad.clone(), w2.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])   # fixed-order reductions
    yr = (x.double() @ w1.detach().double().t()) @ w2.detach().double().t()
    torch.testing.assert_close(y.double(), yr, rtol=1e-4, atol=1e-3)


# ----------------------------------------------------------------------------- softmax cross-entropy
@pytest.mark.parametrize("N,Cc,ld", [(8, 4, 4), (1, 4, 4), (1000, 4, 4), (37, 10, 10), (64, 1000, 1000), (5, 1001, 1001),
                                     (33, 5000, 5000), (16, 50257, 50304), (3, 50257, 50257)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_xent_forward_backward(N, Cc, ld, dtype):
    from distributed_pytorch_amd.ops import softmax_cross_entropy

    torch.manual_seed(N * 31 + Cc)
    logits = (torch.randn(N, ld, device=DEV) * 3).to(dtype)
    targets = torch.randint(0, Cc, (N,), device=DEV)
    lg = logits.clone().requires_grad_()
    res = softmax_cross_entropy(lg, targets, n_classes=Cc)
    (res.loss * 1.7).backward()

    ref_in = logits[:, :Cc].double().requires_grad_()
    ref_loss = F.cross_entropy(ref_in, targets)
    (ref_loss * 1.7).backward()
    torch.testing.assert_close(res.loss.double(), ref_loss.detach(), rtol=1e-5, atol=1e-5)
    assert torch.equal(res.correct, ref_in.argmax(1) == targets)
    assert res.metrics[1].item() == float((ref_in.argmax(1) == targets).sum())
    assert res.metrics[2].item() == N
    g = lg.grad.double()
    gt = dict(rtol=1e-4, atol=1e-7) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-4)
    torch.testing.assert_close(g[:, :Cc], ref_in.grad, **gt)
    if ld > Cc:
        assert torch.count_nonzero(g[:, Cc:]) == 0


def test_xent_extreme_logits_ties_and_ignore_index():
    from distributed_pytorch_amd.ops import softmax_cross_entropy

    for Cc in (4, 300, 9000):
        logits = torch.zeros(6, Cc, device=DEV)
        logits[0, 1] = 1e4          # would overflow exp() without max subtraction
        logits[1, :] = -1e4
        logits[2, 2] = 5.0
        logits[2, 3] = 5.0          # tie: lowest index wins
        logits[3, Cc - 1] = 80.0
        targets = torch.tensor([1, 0, 2, Cc - 1, -100, 3], device=DEV)
        lg = logits.clone().requires_grad_()
        r = softmax_cross_entropy(lg, targets)
        r.loss.backward()
        ref_in = logits.double().requires_grad_()
        ref = F.cross_entropy(ref_in, targets, ignore_index=-100)
        ref.backward()
        assert torch.isfinite(r.loss)
        torch.testing.assert_close(r.loss.double(), ref.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(lg.grad.double(), ref_in.grad, rtol=1e-4, atol=1e-7)
        assert r.correct.tolist() == [True, True, True, True, False, False]
        assert r.metrics[2].item() == 5
        assert torch.count_nonzero(lg.grad[4]) == 0


def test_xent_sum_none_and_inplace_backward():
    from distributed_pytorch_amd.ops import softmax_cross_entropy

    torch.manual_seed(5)
    logits = torch.randn(40, 777, device=DEV)
    t = torch.randint(0, 777, (40,), device=DEV)
    a = logits.clone().requires_grad_()
    r = softmax_cross_entropy(a, t, reduction="sum")
    r.loss.backward()
    b = logits.clone().double().requires_grad_()
    F.cross_entropy(b, t, reduction="sum").backward()
    torch.testing.assert_close(a.grad.double(), b.grad, rtol=1e-4, atol=1e-6)
    c = logits.clone().requires_grad_()
    w = torch.rand(40, device=DEV)
    (softmax_cross_entropy(c, t, reduction="none").loss * w).sum().backward()
    d = logits.clone().double().requires_grad_()
    (F.cross_entropy(d, t, reduction="none") * w.double()).sum().backward()
    torch.testing.assert_close(c.grad.double(), d.grad, rtol=1e-4, atol=1e-6)
    # in-place: the gradient is written over the logits buffer
    e = logits.clone()
    leaf = torch.zeros_like(e, requires_grad=True)
    y = leaf + e
    softmax_cross_entropy(y, t, inplace_backward=True).loss.backward()
    torch.testing.assert_close(leaf.grad.double(), b.grad / 40, rtol=1e-4, atol=1e-7)


# ----------------------------------------------------------------------------- linear layers
def _ref_linear(x, w, b, act):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    if act == "relu":
        y = F.relu(y)
    elif act == "gelu":
        y = F.gelu(y)
    return y


@pytest.mark.parametrize("M,K,N", [(8, 1, 32), (1, 1, 4096), (1000, 1, 4096), (77, 1, 33),      # K = 1 outer product
                                   (8, 32, 4), (1000, 4096, 4), (65, 100, 3), (9, 4096, 16),    # skinny N
                                   (256, 256, 256), (130, 68, 260), (1, 64, 128), (512, 1024, 96), (300, 36, 52),
                                   (129, 67, 261), (33, 7, 19), (255, 129, 127), (64, 1001, 35)])       # odd extents: tail path
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_linear_forward_backward(M, K, N, dtype):
    from distributed_pytorch_amd.ops.linear import linear_bias_act, native_route

    torch.manual_seed(M + 7 * K + 13 * N)
    x = torch.randn(M, K, device=DEV).to(dtype)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, device=DEV).to(dtype)
    assert native_route(M, N, K, dtype) in ("outer", "skinny", "mfma")
    xs, ws, bs = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = linear_bias_act(xs, ws, bs, allow_library_gemm=False)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = _ref_linear(xr, wr, br, None)
    yr.backward(gy.double())
    half = dtype in (torch.bfloat16, torch.float16)
    eps = 3e-2 if dtype == torch.bfloat16 else 4e-3
    t = dict(rtol=eps, atol=eps) if half else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y.double(), yr.detach(), **t)
    scale = math.sqrt(max(M, 1))
    tg = dict(rtol=eps, atol=eps * scale) if half else dict(rtol=1e-4, atol=1e-4 * scale)
    torch.testing.assert_close(xs.grad.double(), xr.grad, **t)
    torch.testing.assert_close(ws.grad.double(), wr.grad, **tg)
    torch.testing.assert_close(bs.grad.double(), br.grad, **tg)


@pytest.mark.parametrize("act", ["relu", "gelu"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(200, 1, 512), (256, 128, 384), (100, 72, 136)])
def test_linear_fused_activation(act, dtype, M, K, N):
    from distributed_pytorch_amd.ops.linear import linear_bias_act

    torch.manual_seed(11)
    x = torch.randn(M, K, device=DEV).to(dtype)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, device=DEV).to(dtype)
    xs, ws, bs = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = linear_bias_act(xs, ws, bs, act=act, allow_library_gemm=False)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = _ref_linear(xr, wr, br, act)
    yr.backward(gy.double())
    t = dict(rtol=3e-2, atol=5e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    tg = dict(rtol=3e-2, atol=0.6) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=2e-3)
    tx = dict(rtol=3e-2, atol=0.2) if dtype == torch.bfloat16 else t  # dx sums N bf16-rounded terms
    torch.testing.assert_close(y.double(), yr.detach(), **t)
    torch.testing.assert_close(xs.grad.double(), xr.grad, **tx)
    torch.testing.assert_close(ws.grad.double(), wr.grad, **tg)
    torch.testing.assert_close(bs.grad.double(), br.grad, **tg)


@pytest.mark.parametrize("a_k,b_k", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_layouts_identity_and_asymmetric(a_k, b_k, dtype):
    """A = I against an asymmetric B catches swapped row/column maps in the MFMA C layout;
    exact small-integer data makes every product exact in bf16 and fp32."""
    M, N, K = 192, 136, 192
    A = torch.eye(M, K, device=DEV)
    Bm = (torch.arange(K, device=DEV).view(K, 1) % 7 - 3.0) + 2.0 * (torch.arange(N, device=DEV).view(1, N) % 5)  # B[k][n]
    Aop = (A if a_k else A.t().contiguous()).to(dtype)
    Bop = (Bm.t().contiguous() if b_k else Bm.contiguous()).to(dtype)
    out = C().gemm(Aop, Bop, a_k, b_k)
    ref = (A.double() @ Bm.double())
    assert torch.equal(out.double(), ref)
    # random integers in every operand, all four layouts
    torch.manual_seed(3)
    M, N, K = 264, 200, 328
    Af = torch.randint(-3, 4, (M, K), device=DEV).float()
    Bf = torch.randint(-3, 4, (K, N), device=DEV).float()
    Aop = (Af if a_k else Af.t().contiguous()).to(dtype)
    Bop = (Bf.t().contiguous() if b_k else Bf.contiguous()).to(dtype)
    out = C().gemm(Aop, Bop, a_k, b_k)
    ref = Af.double() @ Bf.double()
    if dtype == torch.bfloat16:
        torch.testing.assert_close(out.double(), ref, rtol=1e-2, atol=1.0)  # outputs up to ~3000 are rounded to bf16
    else:
        assert torch.equal(out.double(), ref)


def test_gemm_large_random_f32_and_bf16():
    torch.manual_seed(9)
    for dtype, t in ((torch.float32, dict(rtol=1e-4, atol=1e-3)), (torch.bfloat16, dict(rtol=2e-2, atol=0.5))):
        A = torch.randn(1024, 2048, device=DEV).to(dtype)
        B = torch.randn(1536, 2048, device=DEV).to(dtype)
        out = C().gemm(A, B, True, True)
        ref = A.double() @ B.double().t()
        torch.testing.assert_close(out.double(), ref, **t)


# ----------------------------------------------------------------------------- layer norm
@pytest.mark.parametrize("rows,cols", [(1, 768), (5, 768), (4097, 768), (33, 64), (16, 1024), (7, 2048), (9, 1280), (12, 256), (6, 384), (10, 320),
                                       (300, 1600), (5, 4096), (3, 8192), (37, 1001), (600, 5), (2, 7777)])  # the last six: one workgroup per row
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm(rows, cols, dtype):
    """Widths that a wave keeps in registers, widths beyond that (fp32 backward > 1024 columns, anything > 2048 / 4096) and
    widths that are not a multiple of the 16-byte vector all run on native kernels."""
    from distributed_pytorch_amd.ops.layernorm import _supported, layer_norm

    torch.manual_seed(rows + cols)
    assert _supported(torch.empty(1, cols, device=DEV, dtype=dtype), cols, True) and C().layernorm_supported(cols)
    x = (torch.randn(rows, cols, device=DEV) * 2 + 0.5).to(dtype)
    g = torch.randn(cols, device=DEV).to(dtype)
    b = torch.randn(cols, device=DEV).to(dtype)
    xs, gs, bs = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    y = layer_norm(xs, gs, bs, 1e-5)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr, gr, br = x.double().requires_grad_(), g.double().requires_grad_(), b.double().requires_grad_()
    yr = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    yr.backward(gy.double())
    t = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    s = math.sqrt(rows)
    tg = dict(rtol=3e-2, atol=5e-2 * s) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4 * s)
    torch.testing.assert_close(y.double(), yr.detach(), **t)
    torch.testing.assert_close(xs.grad.double(), xr.grad, **t)
    torch.testing.assert_close(gs.grad.double(), gr.grad, **tg)
    torch.testing.assert_close(bs.grad.double(), br.grad, **tg)


@pytest.mark.parametrize("a_k,b_k", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(4096, 2048, 128), (4000, 2056, 192), (2304, 3848, 64)])
def test_gemm_bf16_large_tile_kernel_exact_integers(a_k, b_k, M, N, K):
    """Shapes with >= 128 tiles of 256x256 and K % 64 == 0 take the LDS-DMA kernel; small-integer
    operands make every product and partial sum exact, so any layout / swizzle / clamp error shows
    up as a mismatch (outputs stay below 256 in magnitude, exactly representable in bf16)."""
    torch.manual_seed(M + N + K)
    Af = torch.randint(-2, 3, (M, K), device=DEV).float()
    Bf = torch.randint(-1, 2, (K, N), device=DEV).float()
    # make rows / columns distinguishable so a permutation of rows or columns cannot cancel out
    Af[:, 0] = (torch.arange(M, device=DEV) % 5 - 2).float()
    Bf[0, :] = (torch.arange(N, device=DEV) % 3 - 1).float()
    Aop = (Af if a_k else Af.t().contiguous()).bfloat16()
    Bop = (Bf.t().contiguous() if b_k else Bf.contiguous()).bfloat16()
    out = C().gemm(Aop, Bop, a_k, b_k)
    ref = Af.double() @ Bf.double()
    assert ref.abs().max() < 256
    assert torch.equal(out.double(), ref)


@pytest.mark.parametrize("a_k,b_k", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("M,N,K,busy", [(16384, 3072, 192, False), (16500, 3080, 256, False), (24576, 2304, 192, True)])
def test_gemm_bf16_streaming_walk_exact_integers(a_k, b_k, M, N, K, busy):
    """>= 3 tiles per CU: one workgroup per CU walks the tiles, each XCD through its own segment of the id list
    (845 tiles: segments of 106 and 105 ids, ragged last row / column of tiles).  `busy`: a second stream keeps
    the chip occupied meanwhile, so workgroups of the product start late and segments are handed over."""
    torch.manual_seed(M + N + K)
    Af = torch.randint(-2, 3, (M, K), device=DEV).float()
    Bf = torch.randint(-1, 2, (K, N), device=DEV).float()
    Af[:, 0] = (torch.arange(M, device=DEV) % 5 - 2).float()
    Bf[0, :] = (torch.arange(N, device=DEV) % 3 - 1).float()
    Aop = (Af if a_k else Af.t().contiguous()).bfloat16()
    Bop = (Bf.t().contiguous() if b_k else Bf.contiguous()).bfloat16()
    ref = Af.double() @ Bf.double()
    assert ref.abs().max() < 256
    if busy:
        side = torch.cuda.Stream()
        filler = torch.randn(64 * 1024 * 1024, device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(20):
                filler.mul_(1.0001)
    outs = [C().gemm(Aop, Bop, a_k, b_k) for _ in range(3)]
    torch.cuda.synchronize()
    for out in outs:
        assert torch.equal(out.double(), ref)


@pytest.mark.parametrize("M,N,K", [(4096, 2048, 256), (16384, 3072, 256)])
def test_gemm_bf16_large_tile_fused_epilogue(M, N, K):
    """One tile per workgroup (128 tiles) and the streaming walk (768 tiles: bias fetched a tile ahead)."""
    torch.manual_seed(4)
    x = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(N, K, device=DEV) / 16).bfloat16()
    b = torch.randn(N, device=DEV).bfloat16()
    for act, fn in ((1, torch.relu), (2, torch.nn.functional.gelu)):
        y, pre = C().linear_forward(x, w, b, act, True)
        ref_pre = x.double() @ w.double().t() + b.double()
        torch.testing.assert_close(pre.double(), ref_pre, rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(y.double(), fn(ref_pre), rtol=2e-2, atol=2e-2)


# ----------------------------------------------------------------------------- attention
def _attention_reference(qkv, H, scale):
    """fp32 causal attention on the same (bf16-rounded) inputs, with autograd for the gradients."""
    B, T, C3 = qkv.shape
    Cd = C3 // 3
    q, k, v = (t.reshape(B, T, H, 64).transpose(1, 2).float().detach().requires_grad_() for t in qkv.split(Cd, dim=2))
    s = (q @ k.transpose(-1, -2)) * scale
    mask = torch.ones(T, T, device=qkv.device, dtype=torch.bool).tril()
    s = s.masked_fill(~mask, float("-inf"))
    lse = torch.logsumexp(s, -1)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, Cd)
    return (q, k, v), o, lse


@pytest.mark.parametrize("kernel", ["classic", "wide", "tile32"])
@pytest.mark.parametrize("B,T,H", [(1, 128, 1), (2, 256, 3), (1, 640, 2), (2, 1024, 12), (1, 128, 2), (2, 384, 5)])
def test_flash_attention_forward_backward(B, T, H, kernel):
    if kernel == "wide" and (B * H) % 2:
        pytest.skip("the 64-queries-per-wave kernel takes heads two at a time")
    torch.manual_seed(B * 1000 + T + H)
    Cd = H * 64
    qkv = torch.randn(B, T, 3 * Cd, device=DEV).bfloat16()
    scale = 1.0 / math.sqrt(64)
    assert C().attention_supported(T, 64, 3 * Cd)
    o, lse = C().attention_forward(qkv, H, scale, kernel)
    (q, k, v), ref_o, ref_lse = _attention_reference(qkv, H, scale)
    # the row sums are taken over the bf16-rounded probabilities (the values that multiply V), by the
    # matrix pipe: the log-sum-exp carries that rounding, at most 2^-9 relative on the sum
    torch.testing.assert_close(lse, ref_lse, rtol=0, atol=4e-3)
    torch.testing.assert_close(o.float(), ref_o, rtol=2e-2, atol=2e-2)
    go = torch.randn_like(o)
    ref_o.backward(go.float())
    ref_g = torch.cat([t.grad.transpose(1, 2).reshape(B, T, Cd) for t in (q, k, v)], dim=2)
    for bwd_kernel in ("classic", "wide"):  # 16x16x32 / 32x32x16 matrix instructions
        dqkv, _ = C().attention_backward(go, qkv, o, lse, H, scale, False, bwd_kernel)
        # bf16 P / dS operands and bf16 outputs: error scales with the gradient magnitude
        err = (dqkv.float() - ref_g).abs().max().item()
        assert err <= 1e-2 * ref_g.abs().max().item() + 1e-2, (bwd_kernel, err)


@pytest.mark.parametrize("kernel", ["classic", "wide", "tile32"])
def test_flash_attention_large_logits_and_causality(kernel):
    """Scores of magnitude ~60 (softmax saturates) stay finite, and outputs at position t do not
    depend on inputs after t."""
    torch.manual_seed(7)
    B, T, H = 1, 256, 2
    Cd = H * 64
    qkv = (torch.randn(B, T, 3 * Cd, device=DEV) * 3).bfloat16()
    scale = 1.0
    o, lse = C().attention_forward(qkv, H, scale, kernel)
    _, ref_o, ref_lse = _attention_reference(qkv, H, scale)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    torch.testing.assert_close(lse, ref_lse, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(o.float(), ref_o, rtol=3e-2, atol=3e-2)
    qkv2 = qkv.clone()
    qkv2[:, 130:] = torch.randn_like(qkv2[:, 130:])
    o2, _ = C().attention_forward(qkv2, H, scale, kernel)
    assert torch.equal(o[:, :130], o2[:, :130])
    g = torch.zeros_like(o)
    g[:, :130] = 1
    for bwd_kernel in ("classic", "wide"):
        d, _ = C().attention_backward(g, qkv, o, lse, H, scale, False, bwd_kernel)
        assert torch.count_nonzero(d[:, 130:].float()) == 0  # no gradient flows to later positions


@pytest.mark.parametrize("kernel", ["classic", "wide", "tile32"])
def test_flash_attention_growing_maximum(kernel):
    """Scores that keep growing along the keys (every block raises the running maximum by far more than the deferral
    threshold) exercise the rescale of the partial output and row sum in every step."""
    torch.manual_seed(11)
    B, T, H = 2, 512, 2
    Cd = H * 64
    q = torch.randn(B, T, H, 64, device=DEV)
    k = torch.randn(B, T, H, 64, device=DEV) * 0.1
    ramp = torch.linspace(0, 6, T, device=DEV).view(1, T, 1)
    q[..., 0] = 4.0                      # score(q, key j) ~ 4 * ramp(j) * 8 / 8: grows by ~24 over the sequence
    k[..., 0] = ramp
    v = torch.randn(B, T, H, 64, device=DEV)
    qkv = torch.cat([q.reshape(B, T, Cd), k.reshape(B, T, Cd), v.reshape(B, T, Cd)], dim=2).bfloat16()
    scale = 1.0
    o, lse = C().attention_forward(qkv, H, scale, kernel)
    _, ref_o, ref_lse = _attention_reference(qkv, H, scale)
    torch.testing.assert_close(lse, ref_lse, rtol=1e-4, atol=4e-3)
    torch.testing.assert_close(o.float(), ref_o, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("kernel", ["classic", "wide"])
@pytest.mark.parametrize("B,T,H", [(1, 128, 1), (2, 384, 3), (4, 1024, 12), (16, 1024, 2), (20, 1024, 1)])  # the last two: >= 512 partial rows (wide reduction kernel)
def test_attention_backward_bias_gradient(B, T, H, kernel):
    """The column sums the backward kernels emit equal the sum over positions of the dqkv they wrote."""
    torch.manual_seed(T + H)
    Cd = H * 64
    qkv = torch.randn(B, T, 3 * Cd, device=DEV).bfloat16()
    scale = 1.0 / math.sqrt(64)
    o, lse = C().attention_forward(qkv, H, scale)
    go = torch.randn_like(o)
    dqkv, db = C().attention_backward(go, qkv, o, lse, H, scale, True, kernel)
    plain, none = C().attention_backward(go, qkv, o, lse, H, scale, False, kernel)
    assert none is None and torch.equal(dqkv, plain)
    ref = dqkv.float().sum((0, 1))
    assert db.shape == (3 * Cd,) and db.dtype == torch.bfloat16
    torch.testing.assert_close(db.float(), ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item())


def test_projected_attention_matches_separate_nodes():
    from distributed_pytorch_amd.ops.attention import causal_attention_qkvpacked, projected_causal_attention

    torch.manual_seed(11)
    B, T, H, Cin = 2, 256, 4, 256
    x = torch.randn(B, T, Cin, device=DEV).bfloat16().requires_grad_()
    w = (torch.randn(3 * H * 64, Cin, device=DEV) * 0.05).bfloat16().requires_grad_()
    b = (torch.randn(3 * H * 64, device=DEV) * 0.1).bfloat16().requires_grad_()
    y = projected_causal_attention(x, w, b, H)
    assert y is not None
    g = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), g)
    xr, wr, br = (t.detach().float().requires_grad_() for t in (x, w, b))
    yr = causal_attention_qkvpacked(torch.nn.functional.linear(xr, wr, br), H, impl="sdpa")
    rx, rw, rb = torch.autograd.grad(yr, (xr, wr, br), g.float())
    torch.testing.assert_close(y.float(), yr, rtol=3e-2, atol=3e-2)
    for got, ref in ((gx, rx), (gw, rw), (gb, rb)):
        assert (got.float() - ref).abs().max().item() <= 3e-2 * ref.abs().max().item() + 1e-2
    # not covered -> None, the caller takes the two-node path
    assert projected_causal_attention(x[:, :100], w, b, H) is None
    assert projected_causal_attention(x, w, None, H) is None


def test_attention_op_autograd_and_fallback():
    from distributed_pytorch_amd.ops.attention import causal_attention_qkvpacked, native_attention_supported

    torch.manual_seed(3)
    B, T, H = 2, 256, 4
    qkv = torch.randn(B, T, 3 * H * 64, device=DEV).bfloat16().requires_grad_()
    assert native_attention_supported(qkv, H)
    y = causal_attention_qkvpacked(qkv, H, impl="native")
    g = torch.randn_like(y)
    (gn,) = torch.autograd.grad(y, qkv, g)
    qkv_s = qkv.detach().clone().requires_grad_()
    ys = causal_attention_qkvpacked(qkv_s, H, impl="sdpa")
    (gs,) = torch.autograd.grad(ys, qkv_s, g)
    torch.testing.assert_close(y.float(), ys.float(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(gn.float(), gs.float(), rtol=5e-2, atol=5e-2)
    # unsupported head sizes fall back, or raise when the native path is demanded
    odd = torch.randn(1, 128, 3 * 32, device=DEV).bfloat16()
    assert not native_attention_supported(odd, 1) and not native_attention_supported(odd, 1, pad_ok=True)
    assert causal_attention_qkvpacked(odd, 1).shape == (1, 128, 32)
    with pytest.raises(RuntimeError):
        causal_attention_qkvpacked(odd, 1, impl="native")


@pytest.mark.parametrize("B,T,H", [(2, 100, 2), (1, 129, 1), (3, 1000, 4), (2, 255, 3)])
def test_attention_op_pads_sequences_to_whole_blocks(B, T, H):
    """A sequence length that is not a multiple of the kernels' 128-row block runs on the native kernels
    after zero padding: outputs and gradients of the real rows equal the fp32 reference, and a
    non-contiguous (sliced) input is accepted."""
    from distributed_pytorch_amd.ops.attention import causal_attention_qkvpacked, native_attention_supported

    torch.manual_seed(T * 7 + H)
    Cd = H * 64
    scale = 1.0 / math.sqrt(64)
    base = torch.randn(B, T + 3, 3 * Cd, device=DEV).bfloat16()
    qkv = base[:, 3:].detach().requires_grad_()  # not contiguous
    assert not native_attention_supported(qkv, H) and native_attention_supported(qkv, H, pad_ok=True)
    y = causal_attention_qkvpacked(qkv, H, impl="native")
    assert y.shape == (B, T, Cd) and y.is_contiguous()
    (q, k, v), ref_o, _ = _attention_reference(qkv.detach(), H, scale)
    torch.testing.assert_close(y.float(), ref_o, rtol=2e-2, atol=2e-2)
    go = torch.randn_like(y)
    (g,) = torch.autograd.grad(y, qkv, go)
    ref_o.backward(go.float())
    ref_g = torch.cat([t.grad.transpose(1, 2).reshape(B, T, Cd) for t in (q, k, v)], dim=2)
    err = (g.float() - ref_g).abs().max().item()
    assert g.shape == qkv.shape and err <= 1e-2 * ref_g.abs().max().item() + 1e-2, err


@pytest.mark.parametrize("a_k,b_k", [(False, False), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(768, 768, 2048), (520, 1032, 3072)])
def test_gemm_bf16_split_k_exact_integers(a_k, b_k, M, N, K):
    """Few output tiles + long contraction (the weight-gradient shape) takes the split-K path:
    fp32 partials per contraction range, summed in a fixed order.  Integer operands keep every
    partial exact, so the result must equal the fp64 product bit for bit."""
    torch.manual_seed(K + M)
    Af = torch.randint(-1, 2, (M, K), device=DEV).float()
    Bf = torch.randint(-1, 2, (K, N), device=DEV).float()
    Af[:, 0] = (torch.arange(M, device=DEV) % 3 - 1).float()
    Bf[0, :] = (torch.arange(N, device=DEV) % 3 - 1).float()
    Aop = (Af if a_k else Af.t().contiguous()).bfloat16()
    Bop = (Bf.t().contiguous() if b_k else Bf.contiguous()).bfloat16()
    out = C().gemm(Aop, Bop, a_k, b_k)
    ref = Af.double() @ Bf.double()
    assert ref.abs().max() < 256
    assert torch.equal(out.double(), ref)
    again = C().gemm(Aop, Bop, a_k, b_k)
    assert torch.equal(out, again)  # deterministic reduction order


@pytest.mark.parametrize("rows,cols,dtype", [(300, 768, torch.bfloat16), (64, 1024, torch.bfloat16), (33, 256, torch.float32),
                                             (40, 1280, torch.float32), (21, 5120, torch.bfloat16), (9, 1003, torch.bfloat16)])  # the last three: one workgroup per row
def test_add_layernorm_fused_matches_unfused(rows, cols, dtype):
    """s = x + r, y = LN(s) in one kernel; backward folds the gradient arriving at s into dx."""
    from distributed_pytorch_amd.ops.layernorm import add_layer_norm, layer_norm

    torch.manual_seed(rows + cols)
    x = torch.randn(rows, cols, device=DEV).to(dtype)
    r = torch.randn(rows, cols, device=DEV).to(dtype)
    g = torch.randn(cols, device=DEV).to(dtype)
    b = torch.randn(cols, device=DEV).to(dtype)
    gs_, gy_ = torch.randn(rows, cols, device=DEV).to(dtype), torch.randn(rows, cols, device=DEV).to(dtype)

    def run(fused):
        xs, rs, gg, bb = (t.clone().requires_grad_() for t in (x, r, g, b))
        if fused:
            s, y = add_layer_norm(xs, rs, gg, bb, 1e-5)
        else:
            s = xs + rs
            y = layer_norm(s, gg, bb, 1e-5)
        torch.autograd.backward([s, y], [gs_, gy_])
        return s, y, xs.grad, rs.grad, gg.grad, bb.grad

    fs, fy, fdx, fdr, fdg, fdb = run(True)
    us, uy, udx, udr, udg, udb = run(False)
    assert torch.equal(fs, us) and torch.equal(fy, uy)  # the sum is rounded before normalising, as unfused
    t = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fdx.float(), udx.float(), **t)
    torch.testing.assert_close(fdr.float(), udr.float(), **t)
    assert torch.equal(fdx, fdr)
    torch.testing.assert_close(fdg.float(), udg.float(), **t)
    torch.testing.assert_close(fdb.float(), udb.float(), **t)
    # fp64 reference for the fused gradient of x
    xr, rr = x.double().requires_grad_(), r.double().requires_grad_()
    sr = xr + rr
    yr = F.layer_norm(sr, (cols,), g.double(), b.double(), 1e-5)
    torch.autograd.backward([sr, yr], [gs_.double(), gy_.double()])
    t2 = dict(rtol=4e-2, atol=6e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(fdx.double(), xr.grad, **t2)


@pytest.mark.parametrize("N,K,V,nc,dtype", [(300, 64, 1000, 1000, torch.float32), (257, 128, 50304, 50257, torch.bfloat16),
                                             (64, 32, 2048, 2000, torch.bfloat16), (33, 16, 130, 129, torch.float32)])
def test_projection_cross_entropy_fused_matches_reference(N, K, V, nc, dtype):
    """lm-head GEMM + loss + logits gradient as one node (gradient written over the logits by the
    loss kernel itself) against fp64 autograd, including ignored rows, padding classes, rows whose
    length is not a multiple of the vector width, and a non-unit upstream gradient."""
    from distributed_pytorch_amd.ops.cross_entropy import projection_cross_entropy

    torch.manual_seed(N + V)
    x = (torch.randn(N, K, device=DEV) * 0.5).to(dtype)
    w = (torch.randn(V, K, device=DEV) * 0.5).to(dtype)
    t = torch.randint(0, nc, (N,), device=DEV)
    t[::7] = -100
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    r = projection_cross_entropy(xs, ws, t, n_classes=nc)
    (r.loss * 3.0).backward()
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    logits = (xr @ wr.t())[:, :nc]
    ref = F.cross_entropy(logits, t, ignore_index=-100)
    (ref * 3.0).backward()
    tl = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(r.loss.double(), ref.detach(), **tl)
    valid = t != -100
    ref_correct = (logits.argmax(1) == t) & valid
    if dtype == torch.float32:
        assert torch.equal(r.correct, ref_correct)
    assert not r.correct[~valid].any()
    assert abs(float(r.metrics[2]) - float(valid.sum())) < 0.5
    tg = dict(rtol=3e-2, atol=3e-3) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(xs.grad.double(), xr.grad, **tg)
    torch.testing.assert_close(ws.grad.double(), wr.grad, **tg)
    assert torch.count_nonzero(ws.grad[nc:].float()) == 0  # padding classes get no gradient


@pytest.mark.parametrize("ld,nc,dtype", [(8192, 8191, torch.bfloat16), (8192, 8192, torch.float16), (50304, 50257, torch.bfloat16),
                                         (57344, 57344, torch.bfloat16), (57352, 57350, torch.bfloat16), (50304, 50257, torch.float32),
                                         (4104, 4097, torch.bfloat16)])
def test_fused_loss_kernel_row_in_registers_and_fallbacks(ld, nc, dtype):
    """The in-place loss + gradient kernel on its own: 16-bit rows of up to 7 * 1024 vectors are kept in
    registers (with the class count anywhere inside the last vector), longer and fp32 rows take the
    two-read path; both against fp64, with ignored targets, ties and padding columns."""
    from distributed_pytorch_amd import _ext

    C = _ext.C()
    torch.manual_seed(ld + nc)
    N = 37
    logits = (torch.randn(N, ld, device=DEV) * 4).to(dtype)
    logits[3, 5] = logits[3, 9] = 30.0       # tie: the lower index is the prediction
    logits[4, nc - 1] = 40.0                 # maximum in the last real column
    logits[:, nc:] = 50.0                    # padding columns must not matter
    t = torch.randint(0, nc, (N,), device=DEV)
    t[3], t[4], t[5], t[6] = 5, nc - 1, -100, nc - 1
    ref_in = logits[:, :nc].double().requires_grad_()
    ref = F.cross_entropy(ref_in, t, ignore_index=-100)
    ref.backward()
    work = logits.clone()
    correct, metrics = C.xent_fused(work, t, nc, -100, True)
    valid = t != -100
    torch.testing.assert_close(metrics[3].double(), ref.detach(), rtol=1e-5, atol=1e-5)
    assert torch.equal(correct, (ref_in.argmax(1) == t) & valid)
    assert bool(correct[3]) and bool(correct[4]) and not bool(correct[5])
    tol = dict(rtol=1e-4, atol=1e-7) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=2e-5)
    torch.testing.assert_close(work[:, :nc].double(), ref_in.grad, **tol)
    assert torch.count_nonzero(work[:, nc:]) == 0 and torch.count_nonzero(work[5]) == 0


@pytest.mark.parametrize("rows,cols", [(700, 768), (130, 1600), (64, 2500)])  # wave per row; workgroup per row (> 1365 columns with the column sum)
def test_add_layernorm_with_branch_bias_and_its_gradient(rows, cols):
    """s = x + r + r_bias in the forward kernel; the backward kernel also returns colsum(ds), the
    gradient of r_bias, next to dgamma / dbeta."""
    from distributed_pytorch_amd.ops.layernorm import add_layer_norm

    torch.manual_seed(5)
    x = torch.randn(rows, cols, device=DEV).bfloat16()
    r = torch.randn(rows, cols, device=DEV).bfloat16()
    rb = torch.randn(cols, device=DEV).bfloat16()
    g = torch.randn(cols, device=DEV).bfloat16()
    b = torch.randn(cols, device=DEV).bfloat16()
    gs_, gy_ = torch.randn(rows, cols, device=DEV).bfloat16(), torch.randn(rows, cols, device=DEV).bfloat16()
    xs, rs, rbs, gg, bb = (t.clone().requires_grad_() for t in (x, r, rb, g, b))
    s, y = add_layer_norm(xs, rs, gg, bb, 1e-5, rbs)
    torch.autograd.backward([s, y], [gs_, gy_])
    xr, rr, rbr, gr, br = (t.double().requires_grad_() for t in (x, r, rb, g, b))
    sr = xr + (rr + rbr)
    yr = F.layer_norm(sr, (cols,), gr, br, 1e-5)
    torch.autograd.backward([sr, yr], [gs_.double(), gy_.double()])
    torch.testing.assert_close(s.double(), sr.detach(), rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(y.double(), yr.detach(), rtol=3e-2, atol=6e-2)
    torch.testing.assert_close(xs.grad.double(), xr.grad, rtol=4e-2, atol=6e-2)
    assert torch.equal(xs.grad, rs.grad)
    scale = math.sqrt(rows)
    torch.testing.assert_close(rbs.grad.double(), rbr.grad, rtol=3e-2, atol=5e-2 * scale)
    torch.testing.assert_close(gg.grad.double(), gr.grad, rtol=3e-2, atol=5e-2 * scale)
    torch.testing.assert_close(bb.grad.double(), br.grad, rtol=3e-2, atol=5e-2 * scale)


@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_fused_mlp_matches_reference(act):
    """c_fc -> act -> c_proj as one node: the backward's fused GEMM epilogue (act' and the bias
    column sums) against fp64 autograd."""
    from distributed_pytorch_amd.ops.mlp import fused_mlp, fused_mlp_supported

    torch.manual_seed(11)
    M, Cin, Hid = 512, 256, 1024
    x = (torch.randn(M, Cin, device=DEV) * 0.5).bfloat16()
    w1 = (torch.randn(Hid, Cin, device=DEV) / 16).bfloat16()
    b1 = (torch.randn(Hid, device=DEV) * 0.1).bfloat16()
    w2 = (torch.randn(Cin, Hid, device=DEV) / 32).bfloat16()
    b2 = (torch.randn(Cin, device=DEV) * 0.1).bfloat16()
    assert fused_mlp_supported(x, w1, b1, w2, act)
    ts = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2)]
    y = fused_mlp(ts[0], ts[1], ts[2], ts[3], ts[4], act)
    gy = torch.randn_like(y)
    y.backward(gy)
    tr = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2)]
    pre = tr[0] @ tr[1].t() + tr[2]
    h = F.gelu(pre) if act == "gelu" else F.relu(pre)
    yr = h @ tr[3].t() + tr[4]
    yr.backward(gy.double())
    torch.testing.assert_close(y.double(), yr.detach(), rtol=3e-2, atol=3e-2)
    names = ["dx", "dw_fc", "db_fc", "dw_proj", "db_proj"]
    for n, a, b in zip(names, ts, tr):
        ref = b.grad
        err = (a.grad.double() - ref).abs().max().item()
        assert err <= 3e-2 * ref.abs().max().item() + 1e-2, (n, err, ref.abs().max().item())
    # shapes that do not fill whole tiles take the unfused path and still agree
    xs = x[:100].clone().requires_grad_()
    assert not fused_mlp_supported(xs, w1, b1, w2, act)
    y2 = fused_mlp(xs, w1, b1, w2, b2, act)
    torch.testing.assert_close(y2.double(), yr.detach()[:100], rtol=3e-2, atol=3e-2)


def test_dgrad_act_backward_gelu_table_over_all_magnitudes():
    """The epilogue looks GELU' up per bf16 input: every binade from denormal to huge, both signs,
    zeros and infinities, against the closed form in fp64."""
    from distributed_pytorch_amd.ops.linear import ACT_GELU

    torch.manual_seed(2)
    M, N, K = 256, 512, 128
    dy = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(K, N, device=DEV) / 8).bfloat16()
    mags = torch.cat([torch.tensor([0.0, 1e-40, 1e-30, 1e-10, 2.0 ** -13, 2.0 ** -12, 7.97, 8.0, 30.0, 1e30, float("inf")]),
                      torch.logspace(-5, 1.2, 4000)])
    vals = torch.cat([mags, -mags]).to(DEV)
    aux = vals[torch.randint(0, vals.numel(), (M, N), device=DEV)].bfloat16()
    aux.view(-1)[: vals.numel()] = vals.bfloat16()  # every probe value at least once
    assert C().dgrad_act_backward_supported(M, N, K)
    g, db = C().dgrad_act_backward(dy, w, aux, ACT_GELU, True)
    x = aux.double()
    d = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    d = torch.where(torch.isinf(x), (x > 0).double(), d)
    ref = (dy.double() @ w.double()) * d
    assert torch.isfinite(g.float()).all()
    err = (g.double() - ref).abs()
    assert (err <= 8e-3 * ref.abs() + 2e-3).all(), err.max().item()
    torch.testing.assert_close(db.double(), g.double().sum(0), rtol=1e-2, atol=0.15)


@pytest.mark.parametrize("M,N,K,act", [(16384, 3072, 192, "gelu"), (20480, 2560, 256, "relu"), (16384, 3072, 768, "gelu")])
def test_dgrad_act_backward_many_tiles(M, N, K, act):
    """Several waves of workgroups (>= 3 tiles per CU, the benchmark's regime): input gradient and its column sums
    against fp64 on random pre-activations (every table entry in use), three runs bit-equal."""
    from distributed_pytorch_amd.ops.linear import ACT_GELU, ACT_RELU

    torch.manual_seed(M + N + K)
    dy = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(K, N, device=DEV) / 8).bfloat16()
    aux = (torch.randn(M, N, device=DEV) * 1.5).bfloat16()
    assert C().dgrad_act_backward_supported(M, N, K)
    code = ACT_GELU if act == "gelu" else ACT_RELU
    runs = [C().dgrad_act_backward(dy, w, aux, code, True) for _ in range(3)]
    torch.cuda.synchronize()
    g, db = runs[0]
    for g2, db2 in runs[1:]:
        assert torch.equal(g, g2) and torch.equal(db, db2)
    x = aux.double()
    if act == "gelu":
        d = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    else:
        d = (x > 0).double()
    ref = (dy.double() @ w.double()) * d
    err = (g.double() - ref).abs()
    assert (err <= 8e-3 * ref.abs() + 4e-3).all(), err.max().item()
    torch.testing.assert_close(db.double(), g.double().sum(0), rtol=1e-2, atol=0.02 * math.sqrt(M))


# ----------------------------------------------------------------------------- batch normalisation (channels last)
@pytest.mark.parametrize("shape", [(8, 64, 28, 28), (3, 64, 7, 5), (4, 256, 14, 14), (2, 2048, 7, 7), (5, 512, 3, 3), (16, 128, 1, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu,residual", [(False, False), (True, False), (True, True), (False, True)])
def test_batchnorm_training_forward_backward(shape, dtype, relu, residual):
    """relu(bn(x) + residual) with batch statistics against fp64 autograd on the same (rounded) inputs: output,
    running statistics, and the gradients of x, weight, bias and the residual."""
    from distributed_pytorch_amd.ops import BatchNorm2d

    torch.manual_seed(sum(shape) + relu + 2 * residual)
    n, c, h, w = shape
    x = (torch.randn(shape, device=DEV) * 1.5 + 0.3).to(dtype).contiguous(memory_format=torch.channels_last)
    r = torch.randn(shape, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last) if residual else None
    bn = BatchNorm2d(c).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, device=DEV) + 0.5)
        bn.bias.copy_(torch.randn(c, device=DEV) * 0.2)
    ref = torch.nn.BatchNorm2d(c).to(DEV).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    xs = x.clone().requires_grad_()
    rs = r.clone().requires_grad_() if residual else None
    y = bn(xs, residual=rs, relu=relu)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    go = torch.randn(shape, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    y.backward(go)

    xd = x.double().requires_grad_()
    rd = r.double().requires_grad_() if residual else None
    yd = ref(xd)
    if residual:
        yd = yd + rd
    if relu:
        yd = torch.relu(yd)
    yd.backward(go.double())
    lo = dtype == torch.bfloat16
    torch.testing.assert_close(y.double(), yd.detach(), rtol=2e-2 if lo else 1e-4, atol=3e-2 if lo else 1e-4)
    torch.testing.assert_close(bn.running_mean.double(), ref.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn.running_var.double(), ref.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1
    # where the bf16 output rounds to a different side of zero than the fp64 one, the ReLU masks differ: compare the
    # gradients only where both agree (everywhere for fp32 up to measure zero)
    agree = ((y.double() > 0) == (yd > 0)) if relu else torch.ones_like(yd, dtype=torch.bool)
    frac = agree.double().mean().item()
    assert frac > (0.99 if lo else 0.9999)
    scale = go.double().abs().mean() * (bn.weight.double().abs().mean())
    gx_err = ((xs.grad.double() - xd.grad) * agree).abs().mean() / scale
    assert gx_err < (2e-2 if lo else 1e-4), gx_err
    if residual:
        torch.testing.assert_close((rs.grad.double() * agree), (rd.grad * agree), rtol=1e-2 if lo else 1e-5, atol=1e-2 if lo else 1e-6)
    if frac == 1.0:
        rows = n * h * w
        tol = dict(rtol=3e-2, atol=3e-2 * rows ** 0.5) if lo else dict(rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(bn.weight.grad.double(), ref.weight.grad, **tol)
        torch.testing.assert_close(bn.bias.grad.double(), ref.bias.grad, **tol)


def test_batchnorm_offset_channels_determinism_eval_and_bf16_module():
    """Statistics of channels with a large common offset (variance must not cancel), bitwise repeatability of the
    two-stage reductions, the evaluation-mode path, and a module converted with .to(bfloat16) (bf16 affine
    parameters and running statistics, as the ResNet benchmark uses it)."""
    from distributed_pytorch_amd.ops import BatchNorm2d

    torch.manual_seed(11)
    c = 128
    x = (torch.randn(6, c, 9, 9, device=DEV) * 0.01 + 1000.0).contiguous(memory_format=torch.channels_last)
    bn = BatchNorm2d(c).to(DEV)
    y = bn(x)
    ref = torch.nn.functional.batch_norm(x.double(), None, None, bn.weight.double(), bn.bias.double(), True, 0.1, bn.eps)
    torch.testing.assert_close(y.double(), ref, rtol=1e-2, atol=2e-2)   # fp32 input near 1000 carries ~6e-5 of rounding on a 1e-2 spread
    assert abs(float(y.detach().double().var()) - 1e-4 / (1e-4 + bn.eps)) < 0.02   # the spread survived the offset (eps = 1e-5 on a variance of 1e-4)

    xb = torch.randn(32, c, 14, 14, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    for _ in range(2):
        m = BatchNorm2d(c).to(DEV)
        xs = xb.clone().requires_grad_()
        out = m(xs, relu=True)
        out.backward(torch.ones_like(out))
        outs.append((out.detach().clone(), xs.grad.clone(), m.weight.grad.clone(), m.running_var.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)

    m = BatchNorm2d(c).to(DEV).to(torch.bfloat16)
    assert m.weight.dtype == torch.bfloat16 and m.running_mean.dtype == torch.bfloat16
    r = torch.randn_like(xb)
    out = m(xb.clone().requires_grad_(), residual=r, relu=True)
    out.float().sum().backward()
    assert m.weight.grad.dtype == torch.bfloat16 and torch.isfinite(m.weight.grad.float()).all()
    refm = torch.nn.BatchNorm2d(c).to(DEV).to(torch.bfloat16)
    torch.testing.assert_close(out.float(), torch.relu(refm(xb) + r).float(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(m.running_var.float(), refm.running_var.float(), rtol=2e-2, atol=2e-2)
    m.eval()
    with torch.no_grad():
        e = m(xb, residual=r, relu=True)
        refm.eval()
        torch.testing.assert_close(e.float(), torch.relu(refm(xb) + r).float(), rtol=3e-2, atol=3e-2)
    # same keys as the stock module
    assert list(m.state_dict().keys()) == list(refm.state_dict().keys())


@pytest.mark.parametrize("n,cin,cout,hw", [(8, 256, 1024, 14), (4, 512, 128, 9), (2, 2048, 512, 7), (3, 64, 256, 10)])
def test_pointwise_conv_as_gemm_matches_stock_convolution(n, cin, cout, hw):
    """1x1 stride-1 convolution over channels-last bf16 activations on the GEMM kernels (layers narrower than 128
    channels stay on the stock path): output and all gradients against the fp32 convolution of the same values."""
    from distributed_pytorch_amd.ops import PointwiseConv2d

    torch.manual_seed(cin + cout)
    conv = PointwiseConv2d(cin, cout).to(DEV).bfloat16().to(memory_format=torch.channels_last)
    x = torch.randn(n, cin, hw, hw, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    go = torch.randn(n, cout, hw, hw, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    y = conv(x)
    assert y.shape == (n, cout, hw, hw) and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(go)
    xr = x.detach().float().requires_grad_()
    wr = conv.weight.detach().float().requires_grad_()
    yr = F.conv2d(xr, wr)
    yr.backward(go.float())
    torch.testing.assert_close(y.float(), yr.detach(), rtol=2e-2, atol=2e-2 * cin ** 0.5 * 0.1)
    torch.testing.assert_close(x.grad.float(), xr.grad, rtol=2e-2, atol=2e-2 * cout ** 0.5 * 0.1)
    assert conv.weight.grad.shape == conv.weight.shape
    werr = (conv.weight.grad.float() - wr.grad).norm() / wr.grad.norm()
    assert werr < 1e-2, werr


# ----------------------------------------------------------------------------- embedding
@pytest.mark.parametrize("B,T,C,V,P", [(2, 16, 64, 50, 32), (3, 128, 768, 1000, 128), (8, 1024, 768, 50304, 1024), (5, 7, 8, 3, 9)])
def test_token_position_embedding_forward_backward(B, T, C, V, P):
    """One-kernel token + position embedding against the stock gather / add and their autograd: the forward
    is bit-equal (fp32 sum, one rounding), the gradients equal an fp32 accumulation rounded to bf16, and two
    runs of the backward agree bit for bit (fixed summation order, no atomics)."""
    from distributed_pytorch_amd.ops.embedding import native_embedding_supported, token_position_embedding

    torch.manual_seed(B * 100 + T)
    wte = torch.randn(V, C, device=DEV).bfloat16().requires_grad_()
    wpe = torch.randn(P, C, device=DEV).bfloat16().requires_grad_()
    idx = torch.randint(0, V, (B, T), device=DEV)
    idx[0, 0] = V - 1
    assert native_embedding_supported(idx, wte, wpe)
    x = token_position_embedding(idx, wte, wpe)
    pos = torch.arange(T, device=DEV)
    ref = torch.nn.functional.embedding(idx, wte) + torch.nn.functional.embedding(pos, wpe)
    assert x.shape == (B, T, C) and torch.equal(x, ref)
    g = torch.randn_like(x)
    dwte, dwpe = torch.autograd.grad(x, (wte, wpe), g)
    ref_wte = torch.zeros(V, C, device=DEV).index_add_(0, idx.reshape(-1), g.reshape(-1, C).float())
    ref_wpe = torch.zeros(P, C, device=DEV)
    ref_wpe[:T] = g.float().sum(0)
    # fp32 accumulation in a different order, then one bf16 rounding: half a unit in the last place of the result
    for got, want in ((dwte, ref_wte), (dwpe, ref_wpe)):
        assert got.dtype == torch.bfloat16 and got.shape == want.shape
        torch.testing.assert_close(got.float(), want, rtol=2 ** -7, atol=1e-3 * max(1.0, float(want.abs().max())))
    assert torch.count_nonzero(dwpe[T:]) == 0
    absent = torch.ones(V, dtype=torch.bool, device=DEV)
    absent[idx.reshape(-1)] = False
    assert torch.count_nonzero(dwte[absent]) == 0
    again = torch.autograd.grad(token_position_embedding(idx, wte, wpe), (wte, wpe), g)
    assert torch.equal(again[0], dwte) and torch.equal(again[1], dwpe)


@pytest.mark.parametrize("B,T,C,V,frac", [(16, 1024, 768, 50304, 0.5), (4, 300, 64, 7, 0.9), (2, 256, 128, 100, 1.0), (3, 513, 64, 50, 0.3)])
def test_token_embedding_backward_with_very_frequent_tokens(B, T, C, V, frac):
    """Padded / real-text batches: one id takes a large share of all positions (up to all of them), so its run in the
    sorted list spans many 256-position chunks; the chunk partials and their ordered join must give the same sums
    as an fp32 accumulation, bit-identically from run to run, and in a time that does not grow with the run."""
    import time

    from distributed_pytorch_amd.ops.embedding import token_position_embedding

    torch.manual_seed(B + T)
    wte = torch.randn(V, C, device=DEV).bfloat16().requires_grad_()
    wpe = torch.randn(T, C, device=DEV).bfloat16().requires_grad_()
    idx = torch.randint(0, V, (B, T), device=DEV)
    hot = V // 2
    idx[torch.rand(B, T, device=DEV) < frac] = hot                   # the padding / most frequent token
    idx[0, :3] = torch.tensor([0, V - 1, hot], device=DEV)
    g = torch.randn(B, T, C, device=DEV).bfloat16()
    dwte, _ = torch.autograd.grad(token_position_embedding(idx, wte, wpe), (wte, wpe), g)
    ref = torch.zeros(V, C, device=DEV, dtype=torch.float64).index_add_(0, idx.reshape(-1), g.reshape(-1, C).double())
    n_hot = int((idx == hot).sum())
    assert n_hot >= frac * B * T * 0.8
    torch.testing.assert_close(dwte.double(), ref, rtol=2 ** -7, atol=2e-3 * max(1.0, float(ref.abs().max())))
    again, _ = torch.autograd.grad(token_position_embedding(idx, wte, wpe), (wte, wpe), g)
    assert torch.equal(again, dwte)
    if B * T >= 16384:   # 8192 occurrences of one token: still a sub-millisecond backward
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            torch.autograd.grad(token_position_embedding(idx, wte, wpe), (wte, wpe), g)
        torch.cuda.synchronize()
        assert (time.perf_counter() - t0) / 5 < 5e-3


def test_token_position_embedding_fallback_and_range_check(monkeypatch):
    from distributed_pytorch_amd.ops.embedding import native_embedding_supported, token_position_embedding

    wte = torch.randn(10, 12, device=DEV).bfloat16()   # 12 columns: not a whole number of 16-byte vectors
    wpe = torch.randn(4, 12, device=DEV).bfloat16()
    idx = torch.randint(0, 10, (2, 4), device=DEV)
    assert not native_embedding_supported(idx, wte, wpe)
    assert token_position_embedding(idx, wte, wpe).shape == (2, 4, 12)
    f32 = token_position_embedding(idx, torch.randn(10, 16, device=DEV), torch.randn(4, 16, device=DEV))  # fp32 tables: stock path
    assert f32.dtype == torch.float32
    wte, wpe = torch.randn(10, 16, device=DEV).bfloat16(), torch.randn(4, 16, device=DEV).bfloat16()
    bad = idx.clone()
    bad[1, 2] = 10
    clamped = token_position_embedding(bad, wte, wpe)  # no fault: the id is clamped to the last row
    assert torch.equal(clamped[1, 2], (wte[9].float() + wpe[2].float()).bfloat16())
    monkeypatch.setenv("DIST_CHECK_TARGETS", "1")
    with pytest.raises(IndexError):
        token_position_embedding(bad, wte, wpe)


# ----------------------------------------------------------------------------- 3x3 convolution as implicit GEMM
@pytest.mark.parametrize("B,C_in,C_out,H,W", [(2, 64, 64, 8, 8), (3, 64, 128, 7, 9), (4, 128, 64, 14, 14), (2, 256, 256, 5, 5),
                                               (16, 64, 64, 56, 56), (8, 512, 512, 7, 7), (1, 128, 192, 1, 1), (5, 64, 320, 3, 17)])
def test_conv3x3_implicit_gemm_matches_fp32_reference(B, C_in, C_out, H, W):
    """Forward, input gradient and weight gradient of the implicit-GEMM 3x3 convolution against torch's convolution
    in fp32 on the same bf16-rounded operands."""
    from distributed_pytorch_amd.ops.conv import Conv3x3

    torch.manual_seed(B + C_in + H)
    conv = Conv3x3(C_in, C_out, min_channels=64).to(DEV).bfloat16().to(memory_format=torch.channels_last)
    x = torch.randn(B, C_in, H, W, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    assert conv._implicit_ok(x)
    y = conv(x)
    assert y.shape == (B, C_out, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x.detach().float().requires_grad_()
    wr = conv.weight.detach().float().requires_grad_()
    yr = F.conv2d(xr, wr, padding=1)
    yr.backward(gy.float())
    k = 9 * C_in
    torch.testing.assert_close(y.float(), yr, rtol=2e-2, atol=2e-2 * math.sqrt(k) * 0.05 + 1e-2)
    torch.testing.assert_close(x.grad.float(), xr.grad, rtol=2e-2, atol=2e-2 * math.sqrt(9 * C_out) * 0.05 + 1e-2)
    scale = math.sqrt(B * H * W)
    torch.testing.assert_close(conv.weight.grad.float(), wr.grad, rtol=2e-2, atol=2e-2 * scale)
    rel = (conv.weight.grad.float() - wr.grad).norm() / wr.grad.norm()
    assert rel < 1e-2, rel


def test_conv3x3_falls_back_for_strided_and_fp32_layers():
    from distributed_pytorch_amd.ops.conv import Conv3x3

    conv = Conv3x3(64, 64, stride=2, min_channels=64).to(DEV).bfloat16().to(memory_format=torch.channels_last)
    x = torch.randn(2, 64, 8, 8, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    assert not conv._implicit_ok(x) and conv(x).shape == (2, 64, 4, 4)
    conv32 = Conv3x3(64, 64).to(DEV)
    assert not conv32._implicit_ok(torch.randn(2, 64, 8, 8, device=DEV))
