"""Model-level checks: the GPT-2 used by the benchmark computes the same function on every kernel
configuration (native / library / stock GEMMs, fused or separate residual-add + layer norm, native
or torch attention), on CPU and on the GPU."""
import copy

import pytest
import torch

from distributed_pytorch_amd.models.gpt2 import GPT2, GPT2Config


def _tiny(**kw):
    base = dict(block_size=256, vocab_size=1000, n_layer=2, n_head=2, n_embd=128)
    base.update(kw)
    return GPT2Config(**base)


def _batch(cfg, B=2, T=128, device="cpu", seed=0):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, cfg.vocab_size, (B, T), generator=g).to(device)
    tgt = torch.randint(0, cfg.vocab_size, (B, T), generator=g).to(device)
    return idx, tgt


def test_gpt2_cpu_native_ops_fall_back_to_the_stock_function():
    torch.manual_seed(0)
    a = GPT2(_tiny())
    b = GPT2(_tiny(native_ops=False))
    b.load_state_dict(a.state_dict())
    idx, tgt = _batch(a.cfg)
    la, lb = a(idx, tgt), b(idx, tgt)
    torch.testing.assert_close(la, lb, rtol=1e-5, atol=1e-6)
    la.backward()
    lb.backward()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-4, atol=1e-6, msg=n)
    # weight tying: one matrix serves the embedding and the lm head; padded vocabulary rows get no gradient
    assert a.wte.weight.shape[0] == a.cfg.padded_vocab_size == 1024
    assert torch.count_nonzero(a.wte.weight.grad[a.cfg.vocab_size:]) == 0
    # inference path returns logits of the last position over the real vocabulary
    out = a(idx)
    assert out.shape == (2, 1, a.cfg.vocab_size)
    assert a.last_correct is not None and a.last_metrics.shape == (4,)


def test_gpt2_deferred_residual_structure_equals_plain_blocks():
    torch.manual_seed(1)
    a = GPT2(_tiny(fuse_add_norm=True))
    b = GPT2(_tiny(fuse_add_norm=False))
    b.load_state_dict(a.state_dict())
    idx, tgt = _batch(a.cfg, seed=3)
    torch.testing.assert_close(a(idx, tgt), b(idx, tgt), rtol=1e-6, atol=1e-6)


def _rel_l2(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["native", "library", "no_fuse_add_norm", "sdpa", "unfused_nodes"])
def test_gpt2_gpu_kernel_configurations_match_stock_torch(variant):
    """bf16 on the GPU: every configuration of native kernels against unmodified torch ops with the
    same weights -- loss within bf16 tolerance, every parameter gradient close in relative L2 norm."""
    dev = torch.device("cuda")
    torch.manual_seed(0)
    kw = dict(linear_impl="native")
    if variant == "library":
        kw = dict(linear_impl="library")
    elif variant == "no_fuse_add_norm":
        kw["fuse_add_norm"] = False
    elif variant == "sdpa":
        kw["attention_impl"] = "sdpa"
    elif variant == "unfused_nodes":
        kw.update(fused_mlp=False, fused_qkv_attention=False, fused_lm_head_loss=False)
    ours = GPT2(_tiny(**kw)).to(dev, torch.bfloat16)
    stock = GPT2(_tiny(linear_impl="stock", native_ops=False)).to(dev, torch.bfloat16)
    stock.load_state_dict(ours.state_dict())
    idx, tgt = _batch(ours.cfg, B=4, T=256, device=dev)
    lo, ls = ours(idx, tgt), stock(idx, tgt)
    assert abs(float(lo.detach()) - float(ls.detach())) < 1e-2 * abs(float(ls.detach()))
    lo.backward()
    ls.backward()
    for (n, p), q in zip(ours.named_parameters(), stock.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad.float()).all(), n
        # both sides compute in bf16 with different summation orders: compare in relative L2 norm
        assert _rel_l2(p.grad.float(), q.grad.float()) < 6e-2, (n, _rel_l2(p.grad.float(), q.grad.float()))


@pytest.mark.gpu
def test_gpt2_benchmark_width_matches_fp32_torch():
    """The benchmark's layer shapes (768 wide, 12 heads, 1024 tokens; two layers, eight sequences) on the
    native kernels in bf16 against stock torch ops in fp32 on the same weights: loss within 1 %, every
    parameter gradient within 3 % in relative L2 norm."""
    dev = torch.device("cuda")
    torch.manual_seed(0)
    cfg = dict(block_size=1024, vocab_size=50257, n_layer=2, n_head=12, n_embd=768)
    ours = GPT2(GPT2Config(**cfg)).to(dev, torch.bfloat16)
    ref = GPT2(GPT2Config(linear_impl="stock", native_ops=False, **cfg)).to(dev, torch.float32)
    ref.load_state_dict({k: v.float() for k, v in ours.state_dict().items()})
    idx, tgt = _batch(ours.cfg, B=8, T=1024, device=dev)
    lo, lr_ = ours(idx, tgt), ref(idx, tgt)
    assert abs(float(lo.detach()) - float(lr_.detach())) < 1e-2 * abs(float(lr_.detach())), (float(lo), float(lr_))
    lo.backward()
    lr_.backward()
    worst = max((_rel_l2(p.grad.float(), q.grad), n) for (n, p), q in zip(ours.named_parameters(), ref.parameters()))
    assert worst[0] < 3e-2, worst


@pytest.mark.gpu
def test_gpt2_gpu_training_steps_track_stock_torch():
    """Five optimizer steps: native kernels + FusedAdamW (fp32 master weights) against torch ops +
    torch.optim.AdamW on an fp32 copy -- the loss curves stay together and fall."""
    from distributed_pytorch_amd.ops import FusedAdamW

    dev = torch.device("cuda")
    torch.manual_seed(0)
    ours = GPT2(_tiny()).to(dev, torch.bfloat16)
    ref = GPT2(_tiny(linear_impl="stock", native_ops=False)).to(dev, torch.float32)
    ref.load_state_dict({k: v.float() for k, v in ours.state_dict().items()})
    oo = FusedAdamW(ours.parameters(), lr=1e-3, weight_decay=0.01, master_weights=True)
    ro = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)
    idx, tgt = _batch(ours.cfg, B=4, T=256, device=dev)
    lo, lr_ = [], []
    for _ in range(5):
        for m, o, acc in ((ours, oo, lo), (ref, ro, lr_)):
            loss = m(idx, tgt)
            o.zero_grad(set_to_none=True)
            loss.backward()
            o.step()
            acc.append(float(loss.detach()))
    assert lo[-1] < lo[0] and lr_[-1] < lr_[0]
    for a, b in zip(lo, lr_):
        assert abs(a - b) < 5e-2 * abs(b), (lo, lr_)


def test_batchnorm_module_cpu_fallback_matches_stock_module():
    """Off the GPU the module is torch's batch norm followed by the add and the ReLU, with identical state."""
    import torch.nn.functional as F

    from distributed_pytorch_amd.ops import BatchNorm2d

    torch.manual_seed(0)
    x = torch.randn(4, 16, 5, 5)
    r = torch.randn(4, 16, 5, 5)
    a, b = BatchNorm2d(16), torch.nn.BatchNorm2d(16)
    b.load_state_dict(a.state_dict())
    ya = a(x, residual=r, relu=True)
    yb = F.relu(b(x) + r)
    torch.testing.assert_close(ya, yb)
    torch.testing.assert_close(a.running_var, b.running_var)
    assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1
    a.eval(); b.eval()
    torch.testing.assert_close(a(x), b(x))


def test_pointwise_conv_cpu_fallback_and_state_dict():
    from distributed_pytorch_amd.ops import PointwiseConv2d

    torch.manual_seed(1)
    a, b = PointwiseConv2d(8, 12, stride=2), torch.nn.Conv2d(8, 12, 1, stride=2, bias=False)
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 8, 6, 6)
    torch.testing.assert_close(a(x), b(x))
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
