"""GPT-2 (decoder-only transformer) for the data-parallel benchmark configuration
"GPT-2 small, synthetic 1024-token sequences, bf16".

The reference ships only a two-layer toy model (reference min_DDP.py:41-49); this model
family exists because the framework's throughput bar is defined on it.  Hot ops map to the
native kernels: layer norm, the MLP up-projection with fused bias + GELU epilogue, the loss
(fused softmax cross-entropy + accuracy over a vocabulary padded to a multiple of 64 so
every logits row is 16-byte aligned, gradient written in place over the logits), and the
optimizer (FusedAdamW).  Attention runs on the native flash-attention kernels directly on the
packed q|k|v projection (``attention_impl="sdpa"`` selects torch's kernel instead).

``linear_impl``: ``"native"`` (default) routes every GEMM of the model -- projections, the fused MLP /
attention / lm-head nodes, all weight and input gradients -- through the hand-written MFMA GEMM;
``"library"`` sends the plain GEMMs to the vendor library for A/B measurements; ``"stock"`` is
unmodified torch (``F.linear`` / ``F.gelu``), the comparison baseline.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.attention import causal_attention_qkvpacked, projected_causal_attention
from ..ops.cross_entropy import projection_cross_entropy, softmax_cross_entropy
from ..ops.embedding import token_position_embedding
from ..ops.layernorm import LayerNorm
from ..ops.linear import library_linear_bias_act, linear_bias_act
from ..ops.mlp import fused_mlp


@dataclass
class GPT2Config:
    block_size: int = 1024
    vocab_size: int = 50257
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 768
    bias: bool = True
    vocab_pad_multiple: int = 64
    linear_impl: str = "native"   # native | library (A/B only) | stock
    inplace_loss_backward: bool = True
    native_ops: bool = True       # False: stock torch layer norm / GELU / cross-entropy (comparison baseline)
    attention_impl: str = "auto"  # auto | native | sdpa
    fuse_add_norm: bool = True    # fold each residual add into the layer norm that follows it
    fused_lm_head_loss: bool = True  # lm-head GEMM + loss + logits gradient as one autograd node
    fused_mlp: bool = True        # c_fc -> GELU -> c_proj as one autograd node (fused backward epilogue)
    fused_qkv_attention: bool = True  # c_attn + attention as one node (bias gradient out of the attention backward)

    @property
    def padded_vocab_size(self) -> int:
        m = self.vocab_pad_multiple
        return (self.vocab_size + m - 1) // m * m

    @staticmethod
    def small(**kw) -> "GPT2Config":
        return GPT2Config(n_layer=12, n_head=12, n_embd=768, **kw)

    @staticmethod
    def medium(**kw) -> "GPT2Config":
        return GPT2Config(n_layer=24, n_head=16, n_embd=1024, **kw)


class _Proj(nn.Linear):
    """Projection whose GEMM implementation is chosen by the model config."""

    def __init__(self, in_f, out_f, bias, impl: str, act: Optional[str] = None):
        super().__init__(in_f, out_f, bias=bias)
        self.impl = impl
        self.act = act

    def forward(self, x, skip_bias: bool = False):
        """``skip_bias``: leave the bias out (the caller adds it inside the next fused add + layer norm)."""
        bias = None if skip_bias else self.bias
        if self.impl == "native" and x.is_cuda:
            return linear_bias_act(x, self.weight, bias, self.act)
        if self.impl == "library" and x.is_cuda:
            return library_linear_bias_act(x, self.weight, bias, self.act)
        y = F.linear(x, self.weight, bias)
        if self.act == "gelu":
            y = F.gelu(y)
        elif self.act == "relu":
            y = F.relu(y)
        return y


class CausalSelfAttention(nn.Module):
    def __init__(self, cfg: GPT2Config):
        super().__init__()
        assert cfg.n_embd % cfg.n_head == 0
        self.n_head = cfg.n_head
        self.impl = cfg.attention_impl if cfg.native_ops else "sdpa"
        self.gemm_impl = cfg.linear_impl
        self.fused = cfg.fused_qkv_attention and cfg.native_ops and cfg.linear_impl in ("native", "library") and cfg.bias and self.impl != "sdpa"
        self.c_attn = _Proj(cfg.n_embd, 3 * cfg.n_embd, cfg.bias, cfg.linear_impl)
        self.c_proj = _Proj(cfg.n_embd, cfg.n_embd, cfg.bias, cfg.linear_impl)

    def forward(self, x, skip_bias: bool = False):
        y = projected_causal_attention(x, self.c_attn.weight, self.c_attn.bias, self.n_head, impl=self.gemm_impl) if self.fused else None
        if y is None:
            y = causal_attention_qkvpacked(self.c_attn(x), self.n_head, impl=self.impl)
        return self.c_proj(y, skip_bias)


class MLP(nn.Module):
    def __init__(self, cfg: GPT2Config):
        super().__init__()
        self.gemm_impl = cfg.linear_impl
        self.fused = cfg.fused_mlp and cfg.native_ops and cfg.linear_impl in ("native", "library") and cfg.bias
        self.c_fc = _Proj(cfg.n_embd, 4 * cfg.n_embd, cfg.bias, cfg.linear_impl, act="gelu")
        self.c_proj = _Proj(4 * cfg.n_embd, cfg.n_embd, cfg.bias, cfg.linear_impl)

    def forward(self, x, skip_bias: bool = False):
        if self.fused and x.is_cuda:
            # both layers as one autograd node: the backward fuses GELU' and the c_fc bias gradient into
            # the epilogue of c_proj's input-gradient GEMM
            return fused_mlp(x, self.c_fc.weight, self.c_fc.bias, self.c_proj.weight, None if skip_bias else self.c_proj.bias, "gelu",
                             impl=self.gemm_impl)
        return self.c_proj(self.c_fc(x), skip_bias)


class Block(nn.Module):
    def __init__(self, cfg: GPT2Config):
        super().__init__()
        LN = LayerNorm if cfg.native_ops else nn.LayerNorm
        self.ln_1 = LN(cfg.n_embd)
        self.attn = CausalSelfAttention(cfg)
        self.ln_2 = LN(cfg.n_embd)
        self.mlp = MLP(cfg)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        x = x + self.mlp(self.ln_2(x))
        return x

    def forward_deferred(self, x, pending, pending_bias):
        """Same block with every residual add folded into the layer norm that follows it: takes the
        residual stream ``x`` plus the not-yet-added output ``pending`` of the previous branch (whose
        output-projection bias ``pending_bias`` is still to be added as well) and returns the new
        stream, this block's own pending MLP output and that output's bias.  The projections run
        without their bias: it is added inside the next add + layer-norm kernel, whose backward
        also produces the bias gradient, so neither a bias epilogue nor a column-sum pass exists."""
        x, h = self.ln_1.forward_add(x, pending, pending_bias)
        a = self.attn(h, skip_bias=True)
        x, h = self.ln_2.forward_add(x, a, self.attn.c_proj.bias)
        return x, self.mlp(h, skip_bias=True), self.mlp.c_proj.bias


class GPT2(nn.Module):
    def __init__(self, cfg: GPT2Config):
        super().__init__()
        self.cfg = cfg
        V = cfg.padded_vocab_size
        self.wte = nn.Embedding(V, cfg.n_embd)
        self.wpe = nn.Embedding(cfg.block_size, cfg.n_embd)
        self.h = nn.ModuleList([Block(cfg) for _ in range(cfg.n_layer)])
        self.ln_f = (LayerNorm if cfg.native_ops else nn.LayerNorm)(cfg.n_embd)
        # lm_head shares the token-embedding matrix (weight tying), so it has no parameter of its own
        self.apply(self._init_weights)
        for name, p in self.named_parameters():
            if name.endswith("c_proj.weight"):
                nn.init.normal_(p, mean=0.0, std=0.02 / math.sqrt(2 * cfg.n_layer))
        self.last_correct = None
        self.last_metrics = None

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, mean=0.0, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=0.02)

    def num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def flops_per_token(self, seq_len: int, causal: bool = True) -> float:
        """Training FLOPs per token: 6 * (non-embedding params + lm_head) + attention.  The attention term is
        12 * L * C * T when every query is charged the full sequence (the usual "model FLOPs" convention) and
        half of that for the products a causal kernel really executes (``causal=True``, the default: what the
        hardware does here)."""
        cfg = self.cfg
        n = sum(p.numel() for name, p in self.named_parameters() if not name.startswith("wpe"))
        attn = 12.0 * cfg.n_layer * cfg.n_embd * seq_len
        return 6.0 * n + (0.5 * attn if causal else attn)

    def forward(self, idx: torch.Tensor, targets: Optional[torch.Tensor] = None):
        B, T = idx.shape
        assert T <= self.cfg.block_size
        if self.cfg.native_ops:
            x = token_position_embedding(idx, self.wte.weight, self.wpe.weight)  # one kernel; stock operators off the GPU
        else:
            pos = torch.arange(T, device=idx.device)
            x = self.wte(idx) + self.wpe(pos)
        if self.cfg.native_ops and self.cfg.fuse_add_norm:
            pending = pending_bias = None
            for blk in self.h:
                x, pending, pending_bias = blk.forward_deferred(x, pending, pending_bias)
            _, x = self.ln_f.forward_add(x, pending, pending_bias)
        else:
            for blk in self.h:
                x = blk(x)
            x = self.ln_f(x)
        if targets is None:
            return F.linear(x[:, -1:, :], self.wte.weight)[..., : self.cfg.vocab_size]
        if not self.cfg.native_ops:
            logits = F.linear(x, self.wte.weight)
            lg = logits.view(B * T, -1)[:, : self.cfg.vocab_size]
            return F.cross_entropy(lg.float(), targets.reshape(-1))
        if self.cfg.fused_lm_head_loss:
            # vocabulary projection + loss + d(logits) as one node: the logits cross HBM twice, not three times
            r = projection_cross_entropy(x, self.wte.weight, targets, n_classes=self.cfg.vocab_size,
                                         impl="library" if self.cfg.linear_impl == "library" else "native")
        else:
            logits = linear_bias_act(x, self.wte.weight) if self.cfg.linear_impl == "native" else F.linear(x, self.wte.weight)
            r = softmax_cross_entropy(logits.view(B * T, -1), targets.reshape(-1), n_classes=self.cfg.vocab_size,
                                      inplace_backward=self.cfg.inplace_loss_backward)
        self.last_correct, self.last_metrics = r.correct, r.metrics
        return r.loss
