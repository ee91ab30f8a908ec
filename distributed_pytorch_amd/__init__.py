"""MI355X-native single-node data-parallel training framework.

Sub-packages:
  parallel  launcher, TCP-store rendezvous, process group (RCCL / gloo), DDP wrapper, sampler
  ops       hand-written gfx950 kernels behind autograd functions: linear(+bias+act),
            softmax cross-entropy(+accuracy), fused AdamW, bucket pack, layer norm
  models    DummyModel, wide MLP, ResNet-50, GPT-2
  utils     checkpointing, timers / tracing, metric logging, debug switches
  bench     throughput and all-reduce bandwidth runners

The top-level ``distributed`` module (repository root) is the drop-in API facade.
"""
__version__ = "0.1.0"
