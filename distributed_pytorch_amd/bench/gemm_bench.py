"""GEMM micro-benchmark: the hand-written MFMA kernels against the vendor library on the
shapes the models actually run, on random data, variants interleaved round by round in one
process (cross-process timing spread on this GPU is larger than the differences of interest).

    python -m distributed_pytorch_amd.bench.gemm_bench [--dtype bf16|fp32] [--rounds 5] [--json out.json]
"""
from __future__ import annotations

import argparse
import json
import statistics

import torch

from .. import _ext

GPT2_TOKENS = 65536


def model_shapes(tokens=GPT2_TOKENS):
    return [s if not s[0].startswith("gpt2") else (s[0],) + tuple(tokens if d == GPT2_TOKENS else d for d in s[1:4]) + s[4:] for s in SHAPES]


SHAPES = [
    # (name, M, N, K, a_kmajor, b_kmajor)
    ("sq4096 NT", 4096, 4096, 4096, True, True),
    ("sq8192 NT", 8192, 8192, 8192, True, True),
    ("gpt2 qkv fwd", GPT2_TOKENS, 2304, 768, True, True),
    ("gpt2 attn-proj fwd", GPT2_TOKENS, 768, 768, True, True),
    ("gpt2 fc fwd", GPT2_TOKENS, 3072, 768, True, True),
    ("gpt2 mlp-proj fwd", GPT2_TOKENS, 768, 3072, True, True),
    ("gpt2 fc dgrad", GPT2_TOKENS, 768, 3072, True, False),
    ("gpt2 mlp-proj dgrad", GPT2_TOKENS, 3072, 768, True, False),
    ("gpt2 fc wgrad", 3072, 768, GPT2_TOKENS, False, False),
    ("gpt2 mlp-proj wgrad", 768, 3072, GPT2_TOKENS, False, False),
    ("gpt2 lm-head fwd", GPT2_TOKENS, 50304, 768, True, True),
    ("widemlp fwd", 8192, 8192, 8192, True, True),
    ("widemlp dgrad", 8192, 8192, 8192, True, False),
    ("widemlp wgrad", 8192, 8192, 8192, False, False),
]


def _time(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def run(dtype=torch.bfloat16, rounds=5, iters=5, shapes=SHAPES):
    C = _ext.C()
    dev = torch.device("cuda")
    out = []
    for name, M, N, K, ak, bk in shapes:
        A = torch.randn((M, K) if ak else (K, M), device=dev).to(dtype)
        B = torch.randn((N, K) if bk else (K, N), device=dev).to(dtype)
        native = lambda: C.gemm(A, B, ak, bk)
        Aop = A if ak else A.t()
        Bop = B.t() if bk else B
        library = lambda: torch.matmul(Aop, Bop)
        ref = library()
        got = native()
        err = float((got[:4096].float() - ref[:4096].float()).abs().max() / ref[:4096].float().abs().max().clamp(min=1e-6))
        for f in (native, library):
            f()
        tn, tl = [], []
        for _ in range(rounds):
            tn.append(_time(native, iters))
            tl.append(_time(library, iters))
        fl = 2.0 * M * N * K
        rec = dict(shape=name, M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk,
                   native_ms=round(statistics.median(tn), 4), library_ms=round(statistics.median(tl), 4),
                   native_tflops=round(fl / statistics.median(tn) / 1e9, 1), library_tflops=round(fl / statistics.median(tl) / 1e9, 1),
                   native_best_tflops=round(fl / min(tn) / 1e9, 1), rel_err=round(err, 5))
        out.append(rec)
        print(json.dumps(rec), flush=True)
        del A, B, ref, got
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default=None, help="substring filter on shape names")
    ap.add_argument("--json", default=None)
    ap.add_argument("--tokens", type=int, default=GPT2_TOKENS, help="rows of the GPT-2 activations (batch x sequence)")
    a = ap.parse_args()
    shapes = [s for s in model_shapes(a.tokens) if a.only is None or a.only in s[0]]
    if a.dtype == "fp32":
        shapes = [s for s in shapes if s[1] * s[2] * s[3] <= 8192 ** 3 and "lm-head" not in s[0]]
    res = run(torch.bfloat16 if a.dtype == "bf16" else torch.float32, a.rounds, shapes=shapes)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
