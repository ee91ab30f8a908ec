"""Experiment: does chunking the lm-head GEMM + fused loss over tokens (so a chunk's logits are still in the
256 MiB Infinity Cache when the loss kernel reads them) beat one GEMM + one loss launch over all tokens?

    PYTHONPATH=. python tools/exp_lmhead_chunk.py [tokens=131072]
"""
import sys

import torch

from distributed_pytorch_amd import _ext


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


def main():
    C = _ext.C()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    V, K = 50304, 768
    dev = "cuda"
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(V, K, device=dev) * 0.02).to(torch.bfloat16)
    tgt = torch.randint(0, 50257, (M,), device=dev)
    big = torch.empty(M, V, device=dev, dtype=torch.bfloat16)

    def whole():
        C.gemm(x, w, True, True, None, 0, big)
        C.xent_fused(big, tgt, 50257, -100, False)

    def gemm_only():
        C.gemm(x, w, True, True, None, 0, big)

    def loss_only():
        C.xent_fused(big, tgt, 50257, -100, False)

    print(f"tokens {M}: whole {timed(whole):.2f} ms (gemm {timed(gemm_only):.2f}, loss {timed(loss_only):.2f})", flush=True)
    for mc in (1024, 1536, 2048, 3072, 4096, 8192):
        def chunked():
            for m0 in range(0, M, mc):
                sl = big[m0:m0 + mc]
                C.gemm(x[m0:m0 + mc], w, True, True, None, 0, sl)
                C.xent_fused(sl, tgt[m0:m0 + mc], 50257, -100, False)

        def chunked_one_buffer():   # every chunk in the same buffer: lines are overwritten while still cached
            for m0 in range(0, M, mc):
                n = min(mc, M - m0)
                sl = big[:n]
                C.gemm(x[m0:m0 + n], w, True, True, None, 0, sl)
                C.xent_fused(sl, tgt[m0:m0 + n], 50257, -100, False)

        print(f"  chunk {mc}: separate slices {timed(chunked):.2f} ms, one reused buffer {timed(chunked_one_buffer):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
