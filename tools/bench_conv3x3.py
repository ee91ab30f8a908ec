"""3x3 stride-1 convolution, forward + backward, implicit-GEMM path against the stock (MIOpen) convolution on the
ResNet-50 stage shapes at batch 256, bf16, channels-last.

    PYTHONPATH=. python tools/bench_conv3x3.py
"""
import torch
import torch.nn as nn

from distributed_pytorch_amd.ops.conv import Conv3x3


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    dev = "cuda"
    for c, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
        x = torch.randn(256, c, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
        ours = Conv3x3(c, c).to(dev).bfloat16().to(memory_format=torch.channels_last)
        stock = nn.Conv2d(c, c, 3, padding=1, bias=False).to(dev).bfloat16().to(memory_format=torch.channels_last)
        gy = torch.randn(256, c, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)

        def run(m):
            def f():
                x.grad = None
                m.weight.grad = None
                m(x).backward(gy)
            return f

        def fwd(m):
            def f():
                with torch.no_grad():
                    m(x)
            return f

        flops = 2 * 256 * hw * hw * 9 * c * c
        t_o, t_s = timed(run(ours)), timed(run(stock))
        f_o, f_s = timed(fwd(ours)), timed(fwd(stock))
        print(f"C={c:4d} {hw}x{hw}: implicit GEMM fwd {f_o:7.1f} us ({flops / f_o / 1e6:6.1f} TF/s)  fwd+bwd {t_o:7.1f} us | "
              f"stock fwd {f_s:7.1f} us  fwd+bwd {t_s:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
