"""1x1 convolutions of ResNet-50 (batch 256, bf16, channels last): stock convolution vs the GEMM path, forward + backward."""
import torch

from distributed_pytorch_amd.ops import PointwiseConv2d
from distributed_pytorch_amd.ops.linear import native_route


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dev = torch.device("cuda")
    shapes = [(56, 64, 64), (56, 64, 256), (56, 256, 64), (28, 512, 128), (28, 128, 512), (14, 1024, 256), (14, 256, 1024), (7, 2048, 512),
              (7, 512, 2048)]
    for hw, cin, cout in shapes:
        x = torch.randn(256, cin, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
        ours = PointwiseConv2d(cin, cout).to(dev).bfloat16().to(memory_format=torch.channels_last)
        stock = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev).bfloat16().to(memory_format=torch.channels_last)
        go = torch.randn(256, cout, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)

        def run(m):
            def f():
                x.grad = None
                m.weight.grad = None
                m(x).backward(go)
            return f

        rows = 256 * hw * hw
        flops = 3 * 2.0 * rows * cin * cout
        t_o, t_s = timeit(run(ours)), timeit(run(stock))
        print(f"{hw:3d}x{hw:<3d} {cin:5d}->{cout:<5d} route {native_route(rows, cout, cin, torch.bfloat16):12s} gemm path {t_o:8.1f} us ({flops / t_o / 1e6:6.0f} TF)"
              f"   stock conv {t_s:8.1f} us ({flops / t_s / 1e6:6.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
