"""What streaming bandwidth does the chip deliver to the simplest kernels?  (ceiling for the layer-norm and loss kernels)"""
import torch
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for rows in (65536, 262144, 393216):
    a = torch.randn(rows, 768, device="cuda").bfloat16()
    b = torch.randn_like(a); c = torch.empty_like(a); d = torch.empty_like(a)
    nbytes = a.numel() * 2
    t_copy = timeit(lambda: c.copy_(a))
    t_add = timeit(lambda: torch.add(a, b, out=c))
    t_fill = timeit(lambda: c.zero_())
    t_read = timeit(lambda: a.sum(dtype=torch.float32))
    print(f"{rows} x 768 bf16 ({nbytes/1e6:.0f} MB): copy {2*nbytes/t_copy/1e12:.2f} TB/s, add (2 reads + 1 write) {3*nbytes/t_add/1e12:.2f} TB/s, "
          f"fill {nbytes/t_fill/1e12:.2f} TB/s, sum (read only) {nbytes/t_read/1e12:.2f} TB/s", flush=True)
big = torch.empty(393216 * 50304 // 4, device="cuda", dtype=torch.bfloat16)  # 9.9 GB
big2 = torch.empty_like(big)
nb = big.numel() * 2
t = timeit(lambda: big2.copy_(big), n=5)
print(f"9.9 GB copy: {2*nb/t/1e12:.2f} TB/s")
t = timeit(lambda: big.mul_(1.0001), n=5)
print(f"9.9 GB in-place scale (1 read + 1 write): {2*nb/t/1e12:.2f} TB/s")
