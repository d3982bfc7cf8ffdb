"""Scratch: what HBM sustains for the access mixes of the level-0 kernels (torch ops, 268 MB tensors): write-only, read-only, copy."""
import torch
dev = torch.device("cuda:0")
n = 268435456 // 2
x = torch.randn(n, device=dev, dtype=torch.float16); y = torch.empty_like(x); z = torch.empty(2 * n, device=dev, dtype=torch.float16)
def t(f, K=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
mb = n * 2 / 1e6
for name, f, moved in [("write-only  fill_ 268 MB", lambda: y.fill_(1.0), mb), ("write-only  zero_ 537 MB", lambda: z.zero_(), 2 * mb),
                       ("read-only   sum 268 MB", lambda: x.sum(), mb), ("read-only   amax 268 MB", lambda: x.amax(), mb),
                       ("copy        268 MB -> 268 MB", lambda: y.copy_(x), 2 * mb),
                       ("1 read : 2 write (f16 -> f32 convert)", lambda: torch.empty(n, device=dev, dtype=torch.float32).copy_(x), 3 * mb)]:
    us = t(f)
    print(f"{name:42s} {us:7.1f} us  {moved / us:6.2f} TB/s")
