"""HBM streaming rates seen by simple torch kernels (context for the roofline numbers)."""
import torch, time
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n = 4 * 16 * 128 ** 3
y = torch.empty(n, dtype=torch.float32, device=dev)
x16 = torch.empty(n, dtype=torch.float16, device=dev)
z16 = torch.empty(n, dtype=torch.float16, device=dev)
us = t(lambda: y.fill_(1.0)); print(f"fill fp32 {n*4/1e6:.0f} MB: {us:.1f} us  {n*4/us/1e3:.0f} GB/s write")
us = t(lambda: y.zero_()); print(f"zero fp32 {n*4/1e6:.0f} MB: {us:.1f} us  {n*4/us/1e3:.0f} GB/s write")
us = t(lambda: z16.copy_(x16)); print(f"copy f16 {n*2/1e6:.0f} MB: {us:.1f} us  {n*4/us/1e3:.0f} GB/s r+w")
us = t(lambda: y.copy_(x16)); print(f"f16->fp32 read {n*2/1e6:.0f} MB write {n*4/1e6:.0f} MB: {us:.1f} us  {n*6/us/1e3:.0f} GB/s r+w")
us = t(lambda: torch.sum(x16)); print(f"sum f16 {n*2/1e6:.0f} MB: {us:.1f} us  {n*2/us/1e3:.0f} GB/s read")
y2 = torch.empty_like(y)
us = t(lambda: y2.copy_(y)); print(f"copy fp32 {n*4/1e6:.0f} MB: {us:.1f} us  {n*8/us/1e3:.0f} GB/s r+w")
