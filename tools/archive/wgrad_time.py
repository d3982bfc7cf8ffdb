"""Scratch: time the weight-gradient kernel on the level-0 shapes of the contrastive step (2 views of 128^3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd.model import train_ops as T
dev = torch.device("cuda:0"); dt = torch.bfloat16
S, n = 128, 2
for c0, c1, cout, s in [(16, 0, 16, 128), (16, 32, 16, 128), (32, 0, 32, 64), (32, 64, 32, 64), (64, 0, 64, 32), (128, 0, 128, 16)]:
    x0 = torch.randn(n, s, s, s, c0, device=dev).to(dt)
    x1 = torch.randn(n, s // 2, s // 2, s // 2, c1, device=dev).to(dt) if c1 else None
    fr = T.new_framed(n, s, s, s, cout, dt, dev)
    T.interior(fr).copy_(torch.randn(n, s, s, s, cout, device=dev))
    for _ in range(3): T.conv_wgrad(fr, x0, x1, c0 + c1, cout)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): T.conv_wgrad(fr, x0, x1, c0 + c1, cout)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * 27 * (c0 + c1) * cout * n * s ** 3
    print(f"AMX_WGRAD_WGS={os.environ.get('AMX_WGRAD_WGS','1024')} wgrad {c0}+{c1}->{cout} @{s}^3 x{n}: {us:7.1f} us  {fl/us/1e6:6.0f} TF")
