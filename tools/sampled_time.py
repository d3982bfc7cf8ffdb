"""Time amx_conv3d_backward_sampled at the contrastive step's shape (2 views of 128^3, 512 patches, 16 -> 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd.model import train_ops as T
dev = torch.device("cuda:0"); dt = torch.bfloat16
n, s, p = 2, 128, 512
x0 = torch.randn(n, s, s, s, 16, device=dev).to(dt)
w = torch.randn(16, 16, 3, 3, 3, device=dev)
rows = torch.randn(n, p, 16, device=dev)
flat = torch.randperm(s ** 3, device=dev)[:p]
coords = torch.stack([flat // (s * s), (flat // s) % s, flat % s], 1)
for need in (False, True):
    for _ in range(3): T.conv_backward_sampled(rows, coords, x0, w, 16, need)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): T.conv_backward_sampled(rows, coords, x0, w, 16, need)
    e1.record(); torch.cuda.synchronize()
    print("need_din", need, f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
