"""Time conv_wgrad (kernel + its reduce) on every conv shape of the 6 M UNet's contrastive step (2 views of 128^3, bf16) and check
each against an fp64 reference on a sub-problem.  AMX_LIB_PATH selects the library; run with `gpurun`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd.model import train_ops as T

dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("WG_DT", "bf16") == "bf16" else torch.float16
n = int(os.environ.get("WG_N", "2"))
SHAPES = [(16, 0, 16, 128), (16, 32, 16, 128), (16, 0, 32, 64), (32, 0, 32, 64), (32, 64, 32, 64), (32, 0, 64, 32), (64, 0, 64, 32),
          (64, 128, 64, 32), (64, 0, 128, 16), (128, 0, 128, 16), (128, 256, 128, 16), (128, 0, 256, 8), (256, 0, 256, 8)]
tot = 0.0
if os.environ.get("WG_ONLY"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["WG_ONLY"].split(",")]
for c0, c1, cout, s in SHAPES:
    torch.manual_seed(c0 + c1 + cout + s)
    x0 = torch.randn(n, s, s, s, c0, device=dev).to(dt)
    x1 = torch.randn(n, s // 2, s // 2, s // 2, c1, device=dev).to(dt) if c1 else None
    fr = T.new_framed(n, s, s, s, cout, dt, dev)
    T.interior(fr).copy_(torch.randn(n, s, s, s, cout, device=dev))
    for _ in range(3):
        dw = T.conv_wgrad(fr, x0, x1, c0 + c1, cout)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("WG_REPS", "20"))
    e0.record()
    for _ in range(reps):
        T.conv_wgrad(fr, x0, x1, c0 + c1, cout)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * 27 * (c0 + c1) * cout * n * s ** 3
    if os.environ.get("WG_NOCHECK"):
        tot += us
        print(f"wgrad {c0:3d}+{c1:3d}->{cout:3d} @{s:3d}^3 x{n}: {us:7.1f} us  {fl/us/1e6:6.0f} TF", flush=True)
        continue
    # reference: a few output channels / input channels in fp64 through torch's own conv weight gradient
    co_s, ci_s = [0, cout - 1, cout // 2], sorted({0, c0 - 1, c0 + c1 - 1, (c0 + c1) // 2})
    xin = x0.double()
    if c1:
        up = x1.double().repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
        xin = torch.cat((xin, up), -1)
    xin = xin[..., ci_s].permute(0, 4, 1, 2, 3)
    xp = torch.nn.functional.pad(xin, (1,) * 6, mode="reflect")
    g = T.interior(fr).double()[..., co_s].permute(0, 4, 1, 2, 3)
    ref = torch.zeros(len(co_s), len(ci_s), 3, 3, 3, dtype=torch.float64, device=dev)
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                ref[:, :, kz, ky, kx] = torch.einsum("nozyx,nizyx->oi", g, xp[:, :, kz:kz + s, ky:ky + s, kx:kx + s])
    got = dw[co_s][:, ci_s].double()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    tot += us
    print(f"wgrad {c0:3d}+{c1:3d}->{cout:3d} @{s:3d}^3 x{n}: {us:7.1f} us  {fl/us/1e6:6.0f} TF   err {err:.2e}", flush=True)
print(f"sum {tot:.1f} us")
