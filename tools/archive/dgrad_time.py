"""Times the data gradient of a conv block standalone: framed route (conv on (n+4)^3 + pad_fold) against the direct route (interior + shell).
   python tools/dgrad_time.py [cin=16] [cout=16] [size=128] [n=2]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd.model import train_ops as T
cin, cout, S, n = [int(a) for a in (sys.argv[1:5] + ["16", "16", "128", "2"][len(sys.argv) - 1:])]
dev = torch.device("cuda:0")
dt = torch.bfloat16
fr = T.new_framed(n, S, S, S, cout, dt, dev)
T.interior(fr).copy_(torch.randn(n, S, S, S, cout, device=dev).to(dt))
w = (torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5)
def timeit(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
print(f"{cout}->{cin} dgrad @{S}^3 x{n}: framed conv {timeit(lambda: T.conv_dgrad_framed(fr, w)):.1f} us, + pad_fold {timeit(lambda: T.pad_fold(T.conv_dgrad_framed(fr, w))):.1f} us; "
      f"direct {timeit(lambda: T.conv_dgrad_direct(fr, w)):.1f} us")
import ctypes
from anatomix_amd import _lib
lib = _lib.load()
out = T.conv_dgrad_direct(fr, w)
tab = torch.empty(lib.amx_conv3d_dgrad_shell_scratch_bytes(), dtype=torch.uint8, device=dev)
def shell():
    _lib.check(lib.amx_conv3d_dgrad_fold_shell(_lib.ptr(fr), cout, _lib.ptr(w), cout, cin, _lib.ptr(out), out.shape[-1], n, S, S, S, 1, _lib.ptr(tab), T._st(dev)))
print(f"shell kernel alone {timeit(shell):.1f} us")
