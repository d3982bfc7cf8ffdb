"""Scratch: time the two launches of the merged concat conv separately (amx_conv3d_upcat_merged vs the skip conv alone)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd import _lib
c0, c1, S = [int(a) for a in sys.argv[1:4]]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cout = c0
dev = torch.device("cuda:0"); lib = _lib.load()
x0 = torch.randn(n, S, S, S, c0, device=dev).relu().half()
x1 = torch.randn(n, S // 2, S // 2, S // 2, c1, device=dev).relu().half()
w = (torch.randn(cout, c0 + c1, 27, device=dev) / (27 * (c0 + c1)) ** 0.5).float()
sh = torch.zeros(cout, device=dev)
wpk = torch.empty(max(lib.amx_conv3d_upcat_merged_packed_bytes(c0, c1, cout), lib.amx_conv3d_packed_bytes(c0 + c1, cout)), dtype=torch.uint8, device=dev)
part = torch.empty(n * S ** 3 * cout, device=dev, dtype=torch.half)
out = torch.empty(n, S, S, S, cout, device=dev, dtype=torch.half)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def merged():
    _lib.check(lib.amx_conv3d_upcat_merged(_lib.ptr(x0), c0, _lib.ptr(x1), c1, _lib.ptr(w), None, _lib.ptr(sh), cout, n, S, S, S, 1, 0.3, 0,
                                           _lib.ptr(wpk), _lib.ptr(part), _lib.ptr(out), st))
def plain():
    _lib.check(lib.amx_conv3d_k3_reflect(_lib.ptr(x0), c0, _lib.ptr(x1), c1, _lib.ptr(w), None, _lib.ptr(sh), cout, n, S, S, S, 1, 0.3, 0,
                                         _lib.ptr(wpk), _lib.ptr(out), None, st))
def skip_only():
    _lib.check(lib.amx_conv3d_k3_reflect(_lib.ptr(x0), c0, None, 0, _lib.ptr(w[:, :c0].contiguous()), None, _lib.ptr(sh), cout, n, S, S, S, 1, 0.3, 0,
                                         _lib.ptr(wpk), _lib.ptr(out), None, st))
def t(f, K=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
fl = 2 * 27 * (c0 + c1) * cout * n * S ** 3
a, b, c = t(plain), t(merged), t(skip_only)
print(f"{c0}+{c1}->{cout} @{S} n={n}: plain {a:.1f} us ({fl/a/1e6:.0f} TF)  merged {b:.1f} us ({fl/b/1e6:.0f} TF alg)  skip conv alone {c:.1f} us -> merged-tap launch ~{b-c:.1f} us (incl. 2 pack launches)")
