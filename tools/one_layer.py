"""Scratch: time ONE conv layer shape through amx_conv3d_k3_reflect (hipEvents), e.g.
   python tools/one_layer.py 16 0 16 128 [n]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd import _lib
c0, c1, cout, S = [int(a) for a in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 1
planar = len(sys.argv) > 6 and sys.argv[6] == "planar"
dev = torch.device("cuda:0"); lib = _lib.load()
x0 = torch.randn(n, S, S, S, c0, device=dev).half()
x1 = torch.randn(n, S // 2, S // 2, S // 2, c1, device=dev).half() if c1 else None
w = (torch.randn(cout, c0 + c1, 27, device=dev) / (27 * (c0 + c1)) ** 0.5).float()
sh = torch.zeros(cout, device=dev)
wpk = torch.empty(lib.amx_conv3d_packed_bytes(c0 + c1, cout), dtype=torch.uint8, device=dev)
o16 = None if planar else torch.empty(n, S, S, S, cout, device=dev, dtype=torch.half)
o32 = torch.empty(n, cout, S, S, S, device=dev) if planar else None
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
sb = lib.amx_conv3d_scratch_bytes(c0, c1, cout, n, S, S, S, 0)
scratch = torch.empty(max(sb, 4) // 4, device=dev) if sb else None
def run():
    _lib.check(lib.amx_conv3d_k3_reflect_ws(_lib.ptr(x0), c0, _lib.ptr(x1), c1, _lib.ptr(w), None, _lib.ptr(sh), cout, n, S, S, S,
                                            1, 0.3, 0, _lib.ptr(wpk), _lib.ptr(o16), _lib.ptr(o32), _lib.ptr(scratch), sb, st))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
e0.record()
for _ in range(K): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / K * 1e3     # includes the tiny pack kernel (~3 us)
fl = 2 * 27 * (c0 + c1) * cout * n * S ** 3
print(f"AMX_DBG={os.environ.get('AMX_DBG','0')} {c0}+{c1}->{cout} @{S} n={n}: {us:.1f} us  {fl/us/1e6:.0f} TF")
