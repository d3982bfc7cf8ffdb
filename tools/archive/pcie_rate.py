"""PCIe-inclusive rate of the 128^3 forward: host volumes in (as extract_features uploads them), features left on the GPU
(as the reference keeps them), and the variant that also brings the features back to the host."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import anatomix_amd
from oracle import unet_ref as R

dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
with contextlib.redirect_stdout(io.StringIO()):
    m = anatomix_amd.Unet(**kw)
m.load_state_dict(R.synthetic_state_dict(kw, 0))
m = m.to(dev).eval()
B = 4
host = torch.rand(B, 1, 128, 128, 128)
pinned = host.pin_memory()
out_host = torch.empty(B, 16, 128, 128, 128).pin_memory()


def run(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    xd = host.to(dev)
    t_res = run(lambda: m(xd))
    t_page = run(lambda: m(host.to(dev)))
    t_pin = run(lambda: m(pinned.to(dev, non_blocking=True)))
    t_back = run(lambda: out_host.copy_(m(pinned.to(dev, non_blocking=True)), non_blocking=True), 5)
print(f"resident input            {B / t_res:8.1f} volumes/s  ({t_res * 1e3:.2f} ms per batch of {B})")
print(f"pageable host input       {B / t_page:8.1f} volumes/s  ({t_page * 1e3:.2f} ms)")
print(f"pinned host input         {B / t_pin:8.1f} volumes/s  ({t_pin * 1e3:.2f} ms)")
print(f"pinned in + features out  {B / t_back:8.1f} volumes/s  ({t_back * 1e3:.2f} ms)")
