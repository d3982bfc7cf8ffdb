import contextlib, io, os, sys
sys.path.insert(0, os.getcwd())
import torch
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
with contextlib.redirect_stdout(io.StringIO()):
    m = anatomix_amd.Unet(**kw)
m.load_state_dict(R.synthetic_state_dict(kw, 0))
m = m.to(dev).eval()
x = torch.rand(4, 1, 128, 128, 128, device=dev)
with torch.no_grad():
    y_ref = m(x).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            m(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    x.copy_(torch.rand_like(x))
    g.replay()
    torch.cuda.synchronize()
    y2 = m(x)
    print("graph replay equals eager:", torch.equal(y, y2), "differs from first input:", not torch.equal(y, y_ref))
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(50): m(x)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"replay {1e3*(t1-t0)/50:.3f} ms, eager {1e3*(t2-t1)/50:.3f} ms per forward (4 x 128^3)")
