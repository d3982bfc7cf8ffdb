"""Experiment: the ViT forward of a batch as one launch sequence vs split over two HIP streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anatomix_amd.model.load_from_hf import build_variant
from oracle import vit_ref as V
dev = torch.device("cuda:0")
m = build_variant("anatomix-dev-vit"); m.load_state_dict(V.synthetic_state_dict(V.VIT_VARIANTS["anatomix-dev-vit"], 0)); m = m.to(dev).eval()
streams = [torch.cuda.Stream(dev) for _ in range(2)]
def run(x, split):
    if not split:
        return m(x)
    cur = torch.cuda.current_stream(dev)
    ev = cur.record_event()
    outs = []
    h = x.shape[0] // 2
    for s, part in zip(streams, (x[:h], x[h:])):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            outs.append(m(part))
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)
with torch.no_grad():
    for B in (2, 4):
        x = V.synthetic_input(1, B).to(dev)
        for split in (False, True):
            for _ in range(3): run(x, split)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = 10
            for _ in range(K): run(x, split)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"ViT batch {B} {'two streams' if split else 'one stream'}: {dt/K*1e3:.2f} ms, {B*K/dt:.1f} vol/s")
