import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import anatomix_amd
from oracle import unet_ref as R
from _util import rel_l2
dev = torch.device("cuda:0")
kw = dict(dimension=3, input_nc=1, output_nc=16, num_downs=1, ngf=16)
def run(prec, scale, layers, use_out=True):
    hip = anatomix_amd.Unet(**kw); hip.load_state_dict(R.synthetic_state_dict(kw, 3)); hip.precision = prec
    ref = copy.deepcopy(hip); ref.allow_torch_path = True; ref._warned = True
    hip, ref = hip.to(dev).train(), ref.to(dev).train()
    x = R.synthetic_input(11, 2, (32, 32, 64)).to(dev)
    oh, fh = hip(x, layers) if layers else (hip(x), [])
    orr, fr = ref._forward_torch(x, layers, False, False) if layers else (ref._forward_torch(x, [], False, False), [])
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(f.shape, generator=g).to(dev) / f[0].numel() ** 0.5 for f in fr]
    lh = sum((f * c).sum() for f, c in zip(fh, cots)) + (0.1 * oh.square().mean() if use_out else 0)
    lr = sum((f * c).sum() for f, c in zip(fr, cots)) + (0.1 * orr.square().mean() if use_out else 0)
    (lh * scale).backward(); lr.backward()
    print(f"--- {prec} scale {scale} layers {layers} use_out {use_out}")
    for (name, ph), (_, pr) in zip(hip.named_parameters(), ref.named_parameters()):
        if ph.grad is None: print("   ", name, "NO GRAD", pr.grad.norm().item()); continue
        gh = ph.grad / scale
        print("   %-18s rel %.3e  |ref| %.3e |hip| %.3e" % (name, rel_l2(gh.cpu(), pr.grad.cpu()), pr.grad.norm().item(), gh.norm().item()))
print([f"{i}:{type(m).__name__}" for i, m in enumerate(anatomix_amd.Unet(**kw).model)])
run("f16", 1.0, [], True)
run("f16", 1024.0, [], True)
run("bf16", 1.0, [], True)
run("f16", 1024.0, [20], False)
run("f16", 1024.0, [3], False)
