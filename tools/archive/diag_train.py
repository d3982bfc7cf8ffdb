import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import anatomix_amd
from oracle import unet_ref as R, pretrain_inputs as PI
from _util import rel_l2
dev = torch.device("cuda:0")
KW = R.VARIANTS["anatomix"]
for prec in ("f16", "bf16"):
    hip = anatomix_amd.Unet(**KW); hip.load_state_dict(R.synthetic_state_dict(KW, 3, gain=2 ** 0.5)); hip.precision = prec
    ref = copy.deepcopy(hip); ref.allow_torch_path = True; ref._warned = True
    hip, ref = hip.to(dev).train(), ref.to(dev).train()
    A, B, _ = PI.step_inputs(64); x = torch.cat((A, B)).to(dev)
    convs = [i for i, m in enumerate(hip.model) if isinstance(m, torch.nn.Conv3d)]
    layers = convs + [2, 5]
    with torch.no_grad():
        oh, fh = hip(x, layers)
        orr, fr = ref._forward_torch(x, layers, False, False)
    print(prec, "out", rel_l2(oh.cpu(), orr.cpu()))
    for l, a, b in zip(sorted(layers), fh, fr):
        print("   tap", l, tuple(a.shape), "%.3e" % rel_l2(a.cpu(), b.cpu()))
