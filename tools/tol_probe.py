"""Scratch: the numbers behind the tolerance bounds of tests/test_unet_gpu.py (f16 storage, 6 M UNet and the small / wide nets)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import anatomix_amd
from tests.test_unet_gpu import _model, R, KW, rel_l2, max_rel
dev = torch.device("cuda:0")
for seed in (0, 1, 2, 3):
    for size in ((64, 64, 64),):
        m, sd = _model(dev, seed, 1.0)
        x = R.synthetic_input(100 + seed, 1, size)
        with torch.no_grad():
            y = m(x.to(dev)).cpu(); ref = R.forward(x, sd, KW)
        print(seed, size, "rel_l2 %.3e max_rel %.3e" % (rel_l2(y, ref), max_rel(y, ref)), flush=True)
for kw, size in [(dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16), (8, 12, 16)),
                 (dict(dimension=3, input_nc=1, output_nc=64, num_downs=1, ngf=16), (32, 32, 32)),
                 (dict(dimension=3, input_nc=1, output_nc=16, num_downs=3, ngf=16, doubleconv=False), (16, 16, 24))]:
    m = anatomix_amd.Unet(**kw); sd = R.synthetic_state_dict(kw, 4); m.load_state_dict(sd); m = m.to(dev).eval()
    x = R.synthetic_input(21, 2, size)
    with torch.no_grad():
        y = m(x.to(dev)).cpu(); ref = R.forward(x, sd, kw)
    print(kw["output_nc"], kw["num_downs"], size, "rel_l2 %.3e max_rel %.3e" % (rel_l2(y, ref), max_rel(y, ref)), flush=True)
for gain in (1.0, 2 ** 0.5):
    for size, n in [((32, 32, 32), 1), ((64, 64, 64), 1)]:
        m, sd = _model(dev, 0, gain)
        x = R.synthetic_input(100, n, size)
        with torch.no_grad():
            y = m(x.to(dev)).cpu(); ref = R.forward_lowp(x, sd, KW, torch.float16); ref32 = R.forward(x, sd, KW)
        print("gain %.2f" % gain, size, "vs emulation rel_l2 %.3e | vs fp32 rel_l2 %.3e max_rel %.3e" % (rel_l2(y, ref), rel_l2(y, ref32), max_rel(y, ref32)), flush=True)
