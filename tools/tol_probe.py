import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from tests.test_unet_gpu import _model, R, KW, rel_l2, max_rel
dev = torch.device("cuda:0")
for seed in (0, 1, 2, 3):
    for size in ((64, 64, 64),):
        m, sd = _model(dev, seed, 1.0)
        x = R.synthetic_input(100 + seed, 1, size)
        with torch.no_grad():
            y = m(x.to(dev)).cpu(); ref = R.forward(x, sd, KW)
        print(seed, size, "rel_l2 %.3e max_rel %.3e" % (rel_l2(y, ref), max_rel(y, ref)))
