import sys; sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
from _util import run_conv, ref_conv, rel_l2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (cin, cout, size) in [(16,16,(36,36,36)), (16,16,(68,68,68)), (32,32,(36,36,36)), (32,16,(36,36,36)), (64,64,(20,20,20)), (128,128,(12,12,12)), (256,128,(12,12,12)),
                          (64,32,(36,36,36)), (16,16,(132,20,36)), (256,256,(6,6,6)), (48,16,(36,36,36)), (16,32,(36,36,36))]:
    x = torch.randn(1, cin, *size, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    try:
        y = run_conv(dev, x, None, w, None, None, 0, "f16")
        r = ref_conv(x, None, w, None, None, 0, "f16")
        print(cin, cout, size, "rel_l2 %.2e" % rel_l2(y, r), "nan" if torch.isnan(y).any() else "")
    except Exception as e:
        print(cin, cout, size, "ERR", str(e)[:100])
