import sys; sys.path.insert(0, "/root/repo")
import torch, anatomix_amd
from oracle import unet_ref as R
kw = R.VARIANTS["anatomix"]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.cuda().eval()
x = R.synthetic_input(100, 1, (32, 32, 32)).cuda()
with torch.no_grad():
    y0 = m(x)
    cm = torch.compile(m)
    try:
        y1 = cm(x)
        print("compiled forward OK, equal:", torch.equal(y0, y1))
    except Exception as e:
        print("compile FAILED:", type(e).__name__, str(e)[:400])
seq = torch.nn.Sequential(m, torch.nn.Conv3d(16, 3, 1).cuda())
with torch.no_grad():
    a = seq(x); b = torch.compile(seq)(x)
print("compiled Sequential(Unet, head) OK, max diff:", (a - b).abs().max().item())
