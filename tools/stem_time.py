import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0"); kw = R.VARIANTS["anatomix"]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.to(dev).eval()
x = R.synthetic_input(100, 4, (128,)*3).to(dev)
with torch.no_grad():
    for _ in range(2): m(x)
    tot = {}
    for _ in range(5):
        _, recs = m.profile_forward(x)
        for r in recs[:1]: tot[r["kernel"]] = tot.get(r["kernel"], 0) + r["ms"]
print("AMX_DBG", os.environ.get("AMX_DBG", "0"), {k: round(v / 5 * 1e3, 1) for k, v in tot.items()})
