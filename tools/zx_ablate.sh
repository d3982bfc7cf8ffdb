#!/bin/bash
# ablations of the fused 32 -> 32 kernel (AMX_DBG: 2 no MFMA sweeps, 4 no stores, 16 converters idle after the first ring fill);
# ablated runs produce garbage, so the overflow guard's error is swallowed and the kernel time comes from rocprofv3
cd /tmp; export TMPDIR=/tmp
cat > /tmp/zx_fwd.py <<'PY'
import torch, sys
sys.path.insert(0, "/root/repo")
from anatomix_amd.model.load_from_hf import build_variant
torch.manual_seed(0)
m = build_variant("anatomix-dev").cuda().eval()
x = torch.rand(4, 1, 128, 128, 128, device="cuda")
for i in range(6):
    try:
        with torch.no_grad(): m(x)
    except Exception as e:
        pass
torch.cuda.synchronize()
PY
for dbg in ${DBGS:-0 2 4 16 18 22}; do
  rm -rf /tmp/zxp
  AMX_DBG=$dbg rocprofv3 --kernel-trace --stats -d /tmp/zxp -o r -- python /tmp/zx_fwd.py > /tmp/zx.log 2>&1
  f=$(find /tmp/zxp -name "*kernel_stats.csv" | head -1)
  echo "dbg=$dbg $(grep -E 'conv3d_k3_zx' $f | cut -d, -f1-5 | cut -c1-30,60-)"
done
