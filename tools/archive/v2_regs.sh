#!/bin/bash
# register / scratch usage of the conv3d_k3_v2 instantiations (arg: regex on the last two template bools, default MX = "Lb1ELb1E")
cd /root/repo/anatomix_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c amx_conv3d_v2.hip -o amx_conv3d_v2.o -Rpass-analysis=kernel-resource-usage 2>/tmp/v2_res.txt
python3 - "$@" <<'PY'
import re, sys
txt = open('/tmp/v2_res.txt').read()
if ' error' in txt: print(txt[:4000])
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    m = re.search(r'conv3d_k3_v2_kernelI(\w+?)_?Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)', name)
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
    if m and (m.group(13) == '1' or len(sys.argv) > 1):
        t, wz, wy, wx, nwz, nwy, q, nch, om, nlw, nbuf, sp, mx = m.groups()
        print("%-5s brick %dx%dx%-2d q%s nch%s o%s lw%s b%s split%s mx%s : VGPR %3d AGPR %3d scratch %4d occ %d" % (t, int(wz)*int(nwz), int(wy)*int(nwy), int(wx), q, nch, om, nlw, nbuf, sp, mx, g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')))
PY
