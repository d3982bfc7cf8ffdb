#!/bin/bash
# Kernel trace of tools/wgrad_layers.py: per-launch durations of the weight-gradient kernel and its reduce, in launch order.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/wgp && mkdir -p /tmp/wgp
WG_REPS=${WG_REPS:-3} rocprofv3 --kernel-trace --output-format csv -d /tmp/wgp -o wg -- python $R/tools/wgrad_layers.py > /tmp/wgp/log.txt 2>&1
f=$(find /tmp/wgp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", "")) for r in rows if "wgrad" in r["Kernel_Name"]]
# group consecutive identical (name, grid) launches
out = []
for name, us, grid, lds in seq:
    short = name.split("(")[0][-70:]
    key = (short, grid)
    if out and out[-1][0] == key: out[-1][1].append(us)
    else: out.append([key, [us]])
for key, v in out:
    v2 = sorted(v)
    print(f"{key[0]:72s} grid {key[1]:>8s}  n={len(v):3d}  median {v2[len(v2)//2]:8.1f} us  min {v2[0]:8.1f}")
PY
