#!/bin/bash
# ordered kernel trace of ONE headline forward (GPU box): name, start offset, duration -> stdout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ft
rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -o t -- python $R/bench.py --no-cpu-baseline --no-secondary --no-parity --sustain 0 --steps 6 --warmup 2 > /dev/null 2>&1
python - $(find /tmp/ft -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last forward: from the last stem-fed launch backwards to the previous one
idx = [i for i, r in enumerate(rows) if "StemIn" in r["Kernel_Name"] and "Lb1EEE" in r["Kernel_Name"].replace(" ", "")] or \
      [i for i, r in enumerate(rows) if "true>(amx::ConvParams, int, int, amx::StemIn)" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:100]}")
    prev_end = e
print("forward period", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")
PY
