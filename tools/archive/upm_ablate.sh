#!/bin/bash
# per-kernel medians of the merged concat conv under ablation switches (rocprofv3 kernel trace)
for dbg in 0 16 4 20 2; do
  AMX_DBG=$dbg tools/prof_cmd.sh r03d_$dbg $PWD/tools/upmerge_time.py ${1:-32} ${2:-64} ${3:-64} > /dev/null
  python - <<PY
import csv, statistics as st
rows=list(csv.DictReader(open("gpurun_out/r03d_${dbg}_prof/r03d_${dbg}_kernel_trace.csv")))
v=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "upmerge_kernel" in r["Kernel_Name"]]
print("AMX_DBG=$dbg upmerge median %.1f us (n=%d)" % (st.median(v), len(v)))
PY
done
