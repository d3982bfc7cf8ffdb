#!/bin/bash
# issue-side SQ counters of the weight-gradient kernel (two passes of 8 counters), layers WG_ONLY (default 0,3,9)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export WG_REPS=1 WG_NOCHECK=1 WG_ONLY=${WG_ONLY:-0,3,9}
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_WAVES"; do
  rm -rf /tmp/wgpmc
  rocprofv3 --pmc $pass --output-format csv -d /tmp/wgpmc -o p -- python $R/tools/wgrad_layers.py > /tmp/wgpmc.log 2>&1
  f=$(find /tmp/wgpmc -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad_tr" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
last = {}
for d, v in by.items(): last[v["name"].split("WtCfg")[1][:16] + str(round(v.get("SQ_INSTS_LDS", v.get("SQ_ACTIVE_INST_LDS", 0)) / 1e5))] = v
for k, v in last.items():
    print(k, " ".join(f"{c[3:]}={x:.4g}" for c, x in v.items() if c != "name"))
PY
done
tail -2 /tmp/wgpmc.log
