#!/bin/bash
# SQ counters of the weight-gradient kernel per layer shape of tools/wgrad_layers.py (counters only, no trace domains).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wgpmc
WG_REPS=1 WG_NOCHECK=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS \
  --output-format csv -d /tmp/wgpmc -o p -- python $R/tools/wgrad_layers.py > /tmp/wgpmc.log 2>&1
f=$(find /tmp/wgpmc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad_tr" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "")})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for d, v in by.items():
    key = (v["name"], v["grid"], round(v.get("SQ_INSTS_LDS", 0)))
    if key in seen: continue
    seen.add(key)
    cfg = v["name"].split("WtCfg")[1][:22]
    idx = v.get("SQ_LDS_IDX_ACTIVE", 0) or 1
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{cfg:24s} grid {v['grid']:>7s} lds_insts {v.get('SQ_INSTS_LDS',0):.3g} idx_active {idx:.3g} conflict {v.get('SQ_LDS_BANK_CONFLICT',0):.3g} "
          f"({v.get('SQ_LDS_BANK_CONFLICT',0)/idx:.2f}) mfma_busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} wait_lds {v.get('SQ_WAIT_INST_LDS',0)/wc:.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}")
PY
tail -3 /tmp/wgpmc.log
