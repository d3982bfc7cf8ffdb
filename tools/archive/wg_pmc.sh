#!/bin/bash
# SQ counters of the weight-gradient kernel on the step's shapes (GPU box).  usage: bash tools/wg_pmc.sh
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wgp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
  --output-format csv -d /tmp/wgp -o p -- python $REPO/tools/wgrad_time.py > $OUT/wg_pmc.log 2>&1
f=$(ls /tmp/wgp/*counter_collection.csv | head -1)
python - "$f" <<'PY' | tee $OUT/wg_pmc.txt
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    if "wgrad_kernel" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"][:60] + " grid " + r.get("Grid_Size", "?")
    a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    d = {c: v[1] / v[0] for c, v in cs.items()}
    print(k)
    print("   ", {c: f"{v:.4g}" for c, v in d.items()}, "lds_conflict_frac %.3f" % (d.get("SQ_LDS_BANK_CONFLICT", 0) / (d.get("SQ_LDS_IDX_ACTIVE", 1) or 1)))
PY
