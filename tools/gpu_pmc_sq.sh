#!/bin/bash
# SQ-side PMC pass of the headline bench (GPU box): wave-cycle breakdown, MFMA busy cycles, LDS conflicts per kernel.
# Counters only (no trace domains).  usage: tools/gpu_pmc_sq.sh <tag> [bench args, e.g. --variant anatomix-dev-vit]  ->  gpurun_out/<tag>_pmc_sq.json
TAG=${1:-rXX}; shift
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-parity "$@" > $OUT/${TAG}_pmc_sq.log 2>&1
cd $REPO
f=$(ls $OUT/${TAG}_pmc_sq/*counter_collection.csv | head -1)
python - "$f" "$OUT/${TAG}_pmc_sq.json" <<'PY'
import csv, collections, json, sys
sys.path.insert(0, "tools")
from pmc_summary import friendly
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = friendly(r["Kernel_Name"])
    if not k: continue
    a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {"_meta": {"unit": "counter value per launch, batch 4", "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; "
                 "SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (MI355X_MICROARCH.md)"}}
for k, cs in sorted(agg.items()):
    d = {c: v[1] / v[0] for c, v in cs.items()}
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1.0
    d["frac_wait_any"] = d.get("SQ_WAIT_ANY", 0) / wc
    d["frac_wait_inst_any"] = d.get("SQ_WAIT_INST_ANY", 0) / wc
    d["frac_active_inst"] = d.get("SQ_ACTIVE_INST_ANY", 0) / wc
    d["frac_wait_inst_lds"] = d.get("SQ_WAIT_INST_LDS", 0) / wc
    d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / (d.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0)
    out[k] = d
    print(f"{k:58s} wait_any {d['frac_wait_any']:.2f} wait_inst {d['frac_wait_inst_any']:.2f} active {d['frac_active_inst']:.2f} "
          f"wait_lds {d['frac_wait_inst_lds']:.2f} mfma_busy {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} lds_conflict {d['lds_conflict_frac']:.3f}")
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
rm -rf $OUT/${TAG}_pmc_sq
