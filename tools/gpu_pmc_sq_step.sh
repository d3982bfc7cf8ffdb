#!/bin/bash
# SQ-side PMC pass of the training step (eager, no graph): wave-cycle breakdown per kernel family.  Counters only.
# usage: tools/gpu_pmc_sq_step.sh <tag>  ->  gpurun_out/<tag>_pmc_sq_step.json
TAG=${1:-rXX}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/${TAG}_pmc_sq_step -o p -- python $REPO/bench.py --workload step --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/${TAG}_pmc_sq_step.log 2>&1
cd $REPO
f=$(ls $OUT/${TAG}_pmc_sq_step/*counter_collection.csv | head -1)
python - "$f" "$OUT/${TAG}_pmc_sq_step.json" <<'PY'
import csv, collections, json, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "amx" not in n: continue
    m = re.search(r"amx[:0-9]*([a-z0-9_]+_kernel)", n)
    k = m.group(1) if m else n[:40]
    if "wgrad_kernel" in n:
        k += "/grid" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, cs in sorted(agg.items()):
    d = {c: v[1] / v[0] for c, v in cs.items()}
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1.0
    d["frac_wait_any"] = d.get("SQ_WAIT_ANY", 0) / wc
    d["frac_wait_inst_any"] = d.get("SQ_WAIT_INST_ANY", 0) / wc
    d["frac_active_inst"] = d.get("SQ_ACTIVE_INST_ANY", 0) / wc
    d["frac_wait_inst_lds"] = d.get("SQ_WAIT_INST_LDS", 0) / wc
    d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / (d.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0)
    out[k] = d
    print(f"{k:44s} wave_cyc {wc:.3g} wait_any {d['frac_wait_any']:.2f} wait_inst {d['frac_wait_inst_any']:.2f} active {d['frac_active_inst']:.2f} "
          f"wait_lds {d['frac_wait_inst_lds']:.2f} mfma_busy {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} lds_conflict {d['lds_conflict_frac']:.3f}")
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
head -2 $f | cut -c1-400
rm -rf $OUT/${TAG}_pmc_sq_step
