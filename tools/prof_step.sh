#!/bin/bash
# rocprofv3 kernel stats of the graph-replayed contrastive step (GPU box).  usage: tools/prof_step.sh <tag>
TAG=${1:-t}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python $REPO/bench.py --no-cpu-baseline --no-secondary --no-parity --sustain 0 --workload step --steps 6 --warmup 2 > $OUT/${TAG}_prof.log 2>&1
f=$(ls $OUT/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -70 $f > $OUT/${TAG}_step_kernel_stats.csv
rm -rf $OUT/${TAG}_prof
tail -1 $OUT/${TAG}_prof.log | cut -c1-200
python3 - $OUT/${TAG}_step_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("listed kernels total ms:", tot/1e6, "calls:", sum(int(r["Calls"]) for r in rows))
for r in rows[:40]:
    print(f'{r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e3:10.1f} us  avg {float(r["AverageNs"])/1e3:8.1f}')
PY
