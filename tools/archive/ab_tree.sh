#!/bin/bash
# A/B of the headline between this tree and ANOTHER REVISION on one box (boxes differ by +-3 %, more than most changes):
#   rm -rf ab_old && mkdir ab_old && git archive <rev> anatomix_amd include oracle bench.py __graft_entry__.py BASELINE.json | tar -x -C ab_old
#   make -C ab_old/anatomix_amd/csrc        (ab_old/ is git-ignored but travels with gpurun)
#   gpurun -- 'bash tools/ab_tree.sh'
# Round 3, end of round vs the revision before the ViT engine (42b9fcf): new 2739 / 2753 / 2759, old 2744 / 2744 / 2754 volumes/s.
cd $GRAFT_REPO_ROOT
for which in new old new old new old; do
  if [ $which = old ]; then d=ab_old; else d=.; fi
  (cd $d && python bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which headline', d['value'], d['ms_per_step'])")
done
