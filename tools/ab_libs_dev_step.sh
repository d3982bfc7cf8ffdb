cd $GRAFT_REPO_ROOT
L=anatomix_amd/csrc/libanatomix_amd.so
cp $L /tmp/new.so
for which in new old new old; do
  if [ $which = old ]; then cp anatomix_amd/csrc/lib_old.bin $L; else cp /tmp/new.so $L; fi
  python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --variant anatomix-dev --batch 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which dev', d['value'], d['roofline']['avg_launch_us'])"
  python bench.py --workload step --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which step', d['value'], d['ms_per_step'])"
done
cp /tmp/new.so $L
timeout 900 python -m pytest tests/test_unet_dev_gpu.py tests/test_train_ops_gpu.py tests/test_train_step_gpu.py tests/test_pretrain_gpu.py tests/test_range_safety_gpu.py tests/test_unet_taps_gpu.py -x -q -m gpu 2>&1 | tail -2
