cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_split_storage_gpu.py tests/test_split_hardening_gpu.py tests/test_phiseg_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
for r in 1 2 3; do
timeout 300 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
UZ_BN_MID_FWD=0 timeout 300 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
done
