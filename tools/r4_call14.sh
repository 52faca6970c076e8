cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_split_storage_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
for r in 1 2 3; do
timeout 300 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
UZ_WGRAD_TABLE=0 timeout 300 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1800 2>&1 | tail -6
