cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_split_storage_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
for r in 1 2 3; do
timeout 300 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
UZ_BN_CLUSTER=0 timeout 300 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
done
timeout 600 python tools/op_profile.py 32 phiseg 2>&1 | grep -E "BN_RELU_BWD|^total|bn_relu" | head -12
timeout 900 python -m pytest tests/test_phiseg_gpu.py tests/test_full_configs_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
