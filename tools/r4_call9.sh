cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_split_storage_gpu.py -x -q -p no:cacheprovider -k "bn or split_storage or unit" 2>&1 | tail -4
python -m pytest tests/test_phiseg_gpu.py -x -q -p no:cacheprovider -s 2>&1 | grep -E "^step|passed|failed|Error" | head
for r in 1 2; do
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
UZ_BN_MID=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
done
python tools/op_profile.py 32 phiseg 2>&1 | grep -E "BN_RELU_BWD|^total|bn_relu" | head -20
