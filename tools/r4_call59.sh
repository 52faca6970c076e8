cd $GRAFT_REPO_ROOT
python -m pytest tests/test_b16_storage_gpu.py tests/test_ops_gpu.py tests/test_split_storage_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do python bench.py --model phiseg3d --steps 20 --warmup 5 --skip-cpu --no-profile 2>/dev/null | tail -1 | cut -c1-200; done
python -m pytest tests/test_phiseg3d.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | cut -c1-160
