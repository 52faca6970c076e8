cd $GRAFT_REPO_ROOT
python -m pytest tests/test_split_storage_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -30
python -m pytest tests/test_ops_gpu.py tests/test_phiseg_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -15
for r in 1 2; do python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-175; done
UZ_PACK_ACT=0 UZ_PACK_DY=0 UZ_FOLD_BN_BWD=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-175
UZ_PACK_ACT=0 UZ_PACK_DY=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-175
UZ_FOLD_BN_BWD=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-175
