cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_split_hardening_gpu.py -q -x -k "fused_bn or hardening or outlier or zero or bound or unnormalised" -p no:cacheprovider -s 2>&1 | tail -15
UZ_BN_FUSE_STATS=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175
UZ_BN_FUSE_STATS=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175
python -m pytest tests -m gpu -q -x --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
