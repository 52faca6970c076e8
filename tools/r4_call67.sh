cd $GRAFT_REPO_ROOT
python tools/bench_resample.py 2>&1 | grep bilinear
python -m pytest tests/test_ops_gpu.py tests/test_phiseg3d.py tests/test_b16_storage_gpu.py -m gpu -q -p no:cacheprovider -k "bilinear or interpolation or trilinear or pool" 2>&1 | tail -2
