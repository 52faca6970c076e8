cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/op_profile.py 32 unet > gpurun_out/op_profile_unet.txt 2>&1; head -40 gpurun_out/op_profile_unet.txt; tail -12 gpurun_out/op_profile_unet.txt
python -m pytest tests/test_dp_gpu.py tests/test_phiseg3d.py -q -x -k "bench or config5" -p no:cacheprovider 2>&1 | tail -5
