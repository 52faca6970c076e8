# the GPU tier in the three math modes (default, fp32-MFMA only, split forced everywhere)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_default.log 2>&1; echo "pytest default rc=$?"; tail -3 gpurun_out/pytest_gpu_default.log
UZ_CONV_MATH=f32 python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_f32.log 2>&1; echo "pytest f32 rc=$?"; tail -3 gpurun_out/pytest_gpu_f32.log
UZ_CONV_MATH=split python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_split.log 2>&1; echo "pytest split rc=$?"; tail -4 gpurun_out/pytest_gpu_split.log
