cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_cpu_twins.py tests/test_unet_probunet_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15
