cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "large_offset or fused_bn" 2>&1 | tail -25
