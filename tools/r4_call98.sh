# fused latent heads: op test, model A/B test, PHiSeg GPU tests, A/B bench (3 alternations)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "latent" -p no:cacheprovider 2>&1 | tail -15
python -m pytest tests/test_phiseg_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
for r in 1 2 3; do
  for f in 1 0; do
    UZ_FUSE_HEADS=$f python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f', d['value'], d['ms_per_step'])"
  done
done
