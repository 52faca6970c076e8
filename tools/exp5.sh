cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for m in phiseg unet probunet phiseg3d; do python bench.py --model $m --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175; done
