cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -x -k "normal_stream or fused_bn" -p no:cacheprovider 2>&1 | tail -3
for d in 0 8192 32768 0 8192 32768; do echo "== decouple $d"; UZ_DECOUPLE_WGRAD=$d python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c90-175; done
echo "== 3 lanes"; UZ_LANES=3 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c90-175
echo "== probunet"; for d in 0 8192; do UZ_DECOUPLE_WGRAD=$d python bench.py --model probunet --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c90-185; done
python -m pytest tests -m gpu -q -x --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
