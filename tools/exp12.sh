cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_full_configs_gpu.py tests/test_cpu_twins.py tests/test_split_hardening_gpu.py -q -x -k "conv or split or twin" -p no:cacheprovider 2>&1 | tail -3
for l in "224 128 128 128" "128 128 128 128" "256 192 64 64" "192 192 64 64" "64 64 64 64" "128 128 32 32" "192 192 32 32" "256 256 16 16" "192 192 16 16"; do
  echo "== $l"; python tools/bench_conv.py $l 3 5 wgrad 2>&1 | tail -1
done
python tools/stamp_conv.py 224 128 128 128 2>&1 | tail -1
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c90-175
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c90-175
python bench.py --model unet --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c85-175
