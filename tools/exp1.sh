# round-3 experiment 1: phase stamps of the split kernels, dual-workgroup variant, per-op table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for l in "224 128 128 128" "128 128 128 128" "256 192 64 64" "32 32 128 128" "64 64 64 64"; do
  echo "== stamps $l"; python tools/stamp_conv.py $l 2>&1 | tail -3
  echo "== stamps dual $l"; UZ_SPLIT_EXP=1 python tools/stamp_conv.py $l 2>&1 | tail -3
done
UZ_SPLIT_EXP=1 python -m pytest tests/test_ops_gpu.py tests/test_full_configs_gpu.py -q -x -k "conv or split" -p no:cacheprovider 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175
UZ_SPLIT_EXP=1 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175
python tools/op_profile.py 32 phiseg > gpurun_out/op_profile_r3_base.txt 2>&1; head -60 gpurun_out/op_profile_r3_base.txt
