# A/B of conv kernel variants in ONE box call: correctness subset, per-layer times, step time; env var toggles given as arguments.
# usage: bash tools/ab_conv.sh "UZ_CONV_DB=0" "UZ_CONV_DB=1"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_full_configs_gpu.py tests/test_split_storage_gpu.py -q -x -k "conv or split" -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
for v in "$@"; do
  echo "=== $v (round $rep)"
  for l in "224 128 128 128" "128 128 128 128" "256 192 64 64" "192 192 64 64"; do echo "-- $l"; env $v python tools/bench_conv_packed.py $l 2>/dev/null | tail -1; done
  env $v python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-120
done
done
