cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_full_configs_gpu.py -q -x -k "conv or split" -p no:cacheprovider 2>&1 | tail -3
for v in 1 0; do echo "#### UZ_SPLIT_KSPLIT=$v (1 = off)"; export UZ_SPLIT_KSPLIT=$v
for l in "192 192 16 16" "256 256 16 16" "128 128 16 16" "64 64 32 32"; do echo "== $l"; python tools/bench_conv.py $l 2>/dev/null | grep -v wgrad; done
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c90-160
done
