cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_full_configs_gpu.py -q -x -k "conv or split" -p no:cacheprovider 2>&1 | tail -3
for l in "$@"; do echo "== $l"; python tools/bench_conv.py $l 2>/dev/null; done
for m in phiseg unet probunet; do python bench.py --model $m --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175; done
