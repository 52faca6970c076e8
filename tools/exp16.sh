cd $GRAFT_REPO_ROOT
UZ_SPLIT_TINY=1 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_fwd_bwd" -p no:cacheprovider 2>&1 | tail -3
for t in 0 1; do for l in "192 192 8 8" "192 192 4 4" "192 192 2 2" "256 256 8 8"; do echo "tiny=$t $l: $(UZ_SPLIT_TINY=$t python tools/bench_conv.py $l 32 3 5 fwd 2>&1 | tail -1)"; done; done
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for t in 0 1 0 1; do echo "== tiny $t"; UZ_SPLIT_TINY=$t $B 2>/dev/null | cut -c90-160; done
UZ_SPLIT_TINY=1 python -m pytest tests/test_phiseg_gpu.py tests/test_unet_probunet_gpu.py tests/test_full_configs_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
