cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -p no:cacheprovider -k "conv_fwd_bwd or conv_full" 2>&1 | tail -3
UZ_CONV_MATH=f32 python -m pytest tests/test_ops_gpu.py -x -q -p no:cacheprovider -k "conv_fwd_bwd or conv_full" 2>&1 | tail -3
for t in 0 128 512; do echo "#### small-tile threshold $t"
for s in "192 192 2 2" "192 192 4 4" "256 256 4 4" "192 192 8 8" "256 256 8 8"; do echo -n "$s : "; UZ_WG_SMALLTILE=$t python tools/bench_conv.py $s 32 3 20 2>/dev/null | tr '\n' '|'; echo; done; done
echo "#### f32 mode big layers"
for s in "224 128 128 128" "128 128 128 128" "192 192 64 64" "64 64 64 64" "32 32 128 128" "128 128 32 32"; do echo -n "$s : "; UZ_CONV_MATH=f32 python tools/bench_conv.py $s 32 3 5 2>/dev/null | tr '\n' '|'; echo; done
UZ_CONV_MATH=f32 python bench.py --steps 10 --warmup 3 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
for r in 1 2; do python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140; UZ_WG_SMALLTILE=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140; done
