cd $GRAFT_REPO_ROOT
export UZ_CONV_MATH=f32
echo -n "224->128 dgrad/fwd: "; python tools/bench_conv.py 224 128 128 128 32 3 5 2>/dev/null | grep -E "fwd|dgrad" | tr '\n' ' '; echo
B="python bench.py --steps 15 --warmup 3 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do $B 2>/dev/null | tail -1 | cut -c60-100; done
python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv_fwd_bwd" 2>&1 | tail -1
