cd $GRAFT_REPO_ROOT
export UZ_CONV_MATH=bf16
for shape in "288 96 128 64 128" "96 288 128 64 128" "192 64 128 64 128" "576 192 64 32 64" "192 192 64 32 64"; do
  echo "-- $shape"; python tools/bench_conv.py $shape 3 8 2>/dev/null | grep -E "fwd|dgrad" | tr '\n' ' '; echo
done
python -m pytest tests/test_b16_storage_gpu.py -q -p no:cacheprovider -k "conv_forward" 2>&1 | tail -1
for i in 1 2; do python bench.py --model phiseg3d --steps 20 --warmup 5 --skip-cpu --no-profile 2>/dev/null | tail -1 | cut -c1-200; done
