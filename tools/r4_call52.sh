cd $GRAFT_REPO_ROOT
export UZ_LIB=$GRAFT_REPO_ROOT/unet-zoo_amd/libuz_hip_pair.so
python -m pytest tests/test_ops_gpu.py tests/test_split_storage_gpu.py -m gpu -q -p no:cacheprovider -x -k "conv or split or unit" 2>&1 | tail -3
unset UZ_LIB
for lib in "" unet-zoo_amd/libuz_hip_pair.so; do
  echo "== lib [$lib]"
  for shape in "224 128 128 128 32" "128 128 128 128 32" "256 192 64 64 32" "192 192 32 32 32" "32 32 128 128 32"; do
    echo -n "$shape: "; UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/bench_conv.py $shape 3 8 2>/dev/null | grep -E "fwd|dgrad" | tr '\n' ' '; echo
  done
done
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2 3; do
  echo -n "base: "; $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "pair: "; UZ_LIB=$GRAFT_REPO_ROOT/unet-zoo_amd/libuz_hip_pair.so $B 2>/dev/null | tail -1 | cut -c60-100
done
