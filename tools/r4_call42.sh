cd $GRAFT_REPO_ROOT
export UZ_CONV_MATH=bf16
for lib in "" unet-zoo_amd/libuz_hip_deep.so unet-zoo_amd/libuz_hip_deeppref.so; do
  echo "== lib [$lib]"
  for shape in "288 96 128 64 128" "96 288 128 64 128" "192 64 128 64 128" "96 32 128 64 128" "576 192 64 32 64"; do
    echo "-- $shape"; UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/bench_conv.py $shape 3 8 2>/dev/null | grep -E "fwd|dgrad" | tr '\n' ' '; echo
  done
  UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python -m pytest tests/test_b16_storage_gpu.py -q -p no:cacheprovider -k "conv_forward" 2>&1 | tail -1
done
