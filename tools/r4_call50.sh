cd $GRAFT_REPO_ROOT
for lib in "" unet-zoo_amd/libuz_hip_b64.so; do
  echo "== lib [$lib] default math"
  for shape in "224 128 128 128 32" "128 128 128 128 32" "256 192 64 64 32" "192 192 32 32 32"; do
    echo -n "$shape: "; UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/bench_conv.py $shape 3 8 2>/dev/null | grep -E "fwd|dgrad" | tr '\n' ' '; echo
  done
  echo "== lib [$lib] bf16 math"
  for shape in "288 96 128 64 128" "192 64 128 64 128"; do
    echo -n "$shape: "; UZ_CONV_MATH=bf16 UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/bench_conv.py $shape 3 8 2>/dev/null | grep -E "fwd|dgrad" | tr '\n' ' '; echo
  done
done
