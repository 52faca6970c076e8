cd $GRAFT_REPO_ROOT
for lib in "" unet-zoo_amd/libuz_hip_wghalf.so; do
  echo "== lib [$lib]"
  for shape in "288 96 128 64 128" "192 64 128 64 128" "576 192 64 32 64"; do
    echo -n "bf16 $shape: "; UZ_CONV_MATH=bf16 UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/bench_conv.py $shape 3 8 wgrad 2>/dev/null | grep -E "wgrad" | tr '\n' ' '; echo
  done
  for shape in "224 128 128 128 32" "192 192 64 64 32" "192 192 32 32 32"; do
    echo -n "split $shape: "; UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/bench_conv.py $shape 3 8 wgrad 2>/dev/null | grep -E "wgrad" | tr '\n' ' '; echo
  done
done
