cd $GRAFT_REPO_ROOT
export UZ_CONV_MATH=bf16
for lib in "" unet-zoo_amd/libuz_hip_halfloads.so unet-zoo_amd/libuz_hip_halffrag.so; do
  echo "== lib [$lib]"
  for shape in "288 96 128 64 128" "96 288 128 64 128" "576 192 32 16 32" "192 64 128 64 128"; do
    echo "-- $shape"; UZ_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/bench_conv.py $shape 3 8 fwd 2>/dev/null | tail -1
  done
done
