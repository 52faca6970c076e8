cd $GRAFT_REPO_ROOT
for t in 256 128 64 512; do
  echo "######## target $t"
  for s in "64 64 64 64" "64 64 32 32" "128 128 32 32" "192 192 32 32" "320 192 32 32" "192 192 16 16" "256 256 16 16" "192 192 64 64"; do
    echo -n "$s : "; UZ_WGS_TARGET=$t python tools/bench_conv.py $s 32 3 10 wgrad 2>/dev/null | tail -1
  done
done
