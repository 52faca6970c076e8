cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for l in "224 128 128 128" "128 128 128 128" "256 192 64 64" "192 192 64 64" "64 64 64 64" "128 128 32 32" "192 192 32 32" "256 256 16 16" "192 192 16 16"; do
  echo "== $l"; python tools/bench_conv.py $l 32 3 5 wgrad 2>&1 | tail -1
done
