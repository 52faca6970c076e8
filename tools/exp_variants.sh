# usage: exp_variants.sh "variant variant ..." : A/B of experiment builds (csrc/Makefile VARIANT=...) against the product library -
# per-layer convolution timings and the PHiSeg step.  Timing only: the experiment builds need not compute correct values.
cd $GRAFT_REPO_ROOT
SHAPES=("224 128 128 128" "128 128 128 128" "192 192 64 64" "192 192 32 32" "128 128 32 32" "192 192 16 16" "256 256 16 16" "64 64 64 64" "32 32 128 128")
for v in base $1; do
  if [ $v = base ]; then unset UZ_LIB; else export UZ_LIB=$GRAFT_REPO_ROOT/unet-zoo_amd/libuz_hip_$v.so; fi
  echo "######## $v"
  for s in "${SHAPES[@]}"; do echo "== $s"; python tools/bench_conv.py $s 32 3 10 2>/dev/null; done
  for r in 1 2; do python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-175; done
done
