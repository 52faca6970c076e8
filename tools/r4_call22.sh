cd $GRAFT_REPO_ROOT
for v in base lb3; do
  if [ $v = base ]; then unset UZ_LIB; else export UZ_LIB=$GRAFT_REPO_ROOT/unet-zoo_amd/libuz_hip_$v.so; fi
  echo "######## $v"
  for s in "32 32 128 128" "32 96 128 128" "96 32 128 128" "32 64 64 64"; do echo "== $s"; python tools/bench_conv_packed.py $s 32 10 0.5 2>/dev/null | tail -2; done
  for r in 1 2; do python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | cut -c1-140; done
  python bench.py --model unet --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | cut -c1-140
done
