cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do
  for t in 256 192 128; do echo -n "target $t: "; UZ_WGS_TARGET=$t $B 2>/dev/null | tail -1 | cut -c60-100; done
done
