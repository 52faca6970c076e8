cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do
  echo -n "base: "; $B 2>/dev/null | tail -1 | cut -c60-100
  for k in 0.5 1 2 4; do echo -n "k $k: "; UZ_WGS_K=$k $B 2>/dev/null | tail -1 | cut -c60-100; done
  echo -n "k 1 minwg 32: "; UZ_WGS_K=1 UZ_WGS_MINWG=32 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "k 1 minwg 128: "; UZ_WGS_K=1 UZ_WGS_MINWG=128 $B 2>/dev/null | tail -1 | cut -c60-100
done
