cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do
  echo -n "base: "; $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "splitk_grid 256: "; UZ_SPLITK_GRID=256 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "splitk_grid 400: "; UZ_SPLITK_GRID=400 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "splitk_grid 256 max 2: "; UZ_SPLITK_GRID=256 UZ_SPLITK_MAX=2 $B 2>/dev/null | tail -1 | cut -c60-100
done
