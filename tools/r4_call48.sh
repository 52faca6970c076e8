cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do
  echo -n "base: "; $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "splitk_max 1: "; UZ_SPLITK_MAX=1 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "splitk_max 2: "; UZ_SPLITK_MAX=2 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "splitk_grid 96: "; UZ_SPLITK_GRID=96 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "split16_grid 128: "; UZ_SPLIT16_GRID=128 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "msub_small 0: "; UZ_MSUB_SMALL=0 $B 2>/dev/null | tail -1 | cut -c60-100
done
