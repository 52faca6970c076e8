cd $GRAFT_REPO_ROOT
export UZ_CONV_MATH=f32
B="python bench.py --steps 15 --warmup 3 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do
  echo -n "base: "; $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "slabcost 3: "; UZ_WG_SLABCOST=3 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "slabcost 0.3: "; UZ_WG_SLABCOST=0.3 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "lanes 3: "; UZ_LANES=3 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "lanes 1: "; UZ_LANES=1 $B 2>/dev/null | tail -1 | cut -c60-100
done
