cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2 3; do
  echo -n "base: "; $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "chunk 16 prio: "; UZ_WGRAD_TABLE_CHUNK=16 UZ_TABLE_PRIO=1 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "chunk 8 prio: "; UZ_WGRAD_TABLE_CHUNK=8 UZ_TABLE_PRIO=1 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "chunk 32 prio: "; UZ_WGRAD_TABLE_CHUNK=32 UZ_TABLE_PRIO=1 $B 2>/dev/null | tail -1 | cut -c60-100
done
