cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do
  for c in 1 3 8 25; do echo -n "slabcost $c: "; UZ_WG_SLABCOST=$c $B 2>/dev/null | tail -1 | cut -c60-100; done
  echo -n "decouple 0: "; UZ_DECOUPLE_WGRAD=0 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "decouple 8192: "; UZ_DECOUPLE_WGRAD=8192 $B 2>/dev/null | tail -1 | cut -c60-100
  echo -n "decouple 32768: "; UZ_DECOUPLE_WGRAD=32768 $B 2>/dev/null | tail -1 | cut -c60-100
done
