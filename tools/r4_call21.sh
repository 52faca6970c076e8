cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for r in 1 2; do
echo -n "default: "; b
echo -n "wgrad after dgrad: "; UZ_WGRAD_AFTER_DGRAD=1 b
echo -n "wgrad after dgrad + decouple all nets: "; UZ_WGRAD_AFTER_DGRAD=1 UZ_DECOUPLE_PREFIX="likelihood,posterior,prior" b
echo -n "wgrad after dgrad + decouple everything (px): "; UZ_WGRAD_AFTER_DGRAD=1 UZ_DECOUPLE_WGRAD=100000000 b
done
