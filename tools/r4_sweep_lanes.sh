cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
echo -n "default: "; b
for l in 1 2 3 4; do echo -n "UZ_LANES=$l: "; UZ_LANES=$l b; done
for q in 2 3 4 6; do echo -n "queues=$q lanes2: "; GPU_MAX_HW_QUEUES=$q DEBUG_HIP_FORCE_GRAPH_QUEUES=$q b; done
for q in 3 4; do echo -n "queues=$q lanes3: "; UZ_LANES=3 GPU_MAX_HW_QUEUES=$q DEBUG_HIP_FORCE_GRAPH_QUEUES=$q b; done
echo -n "decouple all (px huge): "; UZ_DECOUPLE_WGRAD=100000000 b
echo -n "decouple none: "; UZ_DECOUPLE_WGRAD=0 UZ_DECOUPLE_PREFIX="" b
echo -n "decouple prefixes post+prior+lik: "; UZ_DECOUPLE_PREFIX="likelihood,posterior,prior" b
echo -n "default again: "; b
