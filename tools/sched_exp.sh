cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run UZ_X=0
run UZ_SCHED_STREAM_LIGHT=1
run UZ_LANES=3
run UZ_LANES=3 UZ_SCHED_STREAM_LIGHT=1
run UZ_LANES=4 UZ_SCHED_STREAM_LIGHT=1
run UZ_DIAG_SKIP_SMALL=8
run UZ_DIAG_SKIP_SMALL=16
