cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=3
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2 UZ_LANES=3
run GPU_MAX_HW_QUEUES=1 UZ_LANES=1
