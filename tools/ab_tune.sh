# A/B of the scheduler's cost model and of the profile-guided schedule (Engine.tune_schedule) on the training step of every model, one box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_phiseg_gpu.py -q -x -m gpu -k "tuned_schedule" -s 2>&1 | tail -4 | cut -c1-400
UZ_BENCH_SINGLE_DEVICE=1 UZ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --skip-cpu --no-profile --no-f32-leg 2>/tmp/dp.err | cut -c1-200; tail -3 /tmp/dp.err
for m in ${MODELS:-phiseg probunet phiseg3d unet}; do for rep in 1 2; do
for v in "UZ_SCHED_COST=alone UZ_TUNE_SCHEDULE=0" "UZ_SCHED_COST=beside UZ_TUNE_SCHEDULE=0" "UZ_SCHED_COST=beside UZ_TUNE_SCHEDULE=3" "UZ_SCHED_COST=beside UZ_TUNE_SCHEDULE=6"; do
  echo -n "$m $v : "; env $v python bench.py --model $m --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: v for k, v in d['config'].get('schedule',{}).items() if k != 'note'})"
done; done; done
