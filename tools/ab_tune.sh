# A/B of the scheduler's cost model and of the profile-guided schedule (Engine.tune_schedule) on the training step of every model, one box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT


for m in ${MODELS:-phiseg probunet phiseg3d}; do for rep in 1 2; do
for v in "UZ_SCHED_COST=alone UZ_TUNE_SCHEDULE=0" "UZ_SCHED_COST=beside UZ_TUNE_SCHEDULE=0" "UZ_SCHED_COST=beside UZ_TUNE_SCHEDULE=3" "UZ_SCHED_COST=beside UZ_TUNE_SCHEDULE=8"; do
  echo -n "$m $v : "; env $v python bench.py --model $m --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: v for k, v in d['config'].get('schedule',{}).items() if k != 'note'})"
done; done; done
