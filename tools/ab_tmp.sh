cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "UZ_WGS_TARGET=128" "UZ_WGS_TARGET=160" "UZ_WGS_TARGET=192" "UZ_WGS_TARGET=256" "UZ_WGS_TARGET=96"; do
  echo -n "$v : "; env $v python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config'].get('schedule',{}); print(d['value'], d['ms_per_step'], s.get('step_ms'), s.get('kept'))"
done; done
