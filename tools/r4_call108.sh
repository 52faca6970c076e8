cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_line.json 2> gpurun_out/bench_default_line.err; echo "rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_default_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['metric'], d['value'], d['ms_per_step'], d['dtype'], d['config'])
print({k:r[k] for k in ('bound','achieved','peak','unit','frac','traffic','sustained_ceiling_note')})
print(d['cpu_baseline'])
P
