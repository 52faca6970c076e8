cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/prof_layers.sh 3 20 20 > gpurun_out/prof_layers.log 2>&1; tail -24 gpurun_out/prof_layers.log | cut -c1-230
python bench.py --steps 20 --warmup 5 --skip-cpu --no-f32-leg > gpurun_out/bench_after_layers.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_after_layers.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], {k:r[k] for k in ("op","layer","avg_launch_ms","achieved","frac","traffic","profiled") if k in r})
PY
