#!/bin/bash
# One parametrised evidence script (replaces the per-call tools/r4_call*.sh pile).  Runs ON the GPU box:
#   gpurun -- 'bash tools/evidence.sh <what> [args]'          outputs under gpurun_out/, summaries are copied to profiles/ by hand
# what:
#   tier [mode...]      smoke + GPU test tier in the given arithmetic modes (default: default f32 split)
#   bench [model...]    bench.py lines (default: phiseg unet probunet phiseg3d)
#   ab VAR=a VAR=b ...  A/B of environment toggles on the training step of $MODEL (default phiseg), three alternations, one process each
#   layers [reps]       per-layer dispatch table with PMC passes (profiles/rN_layer_table.json)
#   stats [model]       rocprofv3 kernel statistics of a bench run (graph replay)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
what=$1; shift
case "$what" in
tier)
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
  modes="$@"; [ -z "$modes" ] && modes="default f32 split"
  for m in $modes; do
    if [ "$m" = default ]; then env= ; else env="UZ_CONV_MATH=$m"; fi
    env $env python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_$m.log 2>&1; echo "pytest $m rc=$?"; tail -3 gpurun_out/pytest_gpu_$m.log
  done ;;
bench)
  models="$@"; [ -z "$models" ] && models="phiseg unet probunet phiseg3d"
  for m in $models; do
    python bench.py --model $m --steps 20 --warmup 5 > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err; echo "bench $m rc=$?"; cut -c1-220 gpurun_out/bench_$m.json
  done ;;
ab)
  for rep in 1 2 3; do for v in "$@"; do
    echo -n "$v : "; env $v python bench.py --model ${MODEL:-phiseg} --steps ${STEPS:-30} --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done; done ;;
layers)
  bash tools/prof_layers.sh ${ROUND:-6} ${1:-20} ${1:-20} > gpurun_out/prof_layers.log 2>&1; tail -24 gpurun_out/prof_layers.log | cut -c1-230 ;;
stats)
  m=${1:-phiseg}; rm -rf gpurun_out/prof_$m
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -- python bench.py --model $m --steps 15 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/prof_${m}_line.json 2>/dev/null
  cp $(ls gpurun_out/prof_$m/*/*kernel_stats.csv | head -1) gpurun_out/r${ROUND:-6}_bench_kernel_stats_graph_$m.csv; rm -rf gpurun_out/prof_$m
  head -25 gpurun_out/r${ROUND:-6}_bench_kernel_stats_graph_$m.csv | cut -c1-160 ;;
*) echo "unknown: $what"; exit 2 ;;
esac
