# One GPU-box call: full -m gpu test tier (no -x: collect every failure), then the bench of the three BASELINE models.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rA --timeout=1500 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for m in phiseg unet probunet; do
  python bench.py --model $m --steps 20 --warmup 5 > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err
  echo "bench $m rc=$?"; cut -c1-400 gpurun_out/bench_$m.json
done
