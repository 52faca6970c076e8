# One GPU-box call: full -m gpu test tier (no -x: collect every failure), key layer timings, then the bench of the three models.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
free -g | head -2 > gpurun_out/host.txt; nproc >> gpurun_out/host.txt
python -m pytest tests -m gpu -q -rA --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for l in "224 128 128 128" "128 128 128 128" "256 192 64 64" "192 192 64 64" "32 32 128 128" "64 64 64 64" "128 128 32 32" "192 192 16 16"; do
  echo "== $l" >> gpurun_out/bench_conv.log; python tools/bench_conv.py $l >> gpurun_out/bench_conv.log 2>&1
done
cat gpurun_out/bench_conv.log
for m in phiseg unet probunet phiseg3d; do
  python bench.py --model $m --steps 20 --warmup 5 > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err
  echo "bench $m rc=$?"; cut -c1-300 gpurun_out/bench_$m.json
done
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 python tools/nccl_world1_check.py > gpurun_out/nccl_world1.log 2>&1; echo "nccl rc=$?"; tail -6 gpurun_out/nccl_world1.log
python bench.py --model phiseg3d --reversible --steps 10 --warmup 3 --skip-cpu --no-profile > gpurun_out/bench_phiseg3d_rev.json 2> gpurun_out/bench_phiseg3d_rev.err; echo "bench phiseg3d rev rc=$?"; cut -c1-300 gpurun_out/bench_phiseg3d_rev.json
for q in "2 2" "3 3" "4 4" "2 4"; do set -- $q; echo "== hw queues $1 graph queues $2"; GPU_MAX_HW_QUEUES=$1 DEBUG_HIP_FORCE_GRAPH_QUEUES=$2 python bench.py --skip-cpu --no-profile --no-f32-leg --steps 30 | cut -c90-200; done
