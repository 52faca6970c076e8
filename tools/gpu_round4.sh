# Round-4 evidence in one GPU-box call: test tier in the three math modes, benches of every model, rocprofv3 kernel statistics
# (graph replay and eager), timeline of a replayed step, PMC passes on the dominant layer, per-layer dispatch table.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
free -g | head -2 > gpurun_out/host.txt; nproc >> gpurun_out/host.txt
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_default.log 2>&1; echo "pytest default rc=$?"; tail -3 gpurun_out/pytest_gpu_default.log
UZ_CONV_MATH=f32 python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_f32.log 2>&1; echo "pytest f32 rc=$?"; tail -3 gpurun_out/pytest_gpu_f32.log
UZ_CONV_MATH=split python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_split.log 2>&1; echo "pytest split rc=$?"; tail -4 gpurun_out/pytest_gpu_split.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_phiseg.json 2> gpurun_out/bench_phiseg.err; echo "bench phiseg rc=$?"; cut -c1-200 gpurun_out/bench_phiseg.json
for m in unet probunet phiseg3d; do
  python bench.py --model $m --steps 20 --warmup 5 > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err; echo "bench $m rc=$?"; cut -c1-200 gpurun_out/bench_$m.json
done
python bench.py --model phiseg3d --storage f32 --steps 20 --warmup 5 --skip-cpu > gpurun_out/bench_phiseg3d_f32storage.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg3d_f32storage.json
python bench.py --model phiseg3d --conv-math default --steps 10 --warmup 3 --skip-cpu --no-profile > gpurun_out/bench_phiseg3d_f32split.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg3d_f32split.json
python bench.py --conv-math bf16 --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/bench_phiseg_bf16math.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg_bf16math.json
python bench.py --model phiseg3d --reversible --steps 10 --warmup 3 --skip-cpu --no-profile > gpurun_out/bench_phiseg3d_rev.json 2> gpurun_out/bench_phiseg3d_rev.err; cut -c1-200 gpurun_out/bench_phiseg3d_rev.json
python bench.py --model unet --cpu-batch 4 --steps 5 --warmup 2 --no-f32-leg --no-profile > gpurun_out/bench_unet_cpu_b4.json 2>/dev/null; python -c "import json; d=json.loads(open('gpurun_out/bench_unet_cpu_b4.json').read().strip().splitlines()[-1]); print('config 1 (Unet-4 B=4 CPU oracle on this box):', d['cpu_baseline'])"
UZ_BENCH_SINGLE_DEVICE=1 UZ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --skip-cpu --no-profile --no-f32-leg > gpurun_out/bench_2ranks_one_device.json 2> gpurun_out/bench_2ranks.err; cut -c1-250 gpurun_out/bench_2ranks_one_device.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 python tools/nccl_world1_check.py > gpurun_out/nccl_world1.log 2>&1; echo "nccl rc=$?"; tail -4 gpurun_out/nccl_world1.log
bash tools/prof_round.sh 4 > gpurun_out/prof_round.log 2>&1; tail -5 gpurun_out/prof_round.log
rm -rf gpurun_out/kt_graph
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_graph -- python bench.py --steps 6 --warmup 4 --skip-cpu --no-profile --no-f32-leg > gpurun_out/kt_graph_line.json 2> gpurun_out/kt_graph.err
F=$(ls gpurun_out/kt_graph/*/*kernel_trace.csv | head -1)
python tools/timeline.py $F gpurun_out/r4_timeline_graph.json | head -20
gzip -c $F > gpurun_out/r4_kernel_trace_graph.csv.gz; rm -rf gpurun_out/kt_graph
bash tools/prof_layers.sh 4 20 20 > gpurun_out/prof_layers.log 2>&1; tail -24 gpurun_out/prof_layers.log | cut -c1-230
python tools/soak_train.py 300 > gpurun_out/r4_soak_300_steps.log 2>&1; tail -3 gpurun_out/r4_soak_300_steps.log
# kernel statistics of the other three models' bench runs (profiles/r4_bench_kernel_stats_graph_<model>.csv)
for m in unet probunet phiseg3d; do
  rm -rf gpurun_out/prof_$m
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -- python bench.py --model $m --steps 10 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/prof_${m}_line.json 2>/dev/null
  cp $(ls gpurun_out/prof_$m/*/*kernel_stats.csv | head -1) gpurun_out/r4_bench_kernel_stats_graph_$m.csv; rm -rf gpurun_out/prof_$m
done
# bf16-storage volume path: kernel statistics of its bench run and the PMC passes on its heaviest layer
bash tools/prof_b16.sh 4 > gpurun_out/prof_b16.log 2>&1; tail -4 gpurun_out/prof_b16.log | cut -c1-200
