# Round-6 evidence in one GPU-box call (summaries -> gpurun_out/, copied to profiles/ by tools/collect_round.py 6):
# test tier in the three math modes, bench lines of every model, rocprofv3 kernel statistics, per-layer dispatch table with PMC passes,
# lane traces + critical paths of the tuned schedule, and the deep-level chain launch (csrc/chain.hip): parity, per-phase timeline, barrier
# micro-benchmark, step matrix, and the what-if table of a diagnostic build (what the step would cost if the small-plane ops were free).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ROUND=6
free -g | head -2 > gpurun_out/host.txt; nproc >> gpurun_out/host.txt
bash tools/evidence.sh tier default f32 split
bash tools/evidence.sh bench phiseg unet probunet phiseg3d
python bench.py --model phiseg3d --storage f32 --steps 20 --warmup 5 --skip-cpu > gpurun_out/bench_phiseg3d_f32storage.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg3d_f32storage.json
UZ_REPLAY=graph python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/bench_phiseg_graph_replay.json 2>/dev/null; cut -c1-160 gpurun_out/bench_phiseg_graph_replay.json
UZ_BENCH_SINGLE_DEVICE=1 UZ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --skip-cpu --no-profile --no-f32-leg > gpurun_out/bench_2ranks_one_device.json 2> gpurun_out/bench_2ranks.err; cut -c1-250 gpurun_out/bench_2ranks_one_device.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 python tools/nccl_world1_check.py > gpurun_out/nccl_world1.log 2>&1; echo "nccl rc=$?"; tail -4 gpurun_out/nccl_world1.log
bash tools/prof_round.sh 6 > gpurun_out/prof_round.log 2>&1; tail -5 gpurun_out/prof_round.log
UZ_TUNE_SCHEDULE=8 python tools/lane_trace.py fwd gpurun_out/r6_lane_trace_fwd_tuned.json > gpurun_out/r6_lane_trace_fwd_tuned.txt 2>&1; UZ_TUNE_SCHEDULE=8 python tools/lane_trace.py bwd gpurun_out/r6_lane_trace_bwd_tuned.json > gpurun_out/r6_lane_trace_bwd_tuned.txt 2>&1
for w in fwd bwd; do python tools/lane_critical_path.py gpurun_out/r6_lane_trace_${w}_tuned.json $w > gpurun_out/r6_lane_critical_path_${w}_tuned.txt 2>&1; head -2 gpurun_out/r6_lane_critical_path_${w}_tuned.txt; done
bash tools/prof_layers.sh 6 20 20 > gpurun_out/prof_layers.log 2>&1; tail -24 gpurun_out/prof_layers.log | cut -c1-230
for m in unet probunet phiseg3d; do bash tools/evidence.sh stats $m > /dev/null 2>&1; done; ls gpurun_out/r6_bench_kernel_stats_graph_*.csv
# ---- the chain launch (off by default)
bash tools/chain_evidence.sh > gpurun_out/r6_chain_evidence.txt 2>&1; tail -30 gpurun_out/r6_chain_evidence.txt
python tools/soak_train.py 300 > gpurun_out/r6_soak_300_steps.log 2>&1; tail -3 gpurun_out/r6_soak_300_steps.log
