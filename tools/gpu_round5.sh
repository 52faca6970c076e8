# Round-5 evidence in one GPU-box call (summaries -> gpurun_out/, copied to profiles/ by tools/collect_round.py 5 or by hand):
# test tier in the three math modes, bench lines of every model, rocprofv3 kernel statistics (lane replay and eager), timeline of a
# replayed step, per-layer dispatch table with PMC passes, PMC passes of the bf16-storage volume path, un-profiled lane traces.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
free -g | head -2 > gpurun_out/host.txt; nproc >> gpurun_out/host.txt
bash tools/evidence.sh tier default f32 split
bash tools/evidence.sh bench phiseg unet probunet phiseg3d
python bench.py --model phiseg3d --storage f32 --steps 20 --warmup 5 --skip-cpu > gpurun_out/bench_phiseg3d_f32storage.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg3d_f32storage.json
UZ_REPLAY=graph python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/bench_phiseg_graph_replay.json 2>/dev/null; cut -c1-160 gpurun_out/bench_phiseg_graph_replay.json
UZ_BENCH_SINGLE_DEVICE=1 UZ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --skip-cpu --no-profile --no-f32-leg > gpurun_out/bench_2ranks_one_device.json 2> gpurun_out/bench_2ranks.err; cut -c1-250 gpurun_out/bench_2ranks_one_device.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 python tools/nccl_world1_check.py > gpurun_out/nccl_world1.log 2>&1; echo "nccl rc=$?"; tail -4 gpurun_out/nccl_world1.log
bash tools/prof_round.sh 5 > gpurun_out/prof_round.log 2>&1; tail -5 gpurun_out/prof_round.log
bash tools/trace_step.sh r5 > gpurun_out/trace_r5.log 2>&1; tail -22 gpurun_out/trace_r5.log; cp gpurun_out/timeline_r5.json gpurun_out/r5_timeline_lanes.json; mv gpurun_out/trace_r5.csv.gz gpurun_out/r5_kernel_trace_lanes.csv.gz
# lane traces of the static schedule and of the profile-guided one (Engine.tune_schedule, what bench.py times), with their critical paths
python tools/lane_trace.py fwd gpurun_out/r5_lane_trace_fwd.json > gpurun_out/r5_lane_trace_fwd.txt 2>&1; python tools/lane_trace.py bwd gpurun_out/r5_lane_trace_bwd.json > gpurun_out/r5_lane_trace_bwd.txt 2>&1; head -3 gpurun_out/r5_lane_trace_bwd.txt | tail -1
UZ_TUNE_SCHEDULE=8 python tools/lane_trace.py fwd gpurun_out/r5_lane_trace_fwd_tuned.json > gpurun_out/r5_lane_trace_fwd_tuned.txt 2>&1; UZ_TUNE_SCHEDULE=8 python tools/lane_trace.py bwd gpurun_out/r5_lane_trace_bwd_tuned.json > gpurun_out/r5_lane_trace_bwd_tuned.txt 2>&1
for w in fwd bwd; do python tools/lane_critical_path.py gpurun_out/r5_lane_trace_${w}_tuned.json $w > gpurun_out/r5_lane_critical_path_${w}_tuned.txt 2>&1; head -2 gpurun_out/r5_lane_critical_path_${w}_tuned.txt; done
bash tools/ab_tune.sh > gpurun_out/r5_ab_schedule_tuning.txt 2>&1; grep -c ' : ' gpurun_out/r5_ab_schedule_tuning.txt
bash tools/prof_layers.sh 5 20 20 > gpurun_out/prof_layers.log 2>&1; tail -24 gpurun_out/prof_layers.log | cut -c1-230
for m in unet probunet phiseg3d; do bash tools/evidence.sh stats $m > /dev/null 2>&1; done; ls gpurun_out/r5_bench_kernel_stats_graph_*.csv
bash tools/prof_b16.sh 5 > gpurun_out/prof_b16.log 2>&1; tail -4 gpurun_out/prof_b16.log | cut -c1-200
python tools/bench_coresidency.py > gpurun_out/r5_coresidency.txt 2>&1; tail -4 gpurun_out/r5_coresidency.txt
python tools/soak_train.py 300 > gpurun_out/r5_soak_300_steps.log 2>&1; tail -3 gpurun_out/r5_soak_300_steps.log
