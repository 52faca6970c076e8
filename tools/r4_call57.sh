cd $GRAFT_REPO_ROOT
UZ_BENCH_SINGLE_DEVICE=1 UZ_BENCH_BACKEND=gloo timeout 600 python bench.py --model phiseg3d --gpus 2 --steps 6 --warmup 2 --skip-cpu --no-profile --no-f32-leg > gpurun_out/bench_3d_2ranks.json 2> gpurun_out/bench_3d_2ranks.err; echo "rc=$?"; cut -c1-400 gpurun_out/bench_3d_2ranks.json; tail -5 gpurun_out/bench_3d_2ranks.err
