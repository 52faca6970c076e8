cd $GRAFT_REPO_ROOT
python bench.py --model phiseg3d --steps 20 --warmup 5 > gpurun_out/bench_phiseg3d.json 2> gpurun_out/bench_phiseg3d.err; echo "bench phiseg3d rc=$?"; cut -c1-200 gpurun_out/bench_phiseg3d.json
python bench.py --model phiseg3d --storage f32 --steps 20 --warmup 5 --skip-cpu > gpurun_out/bench_phiseg3d_f32storage.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg3d_f32storage.json
python bench.py --model phiseg3d --reversible --steps 10 --warmup 3 --skip-cpu --no-profile > gpurun_out/bench_phiseg3d_rev.json 2> gpurun_out/bench_phiseg3d_rev.err; cut -c1-200 gpurun_out/bench_phiseg3d_rev.json
python bench.py --model phiseg3d --conv-math default --steps 10 --warmup 3 --skip-cpu --no-profile > gpurun_out/bench_phiseg3d_f32split.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg3d_f32split.json
bash tools/prof_b16.sh 4 > gpurun_out/prof_b16.log 2>&1; tail -3 gpurun_out/prof_b16.log | cut -c1-200
python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | cut -c1-160
