cd $GRAFT_REPO_ROOT
python -m pytest tests/test_b16_storage_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-400
python bench.py --model phiseg3d --steps 20 --warmup 5 --skip-cpu > gpurun_out/bench_phiseg3d_b.json 2> gpurun_out/bench_phiseg3d_b.err; echo "bench phiseg3d rc=$?"; cut -c1-200 gpurun_out/bench_phiseg3d_b.json
python bench.py --model phiseg3d --storage f32 --steps 20 --warmup 5 --skip-cpu > gpurun_out/bench_phiseg3d_f32storage.json 2>/dev/null; cut -c1-200 gpurun_out/bench_phiseg3d_f32storage.json
