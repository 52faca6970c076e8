cd $GRAFT_REPO_ROOT
python -m pytest tests/test_phiseg3d.py tests/test_b16_storage_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "^E  |passed|failed|Error|bf16" | cut -c1-400
python bench.py --model phiseg3d --steps 20 --warmup 5 > gpurun_out/bench_phiseg3d.json 2> gpurun_out/bench_phiseg3d.err; echo "bench phiseg3d rc=$?"; cut -c1-400 gpurun_out/bench_phiseg3d.json
python bench.py --model phiseg3d --storage f32 --steps 20 --warmup 5 --skip-cpu --no-profile > gpurun_out/bench_phiseg3d_f32storage.json 2>/dev/null; cut -c1-300 gpurun_out/bench_phiseg3d_f32storage.json
