cd $GRAFT_REPO_ROOT
python -m pytest tests/test_b16_storage_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-400
python bench.py --model phiseg3d --steps 20 --warmup 5 --skip-cpu > gpurun_out/bench_phiseg3d_b.json 2> gpurun_out/bench_phiseg3d_b.err; echo "bench phiseg3d rc=$?"; cut -c1-200 gpurun_out/bench_phiseg3d_b.json
bash tools/prof_b16.sh 4 2>&1 | tail -22
python tools/pmc_b16_to_json.py 4 gpurun_out/r4_pmc_b16_fetch_size_summary.txt gpurun_out/r4_pmc_b16_write_size_summary.txt > /dev/null; cp profiles/r4_pmc_traffic_b16.json gpurun_out/
