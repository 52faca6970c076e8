# rocprofv3 evidence for the bf16-storage volume path: kernel statistics of `bench.py --model phiseg3d` and two PMC passes on its heaviest layer
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-4}
rm -rf gpurun_out/prof_b16 gpurun_out/pmc_b16_fetch gpurun_out/pmc_b16_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b16 -- python bench.py --model phiseg3d --steps 10 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/prof_b16_line.json 2> gpurun_out/prof_b16.err
cp $(ls gpurun_out/prof_b16/*/*kernel_stats.csv | head -1) gpurun_out/r${R}_bench_kernel_stats_graph_phiseg3d_b16.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_b16_fetch -- python tools/pmc_traffic_b16.py > gpurun_out/pmc_b16_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_b16_write -- python tools/pmc_traffic_b16.py > gpurun_out/pmc_b16_write.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_b16_fetch > gpurun_out/r${R}_pmc_b16_fetch_size_summary.txt
python tools/pmc_summary.py gpurun_out/pmc_b16_write > gpurun_out/r${R}_pmc_b16_write_size_summary.txt
head -14 gpurun_out/r${R}_bench_kernel_stats_graph_phiseg3d_b16.csv | cut -c1-160
tail -3 gpurun_out/pmc_b16_fetch.log
rm -rf gpurun_out/prof_b16 gpurun_out/pmc_b16_fetch gpurun_out/pmc_b16_write
