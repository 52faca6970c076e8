# rocprofv3 evidence for one round: kernel statistics of the default bench command (graph replay and eager) and the three PMC
# passes on the dominant layer.  Summaries land in gpurun_out/; the ones to be judged are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-3}
rm -rf gpurun_out/prof_graph gpurun_out/prof_eager gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_graph -- python bench.py --steps 10 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/prof_graph_line.json 2> gpurun_out/prof_graph.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_eager -- python bench.py --steps 10 --warmup 5 --skip-cpu --no-profile --no-f32-leg --no-graphs > gpurun_out/prof_eager_line.json 2> gpurun_out/prof_eager.err
cp $(ls gpurun_out/prof_graph/*/*kernel_stats.csv | head -1) gpurun_out/r${R}_bench_kernel_stats_graph.csv
cp $(ls gpurun_out/prof_eager/*/*kernel_stats.csv | head -1) gpurun_out/r${R}_bench_kernel_stats_eager.csv
python tools/dominant_from_trace.py gpurun_out/prof_eager gpurun_out/r${R}_dominant_kernel_launches.json > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python tools/pmc_traffic.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python tools/pmc_traffic.py > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python tools/pmc_traffic.py > gpurun_out/pmc_mfma.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_mfma > gpurun_out/r${R}_pmc_mfma_busy_summary.txt
python tools/pmc_summary.py gpurun_out/pmc_fetch > gpurun_out/r${R}_pmc_fetch_size_summary.txt
python tools/pmc_summary.py gpurun_out/pmc_write > gpurun_out/r${R}_pmc_write_size_summary.txt
head -12 gpurun_out/r${R}_bench_kernel_stats_graph.csv
cat gpurun_out/prof_graph_line.json | cut -c1-200
# keep only the small summaries in the merge-back (traces are large)
rm -rf gpurun_out/prof_graph gpurun_out/prof_eager gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
