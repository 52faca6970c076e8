# fp32-MFMA-only leg: kernel statistics of the replayed bench (which launches make up the 35 ms step)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_f32
export UZ_CONV_MATH=f32
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f32 -- python bench.py --steps 10 --warmup 5 --skip-cpu --no-profile --no-f32-leg > gpurun_out/prof_f32_line.json 2> gpurun_out/prof_f32.err
cut -c1-200 gpurun_out/prof_f32_line.json
cp $(ls gpurun_out/prof_f32/*/*kernel_stats.csv | head -1) gpurun_out/r4_bench_kernel_stats_graph_f32.csv
gzip -c $(ls gpurun_out/prof_f32/*/*kernel_trace.csv | head -1) > gpurun_out/r4_kernel_trace_graph_f32.csv.gz
rm -rf gpurun_out/prof_f32
head -30 gpurun_out/r4_bench_kernel_stats_graph_f32.csv | cut -c1-200
