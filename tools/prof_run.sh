# rocprofv3 kernel statistics of the default bench command (graph replay); summaries are copied to profiles/ by hand
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_graph gpurun_out/prof_eager
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_graph -- python bench.py --steps 10 --warmup 5 --skip-cpu --no-profile > gpurun_out/prof_graph_line.json 2> gpurun_out/prof_graph.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_eager -- python bench.py --steps 10 --warmup 5 --skip-cpu --no-profile --no-graphs > gpurun_out/prof_eager_line.json 2> gpurun_out/prof_eager.err
ls gpurun_out/prof_graph/*/ gpurun_out/prof_eager/*/
