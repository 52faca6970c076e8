cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf gpurun_out/kt_graph
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_graph -- python bench.py --steps 6 --warmup 4 --skip-cpu --no-profile --no-f32-leg > gpurun_out/kt_graph_line.json 2> gpurun_out/kt_graph.err
F=$(ls gpurun_out/kt_graph/*/*kernel_trace.csv | head -1)
python tools/timeline.py $F gpurun_out/r3_timeline_graph.json
python tools/kt_overlap.py $F
python tools/kt_fill.py $F
gzip -c $F > gpurun_out/r3_kernel_trace_graph.csv.gz; ls -la gpurun_out/r3_kernel_trace_graph.csv.gz
rm -rf gpurun_out/kt_graph
python -m pytest tests/test_ops_gpu.py -q -x -k "fused_bn" -p no:cacheprovider 2>&1 | tail -3
