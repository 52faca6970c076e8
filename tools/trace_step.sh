# kernel trace of a replayed PHiSeg step; usage: bash tools/trace_step.sh <tag> [ENV=val ...]   -> gpurun_out/trace_<tag>.csv.gz + timeline summary
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=$1; shift
rm -rf gpurun_out/kt_$tag
env "$@" rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$tag -- python bench.py --steps 6 --warmup 4 --skip-cpu --no-profile --no-f32-leg > gpurun_out/kt_${tag}_line.json 2> gpurun_out/kt_$tag.err
F=$(ls gpurun_out/kt_$tag/*/*kernel_trace.csv | head -1)
python tools/timeline.py $F gpurun_out/timeline_$tag.json | head -20
gzip -c $F > gpurun_out/trace_$tag.csv.gz; rm -rf gpurun_out/kt_$tag
cut -c1-150 gpurun_out/kt_${tag}_line.json
