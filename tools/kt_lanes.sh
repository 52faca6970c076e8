# kernel traces of the bench under different lane counts (tools/kt_fill.py, tools/kt_overlap.py analyse them)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for L in 2; do
rm -rf gpurun_out/kt$L
UZ_LANES=$L rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt$L -- python bench.py --steps 3 --warmup 3 --skip-cpu --no-profile --no-f32-leg > gpurun_out/kt$L.log 2>&1
done
