cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for L in 1 2; do
UZ_LANES=$L rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt$L -- python bench.py --steps 3 --warmup 3 --skip-cpu --no-profile > gpurun_out/kt$L.log 2>&1
done
ls -R gpurun_out/kt2 | head
