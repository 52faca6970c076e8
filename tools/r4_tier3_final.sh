# GPU tier in the three math modes + smoke at HEAD (no benches: profiles/r4_bench_lines.json stays the line of call 101)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_default.log 2>&1; echo "pytest default rc=$?"; tail -2 gpurun_out/pytest_gpu_default.log
UZ_CONV_MATH=f32 python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_f32.log 2>&1; echo "pytest f32 rc=$?"; tail -2 gpurun_out/pytest_gpu_f32.log
UZ_CONV_MATH=split python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_split.log 2>&1; echo "pytest split rc=$?"; tail -3 gpurun_out/pytest_gpu_split.log
python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c1-170
