cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider > gpurun_out/pytest_gpu_default.log 2>&1; echo "pytest default rc=$?"; tail -2 gpurun_out/pytest_gpu_default.log
