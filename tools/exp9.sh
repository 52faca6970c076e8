cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/prof_layers.sh 3 20 5 > gpurun_out/prof_layers.log 2>&1; tail -30 gpurun_out/prof_layers.log
