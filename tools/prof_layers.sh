# Per-layer dispatch table (profiles/r<N>_layer_table.json): kernel trace + three PMC passes over tools/layer_profile.py.
# usage (on the GPU box): bash tools/prof_layers.sh <round> [top] [reps]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-3}; TOP=${2:-20}; REPS=${3:-5}
O=gpurun_out/layers
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python tools/layer_profile.py $O/ops.json $TOP $REPS > $O/trace.log 2>&1
export UZ_PROFILE_OPS=$GRAFT_REPO_ROOT/$O/ops.json
export UZ_PROFILE_FLUSH=1      # PMC passes: every launch starts with a cold Infinity Cache (tools/layer_profile.py:cold)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python tools/layer_profile.py $O/ops_fetch.json $TOP $REPS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python tools/layer_profile.py $O/ops_write.json $TOP $REPS > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python tools/layer_profile.py $O/ops_mfma.json $TOP $REPS > $O/mfma.log 2>&1
tail -2 $O/trace.log
python tools/layer_table.py $O/ops.json $O/trace $O/fetch $O/write $O/mfma gpurun_out/r${R}_layer_table.json
# the op lists of the four runs must agree (the ranking is by a live timing)
python - <<PY
import json
a=[(o["tape"],o["index"]) for o in json.load(open("$O/ops.json"))["ops"]]
for n in ("fetch","write","mfma"):
    b=[(o["tape"],o["index"]) for o in json.load(open("$O/ops_%s.json"%n))["ops"]]
    print(n, "same op list:", a==b, "same set:", set(a)==set(b))
PY
F=$(ls $O/trace/*/*kernel_trace.csv | head -1); gzip -c $F > gpurun_out/r${R}_layer_kernel_trace.csv.gz
rm -rf $O/trace $O/fetch $O/write $O/mfma
