cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo "== new"; python tools/bench_resample.py 2>&1 | grep bilinear | cut -c1-120
echo "== old"; UZ_LIB=$GRAFT_REPO_ROOT/unet-zoo_amd/libuz_hip_bilold.so python tools/bench_resample.py 2>&1 | grep bilinear | cut -c1-120
done
