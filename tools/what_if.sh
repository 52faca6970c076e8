# What would the step cost if a class of ops were free?  (UZ_DIAG_SKIP, csrc/tape.hip: timing only, results are garbage.)
# Needs a DIAGNOSTIC build of the library - the product build has no skipping code:
#   make -C unet-zoo_amd/csrc VARIANT=diag XFLAGS=-DUZ_DIAG && export UZ_LIB=$PWD/unet-zoo_amd/libuz_hip_diag.so
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; UZ_DIAG_SKIP="$1" python bench.py --allow-experiment --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
python bench.py --allow-experiment --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('baseline', d['ms_per_step'], d['value'])"
run "conv:8"
run "conv:8,bn:8,resample:8"
run "conv:16,bn:16,resample:16"
run "conv:32,bn:32,resample:32"
run "bn:128"
run "bn:128,resample:128"
run "convmin:64"
run "convmin:128"
run "conv:128"
python bench.py --allow-experiment --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('baseline', d['ms_per_step'], d['value'])"
