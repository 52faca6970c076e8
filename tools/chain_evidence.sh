# Deep-level chain launch (csrc/chain.hip, UZ_CHAIN=8192; off by default): everything the round-6 notes quote, in one script (GPU box).
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config'].get('schedule') or {}; t=s.get('tape_us') or {}; print(d['value'], 'images/s', d['ms_per_step'], 'ms | tapes us: fwd', min(t.get('fwd',[0])), 'bwd', min(t.get('bwd',[0])), '| chain', d['config'].get('chain'))"; }
echo "== barrier micro-benchmark: a phase with (almost) no work, hand-offs by sc1 accesses / by agent-scope fences"
python tools/chain_barrier_bench.py 2>&1 | grep G=
UZ_CHAIN_SC1=0 python tools/chain_barrier_bench.py 2>&1 | grep G= | sed 's/^/fence: /'
echo "== parity: forward + backward chains against the per-op tape and the reference digest"
python tools/chain_check.py 2>&1 | grep -v "Warn\|dbrows\|detach" | tail -14
UZ_CHAIN=8192 TOP=4 python tools/chain_gradcheck.py 2>&1 | tail -6
echo "== per-phase timeline of the forward chain (256 workgroups, alone on the chip)"
UZ_CHAIN=8192 python tools/chain_profile.py 2>&1 | grep -v Warn | tail -90
echo "== per-phase timeline of the first backward chain"
UZ_CHAIN=8192 TAPE=bwd python tools/chain_profile.py 2>&1 | grep -v Warn | tail -40
echo "== step: per-op tape against chains"
for c in "UZ_CHAIN=0" "UZ_CHAIN=8192 UZ_CHAIN_BWD=0" "UZ_CHAIN=8192" "UZ_CHAIN=2048" "UZ_CHAIN=512" "UZ_CHAIN=8192 UZ_CHAIN_NETS=prior" "UZ_CHAIN=8192 UZ_CHAIN_SC1=0"; do
  echo -n "$c : "; env $c python bench.py --steps 20 --warmup 5 --skip-cpu --no-f32-leg --no-profile 2>/dev/null | line
done
echo "== what-if (diagnostic build, results are garbage): the step with classes of small-plane ops SKIPPED = the ceiling of anything that makes them cheaper"
if [ -f unet-zoo_amd/libuz_hip_diag.so ]; then
  export UZ_LIB=$PWD/unet-zoo_amd/libuz_hip_diag.so
  for c in "UZ_X=0" "UZ_DIAG_SKIP=wgrad:8" "UZ_DIAG_SKIP=wgrad:16" "UZ_DIAG_SKIP=bnf:16,fwd:16" "UZ_DIAG_SKIP=bnb:16,dgrad:16,wgrad:16" "UZ_DIAG_SKIP=bn:16,conv:16,resample:8" "UZ_DIAG_SKIP=bn:8,conv:8,resample:4"; do
    echo -n "$c : "; env $c python bench.py --allow-experiment --steps 20 --warmup 5 --skip-cpu --no-f32-leg --no-profile 2>/dev/null | line
  done
else echo "(no diagnostic library: make -C unet-zoo_amd/csrc VARIANT=diag XFLAGS=-DUZ_DIAG)"; fi
