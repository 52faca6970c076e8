#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export UZ_LIB=$PWD/tools/tmp/libuz_hip_old.so UZ_DP_TABLES=0
{ echo "== old lib, no debug"; UZ_DIAG_STEPS=3 python tools/diag_dp_race.py 2>/dev/null | grep "^step" | cut -c1-120
  echo "== old lib, debug"; UZ_DEBUG_BIL=1 UZ_DIAG_STEPS=3 python tools/diag_dp_race.py 2> gpurun_out/r4_bil_debug.txt | grep "^step" | cut -c1-120
  grep -c "^bil" gpurun_out/r4_bil_debug.txt
} > gpurun_out/r4_call89.txt 2>&1
python - <<'PY' >> gpurun_out/r4_call89.txt
cur=None; d={}
for ln in open('gpurun_out/r4_bil_debug.txt'):
    if ln.startswith('== net'): cur=ln.strip(); d[cur]=[]
    elif ln.startswith('bil') and cur: d[cur].append(ln.strip())
for k,v in d.items(): print(k, len(v), 'misaligned:', [x for x in v if 'dy16=0' not in x or 'dx8=0' not in x][:6])
ks=list(d)
for a,b in zip(ks[0::2], ks[1::2]):
    print(a,'vs',b,'same list:', d[a]==d[b])
PY
