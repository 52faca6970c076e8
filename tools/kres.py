#!/usr/bin/env python3
"""Print per-kernel register / LDS / spill usage of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I/root/repo/include",
       "-I" + src.rsplit("/", 1)[0], "-fno-gpu-rdc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    print(f"{k[:90]:90s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} sgpr={v.get('TotalSGPRs')} spillV={v.get('VGPRs Spill')} "
          f"scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')} lds={v.get('LDS Size [bytes/block]')}")
