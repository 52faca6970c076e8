"""GPU fill over the last training step of a rocprofv3 --kernel-trace CSV: time with 0 / 1 / 2 / 3+ kernels in
flight and time by total workgroups in flight.  usage: python tools/kt_fill.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def wgs(r):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return g // max(1, w)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], wgs(r)) for r in rows)
adam = [k for k, e in enumerate(ev) if "adam" in e[2].lower()]
ends = [k for j, k in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - k > 5]
step = ev[ends[-2] + 1:ends[-1] + 1]
t0, t1 = step[0][0], max(e[1] for e in step)
pts = sorted([(s, +w, 1) for s, e, n, w in step] + [(e, -w, -1) for s, e, n, w in step])
wg = cnt = 0
last = pts[0][0]
fill = {"idle": 0, "<64 WGs": 0, "64-255": 0, ">=256": 0}
conc = {0: 0, 1: 0, 2: 0, 3: 0}
for t, dw, dc in pts:
    dt = t - last
    fill["idle" if cnt == 0 else "<64 WGs" if wg < 64 else "64-255" if wg < 256 else ">=256"] += dt
    conc[min(cnt, 3)] += dt
    wg += dw; cnt += dc; last = t
print("step wall %.2f ms; by workgroups in flight:" % ((t1 - t0) / 1e6), {k: round(v / 1e6, 2) for k, v in fill.items()},
      "; kernels in flight (0/1/2/3+):", {k: round(v / 1e6, 2) for k, v in conc.items()})
