"""Concurrency analysis of a rocprofv3 --kernel-trace CSV: for the last training step, the sum of
kernel durations, the union of busy time, and time with >=2 kernels in flight.
usage: python tools/kt_overlap.py gpurun_out/kt2/*/*_kernel_trace.csv"""
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
    adam = [k for k, e in enumerate(ev) if "adam" in e[2].lower()]
    # a step ends with the last adam launch of a run of adam launches
    ends = [k for j, k in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - k > 5]
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = ev[lo:hi]
    t0, t1 = step[0][0], max(e[1] for e in step)
    pts = sorted([(s, 1) for s, _, _ in step] + [(e, -1) for _, e, _ in step])
    busy = multi = 0
    depth, last = 0, pts[0][0]
    for t, d in pts:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    tot = sum(e - s for s, e, _ in step)
    print(f"{path}: kernels {len(step)}  wall {(t1 - t0) / 1e6:.2f} ms  sum {tot / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  "
          f"idle {((t1 - t0) - busy) / 1e6:.2f} ms  >=2 in flight {multi / 1e6:.2f} ms")
    fam = {}
    for s, e, n in step:
        key = n.split("(")[0].split("<")[0].replace("void ", "").replace("uz::", "")
        f = fam.setdefault(key, [0, 0])
        f[0] += 1
        f[1] += e - s
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {k:34s} {c:5d} {t / 1e6:8.3f} ms")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
