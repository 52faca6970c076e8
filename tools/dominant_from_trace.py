#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `bench.py` (eager run): per kernel name and launch geometry, the number of dispatches
and their mean / min / max duration - so that the launches of ONE layer (e.g. the dominant 224 -> 128 @ 32x128x128 weight
gradient) can be read off next to the live HIP-event figure of bench.py.  usage: dominant_from_trace.py <trace dir> <out.json>"""
import csv, glob, json, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not any(k in name for k in ("conv_split_kernel", "conv_splitp_kernel", "conv_split_bn_kernel", "conv_splitp_bn_kernel", "wgrad_split_kernel", "conv_mfma_kernel", "wgrad_reduce")):
            continue
        grid = (int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0))))
        rows[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for (name, grid), d in rows.items():
    out.append(dict(kernel=name, grid_threads_x=grid[0], workgroup_x=grid[1], dispatches=len(d), mean_us=round(sum(d) / len(d), 1),
                    min_us=round(min(d), 1), max_us=round(max(d), 1)))
out.sort(key=lambda e: -e["mean_us"])
json.dump(dict(source="rocprofv3 --kernel-trace -- python bench.py --steps 10 --warmup 5 --no-graphs (tools/prof_round.sh); durations in microseconds",
               note="the heaviest entry of wgrad_split_kernel<32, 64, 2, ...> / conv_split[p]_kernel_2_512_32 (p = input in split storage) is the 224 -> 128 @ 32x128x128 layer "
                    "(weight gradient: grid 256 workgroups x 512 threads; forward: 2048 workgroups; data gradient: 4096)",
               launches=out[:24]), open(sys.argv[2], "w"), indent=1)
print(json.dumps(out[:6], indent=1))
