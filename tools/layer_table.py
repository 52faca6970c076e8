#!/usr/bin/env python3
"""Per-layer dispatch table from the rocprofv3 outputs of tools/prof_layers.sh (workload: tools/layer_profile.py).

usage: layer_table.py <ops.json> <kernel_trace_dir> <pmc_fetch_dir> <pmc_write_dir> <pmc_mfma_dir> <out.json>

The dispatch stream of the workload is   marker, calibration x R, marker, ..., marker, op_i x R, marker, ...   (marker = axpy_k).
Every CSV is ordered by dispatch and cut at the LAST 2 (len(ops) + 1) markers, taken in pairs, so a row of the table is "what the product
launches for that layer and direction" - all kernels of the op (weight packing, the convolution, its slab reductions) - keyed on
dispatch order.  Durations come from the kernel trace (no counters active); FETCH_SIZE / WRITE_SIZE / MFMA-busy from their own
passes (MI355X_MICROARCH.md: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 - they cannot share a pass; gfx950 tallies 128-B read
requests at 64 B, hence the calibration factor measured on a known 1 GiB read in segment 0)."""
import csv, glob, json, re, sys, collections

meta = json.load(open(sys.argv[1]))
trace_dir, fetch_dir, write_dir, mfma_dir, out_path = sys.argv[2:7]
K, R = len(meta["ops"]), meta["reps"]


def short(n):
    return re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "").split("(")[0]


def segments(rows, name_key, order_key):
    rows = sorted(rows, key=order_key)
    marks = [i for i, r in enumerate(rows) if short(r[name_key]).startswith(meta["marker"])]
    assert len(marks) >= 2 * (K + 1), f"{len(marks)} markers, need {2 * (K + 1)}"
    marks = marks[-2 * (K + 1):]
    # segments are marker PAIRS (what runs between two pairs is the next op's group prefix); zero_k = the cache flush between
    # launches of the PMC passes (tools/layer_profile.py:cold)
    return [[r for r in rows[a + 1:b] if not short(r[name_key]).startswith("zero_k")] for a, b in zip(marks[0::2], marks[1::2])]


def trace_rows():
    f = glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True)[0]
    return segments(list(csv.DictReader(open(f))), "Kernel_Name", lambda r: int(r["Start_Timestamp"]))


def pmc_rows(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    # one row per (dispatch, counter): fold the counters of a dispatch together
    by = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        e = by.setdefault(int(r["Dispatch_Id"]), dict(Kernel_Name=r["Kernel_Name"], Dispatch_Id=int(r["Dispatch_Id"]), c={}))
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return segments(list(by.values()), "Kernel_Name", lambda r: r["Dispatch_Id"])


tr, fe, wr, mf = trace_rows(), pmc_rows(fetch_dir), pmc_rows(write_dir), pmc_rows(mfma_dir)
KB = 1024.0
cal_fetch = sum(r["c"].get("FETCH_SIZE", 0.0) for r in fe[0]) * KB / R
factor = meta["calibration"]["known_read_bytes"] / cal_fetch
table = dict(source="rocprofv3 --kernel-trace | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                    "(four separate runs of tools/layer_profile.py; tools/prof_layers.sh)",
             model=meta["model"], batch=meta["batch"], launches_per_op=R,
             fetch_size_calibration=dict(kernel=meta["calibration"]["kernel"], known_read_bytes=meta["calibration"]["known_read_bytes"],
                                         FETCH_SIZE_bytes=int(cal_fetch), factor=round(factor, 4)),
             layers=[])
for i, op in enumerate(meta["ops"]):
    seg = tr[i + 1]
    kern = collections.OrderedDict()
    for r in seg:
        d = kern.setdefault(short(r["Kernel_Name"]), [])
        d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    per_launch_us = sum(sum(v) for v in kern.values()) / R
    rd = sum(r["c"].get("FETCH_SIZE", 0.0) for r in fe[i + 1]) * KB * factor / R
    wb = sum(r["c"].get("WRITE_SIZE", 0.0) for r in wr[i + 1]) * KB / R
    dom = max(kern, key=lambda k: sum(kern[k]))
    busy = [r for r in mf[i + 1] if short(r["Kernel_Name"]) == dom]
    mb = sum(r["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for r in busy)
    ga = sum(r["c"].get("GRBM_GUI_ACTIVE", 0.0) for r in busy)
    row = dict(op)
    row.update(kernels=[dict(name=k, calls_per_launch=len(v) // R, avg_us=round(sum(v) / len(v), 2), min_us=round(min(v), 2), max_us=round(max(v), 2))
                        for k, v in kern.items()],
               dominant_kernel=dom, avg_launch_us=round(per_launch_us, 2),
               achieved_tflops=round(op["flops"] / per_launch_us / 1e6, 2),
               frac_of_roof=round(op["flops"] / per_launch_us / 1e6 / op["roof_tflops"], 4) if op["roof_tflops"] else None,
               hbm_read_bytes=int(rd), hbm_write_bytes=int(wb), hbm_bytes=int(rd + wb),
               traffic_over_algorithmic=round((rd + wb) / op["algorithmic_bytes"], 3),
               mfma_busy_fraction=round(mb / (ga / 8.0 * 1024), 3) if ga else None)
    table["layers"].append(row)
json.dump(table, open(out_path, "w"), indent=1)
for r in table["layers"]:
    print(f"{r['layer']:28s} {r['op']:16s} {r['avg_launch_us']:9.1f} us  {r['achieved_tflops']:7.1f} TF/s  frac {r['frac_of_roof']}  traffic x{r['traffic_over_algorithmic']}  "
          f"mfma-busy {r['mfma_busy_fraction']}  [{r['dominant_kernel']}]")
