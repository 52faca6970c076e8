#!/usr/bin/env python3
"""Copies what tools/gpu_round3.sh left in gpurun_out/ (scratch, merged back by gpurun) into profiles/ (tracked): the rocprofv3
summaries, the per-layer table with its raw traces, the timeline, and ONE json with every bench line of the run.
usage: collect_round.py [round=3]"""
import glob, json, os, shutil, sys

R = sys.argv[1] if len(sys.argv) > 1 else "3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
copied = []
for f in sorted(glob.glob(os.path.join(src, f"r{R}_*"))):
    shutil.copy2(f, os.path.join(dst, os.path.basename(f)))
    copied.append(os.path.basename(f))


def last_json_line(path):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


names = {"phiseg": "bench_phiseg.json", "unet": "bench_unet.json", "probunet": "bench_probunet.json", "phiseg3d": "bench_phiseg3d.json",
         "phiseg3d_f32split": "bench_phiseg3d_f32split.json", "phiseg3d_f32storage": "bench_phiseg3d_f32storage.json", "phiseg3d_rev": "bench_phiseg3d_rev.json",
         "phiseg_bf16math": "bench_phiseg_bf16math.json", "2ranks_one_device": "bench_2ranks_one_device.json",
         "unet_cpu_b4": "bench_unet_cpu_b4.json", "phiseg_graph_replay": "bench_phiseg_graph_replay.json"}
lines = {k: last_json_line(os.path.join(src, v)) for k, v in names.items()}
lines = {k: v for k, v in lines.items() if v is not None}
tests = {}
for mode in ("default", "f32", "split"):
    p = os.path.join(src, f"pytest_gpu_{mode}.log")
    if os.path.exists(p):
        tail = [l for l in open(p).read().strip().splitlines() if " passed" in l or " failed" in l]
        tests[mode] = tail[-1].strip("= ") if tail else None
lines["gpu_test_tier"] = tests
json.dump(lines, open(os.path.join(dst, f"r{R}_bench_lines.json"), "w"), indent=1)
# the PMC passes of prof_round.sh as one json (tools/pmc_to_json.py reads the three summaries)
os.system(f"cd {root} && python tools/pmc_to_json.py {R} profiles/r{R}_pmc_fetch_size_summary.txt profiles/r{R}_pmc_write_size_summary.txt profiles/r{R}_pmc_mfma_busy_summary.txt > /dev/null")
if os.path.exists(os.path.join(dst, f"r{R}_pmc_b16_fetch_size_summary.txt")):
    os.system(f"cd {root} && python tools/pmc_b16_to_json.py {R} profiles/r{R}_pmc_b16_fetch_size_summary.txt profiles/r{R}_pmc_b16_write_size_summary.txt > /dev/null")
print("copied", len(copied), "files;", "bench lines:", ", ".join(f"{k}={v.get('value')}" for k, v in lines.items() if isinstance(v, dict) and "value" in v))
print("gpu tests:", tests)
