#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel instance of unet-zoo_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled:
no GPU needed) -> profiles/kernel_resources.json, keyed by source file with the sha1 of the sources it was built from.
tests/test_host_cpu.py checks that the committed table belongs to the committed sources and that the instances the step depends on stay
inside their budgets - round 5 lost 3 - 4 % of the step to a wrapper loop that doubled one instance's scratch and nobody looked.
usage: python tools/kernel_resources.py            (about three minutes)"""
import hashlib, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unet-zoo_amd", "csrc")
HEADERS = [os.path.join(CSRC, "uz_common.h"), os.path.join(CSRC, "split_f16.h"), os.path.join(ROOT, "include", "uz_api.h")]


def sha(paths):
    h = hashlib.sha1()
    for p in paths:
        h.update(open(p, "rb").read())
    return h.hexdigest()


def table(src):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-fno-gpu-rdc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur, names = {}, None, []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); names.append(cur); rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n") if names else []
    res = {}
    for mangled, nice in zip(names, dem):
        v = rows[mangled]
        nice = nice.replace("(anonymous namespace)::", "").replace("void ", "")
        nice = re.sub(r"\((SP|WS|BnP|RsP|ConvP|WgP|C1P|HP)\)$", "", nice)
        res[nice] = dict(vgpr=v.get("VGPRs"), agpr=v.get("AGPRs"), sgpr=v.get("TotalSGPRs"), scratch=v.get("ScratchSize [bytes/lane]"),
                         occupancy=v.get("Occupancy [waves/SIMD]"), lds=v.get("LDS Size [bytes/block]"))
    return res


def main():
    out = {}
    path = os.path.join(ROOT, "profiles", "kernel_resources.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip"):
            src = os.path.join(CSRC, f)
            h = sha([src] + HEADERS)
            if f in old and old[f]["sha1"] == h:             # unchanged sources: keep the entry
                out[f] = old[f]
                continue
            out[f] = dict(sha1=h, kernels=table(src))
            print(f, len(out[f]["kernels"]), "kernels", file=sys.stderr)
            for k, v in sorted(out[f]["kernels"].items()):    # what changed against the committed table
                o = old.get(f, {}).get("kernels", {}).get(k)
                if o != v:
                    print("   ", k, o and {q: o[q] for q in ("vgpr", "sgpr", "scratch")}, "->", {q: v[q] for q in ("vgpr", "sgpr", "scratch")}, file=sys.stderr)
    json.dump(out, open(os.path.join(ROOT, "profiles", "kernel_resources.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
