#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_regs.py head_detector_amd/csrc/conv_split.hip [extra hipcc flags]"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-unused-value"]


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
        for k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "LDS Size", "SGPRs"):
            m = re.search(k + r"[^:]*: (\d+)", ln)
            if m and cur is not None and k not in cur:
                cur[k] = int(m.group(1))
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        sys.exit(r.returncode)
    names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
    for x, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
        n = re.sub(r"\(ConvArgs.*", "", n)
        print(f"{n[:80]:80s} vgpr {x.get('VGPRs'):4d} agpr {x.get('AGPRs'):4d} scratch {x.get('ScratchSize'):5d} occ {x.get('Occupancy')}")


if __name__ == "__main__":
    main()
