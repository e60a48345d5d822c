#!/usr/bin/env python3
"""MFMA-busy of the FLAME vertex kernels from `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` around
`FLAME_NS=8192 python tools/flame_sweep.py` (tools/gpu_run.sh pmcflame):  python tools/pmc_flame_summary.py PMC_DIR OUT.txt"""
import collections
import csv
import glob
import re
import sys

pmc_dir, out_path = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(pmc_dir + "/**/f_counter_collection.csv", recursive=True))[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"(flame_\w+(<[^>]*>)?)", r["Kernel_Name"])
    per[m.group(1) if m else "other"][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out_path, "w") as out:
    out.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE around `FLAME_NS=8192 python tools/flame_sweep.py` (three live-coefficient settings)\n")
    out.write("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); v_mfma_f32_32x32x2_f32 = 64 busy cycles\n")
    for k, v in per.items():
        if "flame" not in k:
            continue
        fr = [x / (y / 8 * 1024) for x, y in zip(v["SQ_VALU_MFMA_BUSY_CYCLES"], v["GRBM_GUI_ACTIVE"]) if y > 0]
        if not fr:
            continue
        line = "%-42s %4d dispatches  MFMA busy min %.1f %%  median %.1f %%  max %.1f %%" % (k, len(fr), 100 * min(fr), 100 * sorted(fr)[len(fr) // 2], 100 * max(fr))
        print(line)
        out.write(line + "\n")
