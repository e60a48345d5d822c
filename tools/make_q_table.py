#!/usr/bin/env python3
"""Derive a tile table in which every "p" (patch v2) choice is replaced by its cross-tile pipelined "q" twin where one exists."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "head_detector_amd", "tuning", "conv_cfg.json")
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "conv_cfg_q.json")
qnames = set(re.findall(r'QCFG\((\d+), (\d+), (\d+), (\d+), (\d+)\)', open(os.path.join(ROOT, "head_detector_amd", "csrc", "conv_igemm.hip")).read()))
q = {f"p{th}x{tw}x{bc}_n{a}x{b}": f"q{th}x{tw}x{bc}_n{a}x{b}" for tw, th, bc, a, b in qnames}
t = json.load(open(src))
n = 0
for k, v in t.items():
    if v in q:
        t[k] = q[v]
        n += 1
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(t, open(dst, "w"), indent=0, sort_keys=True)
print(f"{n} of {len(t)} entries switched to q tiles -> {dst}")
