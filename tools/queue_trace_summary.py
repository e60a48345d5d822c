#!/usr/bin/env python3
"""Which hardware queue does each role's stream sit on, engine by engine?  (r06)

    cd /tmp && SEQ_MARK=1 SEQ_NF=12 rocprofv3 --kernel-trace --output-format csv -d DIR -o q -- python tools/engine_sequence_probe.py > seq.log
    python tools/queue_trace_summary.py DIR seq.log [OUT.txt]

tools/engine_sequence_probe.py (SEQ_MARK) launches one spin kernel of a role-specific length on every stream of every engine (main 211 us, lane 223 us, side 239 us):
the kernel trace carries Queue_Id (and Stream_Id where this rocprofv3 writes it), so the marks give stream -> queue per engine, and the kernels between two mark
groups -- that engine's forwards -- give the queues its network kernels and its post-network kernels ACTUALLY ran on, next to the engine's measured ms per forward."""
import csv
import glob
import os
import re
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from head_detector_amd import arch  # noqa: E402

d, log = sys.argv[1], sys.argv[2]
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
has_stream = "Stream_Id" in rows[0]
seq = [l.strip() for l in open(log) if l.startswith("SEQ ")]
ROLES = {211: "main", 223: "lane", 239: "side"}


def role_of(r):
    if "spin_kernel" not in r["Kernel_Name"] or "grid" in r["Kernel_Name"]:
        return None
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in ROLES.items():
        if abs(us - k) < 4.0:
            return v
    return None


groups, cur, last_i = [], None, -10**9
for i, r in enumerate(rows):
    ro = role_of(r)
    if ro is None:
        continue
    if ro == "main":  # a new engine's mark group starts with its main stream
        cur = {"at": i, "marks": {}}
        groups.append(cur)
    if cur is not None:
        cur["marks"][ro] = (r["Queue_Id"], r.get("Stream_Id", "-"))
print(f"# {f}: {len(rows)} dispatches, {len(groups) // 2} engines marked, {len(seq)} SEQ lines; Stream_Id column: {has_stream}", file=out)
print("# per engine: role -> Queue_Id[/Stream_Id] from the marks; then the Queue_Id histogram of the network kernels and of the post-network kernels of the engine's forwards", file=out)
print("# (an engine is marked twice, after its warm-up and before it is closed: the histograms cover exactly its timed loops and, for 8-bit engines, the variations)", file=out)
assert len(groups) % 2 == 0, "marks come in pairs: before and after an engine's loops"
pairs = [(groups[i], groups[i + 1]) for i in range(0, len(groups), 2)]
for k, (g, g_end) in enumerate(pairs):
    win = rows[g["at"]:g_end["at"]]
    netq, postq = Counter(), Counter()
    for r in win:
        if "spin" in r["Kernel_Name"]:
            continue
        key = r["Queue_Id"] + ("/" + r["Stream_Id"] if has_stream else "")
        (netq if arch.is_net_kernel(r["Kernel_Name"]) else postq)[key] += 1
    marks = "  ".join(f"{ro}={q}" + (f"/{s}" if has_stream else "") for ro, (q, s) in g["marks"].items())
    if g_end["marks"] != g["marks"]:
        marks += "   | AFTER the loops: " + "  ".join(f"{ro}={q}" + (f"/{s}" if has_stream else "") for ro, (q, s) in g_end["marks"].items())
    line = seq[k] if k < len(seq) else "(no SEQ line)"
    m = re.search(r"SEQ (\S+ \S+ \S+ \S+)\s*:\s*([\d.]+) ms per forward, network part\s*([\d.]+); again\s*([\d.]+)", line)
    head = f"{m.group(1):34s} {m.group(2):>8s} ms (net {m.group(3)}, again {m.group(4)})" if m else line[:80]
    print(f"engine {k:2d}  {head}\n    marks: {marks}\n    network kernels by queue: {dict(netq.most_common())}\n    post-network kernels by queue: {dict(postq.most_common())}", file=out)
