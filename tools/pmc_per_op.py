"""Per-op HBM traffic: FETCH_SIZE / WRITE_SIZE of each network dispatch (single-lane PMC passes of tools/gpu_run.sh pmc) next to the op's algorithmic bytes.
Dispatch order inside a forward is the op order of the program (one launch per op, one lane).  Conventions as tools/pmc_summary.py (KiB; FETCH doubled on gfx950).

    python tools/pmc_per_op.py PMC_DIR VARIANT BATCH FORWARDS [OUT.txt]
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from head_detector_amd import arch  # noqa: E402

pmc_dir, variant, batch, forwards = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
out = open(sys.argv[5], "w") if len(sys.argv) > 5 else sys.stdout


def is_net(k):
    from head_detector_amd import arch

    return arch.is_net_kernel(k)


def load(name):
    f = sorted(glob.glob(f"{pmc_dir}/**/{name}_counter_collection.csv", recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and is_net(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


P = arch.build_program(variant, arch.random_state_dict(variant, 1), 640)
ops = [op for op in P.ops]
fr, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
per = len(fr) // forwards
pairs = arch.b2b_pairs(P)
stem_in = per == len(ops) - len(pairs) - 1 and pairs and pairs[0] == 1 and ops[0]["kind"] == 0  # u8 images: the stem conv ran inside the stage-1 pair's launch too
fused = (per == len(ops) - len(pairs) or stem_in) and pairs  # the back-to-back pairs ran as one launch each (the engine's default): one row per launch
if fused:
    merged = []
    for i, op in enumerate(ops):
        if i - 1 in pairs or (stem_in and i == 0):
            continue
        if i in pairs:
            m = dict(op, name=op["name"] + " + " + ops[i + 1]["name"].split(".")[-1], _second=ops[i + 1])
            if stem_in and i == 1:
                m = dict(m, name="stem + " + m["name"], _stem=ops[0])
            merged.append(m)
        else:
            merged.append(op)
    ops = merged
assert per == len(ops) and len(wr) == len(fr), (per, len(ops), len(wr))
tot = [0.0] * 4
print(f"# {variant} batch {batch}: HBM bytes per op, mean of {forwards} forwards (MB); excess = measured - algorithmic", file=out)
print(f"{'op':44s} {'alg_rd':>8s} {'rd':>8s} {'alg_wr':>8s} {'wr':>8s} {'excess':>8s}  kernel", file=out)
table = []
for i, op in enumerate(ops):
    a = arch.op_algorithmic_bytes(P, op, batch)
    if "_second" in op:  # a fused pair: the first conv's input + weights in, the second conv's weights in and output out; the tensor between them does not exist
        a2 = arch.op_algorithmic_bytes(P, op["_second"], batch)
        ib2 = P.bufs[op["_second"]["in_buf"]]
        a = dict(read=a["read"] + a2["read"] - batch * ib2["h"] * ib2["w"] * op["_second"]["cin"] * 2, write=a2["write"])
        if "_stem" in op:  # the stem's tensor does not exist either: the image in, the stem's weights, the pair's weights
            a0 = arch.op_algorithmic_bytes(P, op["_stem"], batch)
            ib1 = P.bufs[op["in_buf"]]
            a["read"] += a0["read"] - batch * ib1["h"] * ib1["w"] * min(op["cin"], ib1["pitch"] - op["in_coff"]) * 2
    rd = sum(float(fr[f * per + i]["Counter_Value"]) for f in range(forwards)) * 2048 / forwards
    w = sum(float(wr[f * per + i]["Counter_Value"]) for f in range(forwards)) * 1024 / forwards
    k = fr[i]["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = k[:k.find("(")] if "(" in k else k
    table.append((op.get("name", str(i)), a["read"], rd, a["write"], w, k))
    for j, v in enumerate((a["read"], rd, a["write"], w)):
        tot[j] += v
for name, ar, rd, aw, w, k in table:
    print(f"{name:44s} {ar / 1e6:8.1f} {rd / 1e6:8.1f} {aw / 1e6:8.1f} {w / 1e6:8.1f} {(rd + w - ar - aw) / 1e6:8.1f}  {k[-60:]}", file=out)
print(f"{'total':44s} {tot[0] / 1e6:8.1f} {tot[1] / 1e6:8.1f} {tot[2] / 1e6:8.1f} {tot[3] / 1e6:8.1f} {(tot[1] + tot[3] - tot[0] - tot[2]) / 1e6:8.1f}", file=out)
