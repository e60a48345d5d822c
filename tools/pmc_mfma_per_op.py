"""Per-op MFMA-busy: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) of each network dispatch (single lane, one --pmc pass of
tools/gpu_run.sh pmc around tools/traffic_run.py), grouped by backbone / neck / heads.  north_star asks ">= 60 % MFMA utilisation on the backbone convs": this is
the table that claim is read from (r06).  Conventions as tools/pmc_summary.py (counters summed over the 8 XCDs; dispatch order = op order).

    python tools/pmc_mfma_per_op.py PMC_DIR VARIANT BATCH FORWARDS [OUT.txt]
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from head_detector_amd import arch  # noqa: E402

pmc_dir, variant, batch, forwards = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
out = open(sys.argv[5], "w") if len(sys.argv) > 5 else sys.stdout
f = sorted(glob.glob(f"{pmc_dir}/**/SQ_VALU_MFMA_BUSY_CYCLES_counter_collection.csv", recursive=True))[0]
by = {}
for r in csv.DictReader(open(f)):
    if arch.is_net_kernel(r["Kernel_Name"]):
        by.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
disp = [by[k] for k in sorted(by)]
P = arch.build_program(variant, arch.random_state_dict(variant, 1), 640)
ops = list(P.ops)
per = len(disp) // forwards
pairs = arch.b2b_pairs(P)
stem_in = per == len(ops) - len(pairs) - 1 and pairs and pairs[0] == 1 and ops[0]["kind"] == 0  # u8 images: the stem conv ran inside the stage-1 pair's launch too
if (per == len(ops) - len(pairs) or stem_in) and pairs:  # the back-to-back pairs ran as one launch each (the engine's default): one row per launch, both convs' FLOPs
    merged = []
    for i, op in enumerate(ops):
        if i - 1 in pairs or (stem_in and i == 0):
            continue
        if i in pairs:
            m = dict(op, name=op["name"] + " + " + ops[i + 1]["name"].split(".")[-1], macs=op["macs"] + ops[i + 1]["macs"])
            if stem_in and i == 1:
                m = dict(m, name="stem + " + m["name"], macs=m["macs"] + ops[0].get("macs", 0.0))
            merged.append(m)
        else:
            merged.append(op)
    ops = merged
assert per == len(ops), (per, len(ops))
print(f"# {variant} batch {batch}, one lane, mean of {forwards} forwards: MFMA-busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) per dispatch", file=out)
print(f"# ideal = the op's algorithmic MFMA cycles (FLOPs / 32768 per v_mfma_f32_32x32x16_bf16 * 32 cycles) over the same denominator: busy above ideal = padded or re-computed work", file=out)
print(f"{'op':46s} {'busy %':>7s} {'ideal %':>8s} {'Mcycles':>9s}  kernel", file=out)
grp = {}
for i, op in enumerate(ops):
    busy = sum(disp[fw * per + i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for fw in range(forwards)) / forwards
    act = sum(disp[fw * per + i].get("GRBM_GUI_ACTIVE", 0.0) for fw in range(forwards)) / forwards
    den = act / 8 * 1024
    name = op.get("name", str(i))
    flops = 2.0 * op.get("macs", 0.0) * batch
    ideal = flops / 32768 * 32
    k = disp[i]["k"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = k[:k.find("(")] if "(" in k else k
    print(f"{name:46s} {100 * busy / max(den, 1):7.1f} {100 * ideal / max(den, 1):8.1f} {act / 8 / 1e6:9.3f}  {k[-56:]}", file=out)
    g = name.split(".")[0]
    a = grp.setdefault(g, [0.0, 0.0, 0.0])
    a[0] += busy
    a[1] += den
    a[2] += ideal
print("#", file=out)
tb = td = 0.0
for g, (b, d, idl) in grp.items():
    print(f"# {g:10s}: MFMA-busy {100 * b / max(d, 1):5.1f} % of its dispatch cycles (ideal {100 * idl / max(d, 1):5.1f} %), {d / 1024 / 1e6:8.2f} Mcycles", file=out)
    tb += b
    td += d
print(f"# whole network: {100 * tb / max(td, 1):.1f} %", file=out)
