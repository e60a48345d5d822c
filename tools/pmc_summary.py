"""Summarise the PMC passes of tools/gpu_r2_l.sh: HBM traffic per conv launch (profiles/r02_traffic_*.json) and whole-network MFMA
busy fraction (profiles/r02_pmc_net_*.txt).  Counter conventions per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE counts half of a wide read stream (doubled here); SQ_* / GRBM_* are summed over the 8 XCDs."""
import collections
import csv
import glob
import json
import sys

pmc_dir, variant, batch, out_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def is_conv(k):
    return "conv_igemm" in k or "patch_kernel" in k or "patch3_kernel" in k


def load(tag):
    f = glob.glob(f"{pmc_dir}/{tag}_counter_collection.csv")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    if not f:
        return per, n
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        key = "conv" if is_conv(k) else "stem" if "stem_kernel" in k else "other"
        per[r["Counter_Name"]][key] += float(r["Counter_Value"])
        n[r["Counter_Name"]][key] += 1
    return per, n


fs, fn = load("FETCH_SIZE")
ws, wn = load("WRITE_SIZE")
if fs and ws:
    launches = fn["FETCH_SIZE"]["conv"] + fn["FETCH_SIZE"]["stem"]
    rd = (fs["FETCH_SIZE"]["conv"] + fs["FETCH_SIZE"]["stem"]) * 1024 * 2
    wr = (ws["WRITE_SIZE"]["conv"] + ws["WRITE_SIZE"]["stem"]) * 1024
    d = dict(workload=f"{variant} bf16 batch {batch} @ 640x640, two batch-split lanes", conv_and_stem_launches_in_run=launches,
             read_bytes_per_launch=rd / launches, write_bytes_per_launch=wr / launches, traffic_bytes_per_launch=(rd + wr) / launches,
             method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes around `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                    "--no-accuracy --no-secondary`; counters in KiB summed over every conv + stem dispatch of the run (threshold calibration forwards included: same "
                    "kernels, same shapes) / number of those dispatches; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as is")
    json.dump(d, open(f"{out_dir}/r02_traffic_{variant[-1]}{batch}.json", "w"), indent=1)
    print(json.dumps(d, indent=1))
ms, mn = load("SQ_VALU_MFMA_BUSY_CYCLES")
qs, qn = load("SQ_WAVE_CYCLES")
if ms:
    with open(f"{out_dir}/r02_pmc_net_{variant[-1]}{batch}.txt", "w") as f:
        for key in ("conv", "stem"):
            busy, act = ms["SQ_VALU_MFMA_BUSY_CYCLES"][key], ms["GRBM_GUI_ACTIVE"][key]
            line = (f"{key}: {mn['GRBM_GUI_ACTIVE'][key]} dispatches, GRBM_GUI_ACTIVE {act:.4g} (sum over 8 XCDs), SQ_VALU_MFMA_BUSY_CYCLES {busy:.4g} -> "
                    f"MFMA busy = busy / (active / 8 * 1024 SIMDs) = {100 * busy / max(act / 8 * 1024, 1):.1f} %")
            if qs:
                line += f"; wave cycles waiting on an instruction / wave cycles {qs['SQ_WAIT_INST_ANY'][key] / max(qs['SQ_WAVE_CYCLES'][key], 1):.2f}"
            print(line)
            f.write(line + "\n")
        f.write(f"# whole-network view: every conv dispatch of `python bench.py --steps 2 --warmup 1 ...` ({variant} batch {batch}), separate --pmc passes, kernel-trace only\n")
