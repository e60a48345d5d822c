"""Summarise PMC passes taken around `tools/traffic_run.py --forwards N --split L` (tools/gpu_run.sh pmc): HBM traffic PER FORWARD and
whole-network MFMA busy.  Counter conventions per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts half of a wide read stream (doubled here, its HBM section); WRITE_SIZE is taken as is (uncalibrated); SQ_* / GRBM_* are
summed over the 8 XCDs.  Every dispatch of the run belongs to one of the N forwards (the engine's stream probes are spin kernels, listed
under "other"), so sum / N is the per-forward figure -- no calibration forwards to subtract.

    python tools/pmc_summary.py PMC_DIR VARIANT BATCH FORWARDS SPLIT OUT_DIR [TAG]
"""
import collections
import csv
import glob
import json
import os
import sys

pmc_dir, variant, batch, forwards, split, out_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
tag = sys.argv[7] if len(sys.argv) > 7 else "r03"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def is_net(k):
    from head_detector_amd import arch

    return arch.is_net_kernel(k)


def load(name):
    f = sorted(glob.glob(f"{pmc_dir}/**/{name}_counter_collection.csv", recursive=True))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    if not f:
        return per, n
    for r in csv.DictReader(open(f[0])):
        key = "net" if is_net(r["Kernel_Name"]) else "other"
        per[r["Counter_Name"]][key] += float(r["Counter_Value"])
        n[r["Counter_Name"]][key] += 1
    return per, n


fs, fn = load("FETCH_SIZE")
ws, wn = load("WRITE_SIZE")
suffix = f"{variant[-1]}{batch}_x{split}"
if fs and ws:
    from head_detector_amd import arch

    P = arch.build_program(variant, arch.random_state_dict(variant, 1), 640)
    launches = fn["FETCH_SIZE"]["net"] / forwards
    n_ops = sum(1 for op in P.ops if op["kind"] in (0, 1, 2))
    stem_in = round(launches / split) == n_ops - len(arch.b2b_pairs(P)) - 1  # u8 images: the stem conv ran inside the stage-1 pair's launch (one dispatch fewer): its tensor does not exist
    alg = arch.program_algorithmic_bytes(P, batch, fused_stem=stem_in)
    rd = fs["FETCH_SIZE"]["net"] * 1024 * 2 / forwards
    wr = ws["WRITE_SIZE"]["net"] * 1024 / forwards
    d = dict(workload=f"{variant} bf16 batch {batch} @ 640x640, {split} batch-split lane(s)", forwards_in_run=forwards, launches_per_forward=launches,
             read_bytes_per_forward=rd, write_bytes_per_forward=wr, traffic_bytes_per_forward=rd + wr, traffic_bytes_per_launch=(rd + wr) / launches,
             algorithmic_read_bytes_per_forward=alg["read"], algorithmic_write_bytes_per_forward=alg["write"], algorithmic_bytes_per_forward=alg["read"] + alg["write"],
             traffic_over_algorithmic=(rd + wr) / (alg["read"] + alg["write"]),
             method=f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes around `python tools/traffic_run.py --forwards {forwards} --split {split}` "
                    "(network forwards only: no calibration, no post-network stages); counters in KiB summed over every stem / conv / pool dispatch / forwards; "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as is; algorithmic = every op's input view, residual and weights read once + its stored "
                    "channels written once (head_detector_amd/arch.py::op_algorithmic_bytes)")
    json.dump(d, open(f"{out_dir}/{tag}_traffic_{suffix}.json", "w"), indent=1)
    print(json.dumps(d, indent=1))
ms, mn = load("SQ_VALU_MFMA_BUSY_CYCLES")
qs, qn = load("SQ_WAVE_CYCLES")
if ms:
    with open(f"{out_dir}/{tag}_pmc_net_{suffix}.txt", "w") as f:
        busy, act = ms["SQ_VALU_MFMA_BUSY_CYCLES"]["net"], ms["GRBM_GUI_ACTIVE"]["net"]
        line = (f"net: {mn['GRBM_GUI_ACTIVE']['net']} dispatches in {forwards} forwards, GRBM_GUI_ACTIVE {act:.4g} (sum over 8 XCDs), SQ_VALU_MFMA_BUSY_CYCLES {busy:.4g} -> "
                f"MFMA busy per dispatch cycle = busy / (active / 8 * 1024 SIMDs) = {100 * busy / max(act / 8 * 1024, 1):.1f} %; MFMA-busy cycles per forward {busy / forwards:.4g}")
        if qs:
            line += f"; wave cycles waiting on an instruction / wave cycles {qs['SQ_WAIT_INST_ANY']['net'] / max(qs['SQ_WAVE_CYCLES']['net'], 1):.2f}"
        print(line)
        f.write(line + "\n")
        f.write(f"# every stem / conv / pool dispatch of `python tools/traffic_run.py --forwards {forwards} --split {split}` ({variant} batch {batch}), separate --pmc passes, kernel-trace only.\n"
                "# with 2 lanes the kernels of the two half-batches overlap in time, so the per-dispatch denominator counts wall time twice: divide MFMA-busy cycles per forward by\n"
                "# 1024 SIMDs x the forward's wall time x the shader clock for the wall-clock figure (DESIGN.md 5).\n")
