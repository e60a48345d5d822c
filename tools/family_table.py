#!/usr/bin/env python3
"""Which kernel family runs which ops, at what rate, with what traffic ratio -- generated, not typed:

    python tools/family_table.py profiles/r05_per_layer_l64.json profiles/r05_traffic_per_op_l64.txt [--md]

per_layer.json = bench.py --per-layer (single-stream HIP-event time, algorithmic FLOPs and bytes per op); traffic_per_op.txt = tools/pmc_per_op.py (PMC
FETCH_SIZE / WRITE_SIZE per dispatch + the kernel that ran the op).  Rows: kernel family x bound class (an op is MFMA-bound when its algorithmic
intensity is above the 2.5 PFLOP/s / 8 TB/s ridge of 310 FLOP per byte)."""
import json
import re
import sys

FAMILIES = [
    (r"ds_b2b_kernel<\d, 1>", "u  stem + stage-1 downsample + conv1|conv2 in one persistent launch (r06)"),
    (r"ds_b2b_kernel", "t  stage-1 downsample + conv1|conv2, persistent, weights in registers (r06)"),
    (r"ds_conv_kernel", "r  3x3 stride-2 from 96 channels, persistent, weights in registers (r06)"),
    (r"w_conv_kernel", "w  3x3 stride-1 from 96 / 128 channels, weights in registers (r06)"),
    (r"conv3x3_pp_kernel<(\d), 1, 0, 0, 0, 0, 1>", "s  ping-pong 3x3, two 4x8 sub-patches per wave (r06)"),
    (r"conv_igemm_kernel<[^>]*, [468], 1>$", "b2b  implicit GEMM + the 1x1 conv behind it in one launch (r06)"),
    (r"conv3x3_pp_kernel<(\d), 1", "g  ping-pong 3x3 (two barriers per tap)"),
    (r"conv3x3_pp_kernel<(\d), 2", "h  ping-pong 3x3 (one barrier per tap)"),
    (r"conv3x3_patch3_kernel", "q  halo-patch 3x3, cross-tile pipelined"),
    (r"conv3x3_patch_kernel<.*, 1>$", "d  halo-patch 3x3 stride 2 (parity planes)"),
    (r"conv3x3_patch_kernel", "p  halo-patch 3x3"),
    (r"conv1x1_stream_kernel", "t  streaming 1x1"),
    (r"conv_igemm_kernel", "implicit GEMM (1x1, stride-2 3x3, 20^2 3x3, ConvT, predictions)"),
    (r"stem_ds_kernel", "fused stem + downsample"),
    (r"stem_", "stem"),
    (r"spp_pool", "SPP max-pool"),
]
RIDGE = 2500e12 / 8e12  # FLOP per byte


def family(kernel: str) -> str:
    for pat, name in FAMILIES:
        if re.search(pat, kernel):
            return name
    return kernel[:40]


def main():
    per_layer = json.load(open(sys.argv[1]))
    traffic = {}
    for ln in open(sys.argv[2]):
        if ln.startswith("#") or ln.startswith("op ") or ln.startswith("total"):
            continue
        m = re.match(r"((?:stem \+ )?\S+(?: \+ \S+)?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)\s+(.*)$", ln.rstrip())
        if m:
            traffic[m.group(1)] = dict(alg_rd=float(m.group(2)), rd=float(m.group(3)), alg_wr=float(m.group(4)), wr=float(m.group(5)), kernel=m.group(7).strip())
    rows = {}
    # a back-to-back pair is ONE launch: its traffic row is named "<first> + <second's last name part>"; the per-layer table lists the two ops (the second at ~0 ms)
    fused = {k.split(" + ")[0]: k for k in traffic if " + " in k}
    merged, skip = [], False
    stem_row = next((k for k in traffic if k.startswith("stem + ")), None)  # u8 images: the stem conv ran inside the stage-1 pair's launch: one row for the three ops
    if stem_row and len(per_layer) >= 3 and per_layer[0]["kind"] == 0:
        a, b, c = per_layer[0], per_layer[1], per_layer[2]
        merged.append(dict(a, name=stem_row, ms=a["ms"] + b["ms"] + c["ms"], gflop=a["gflop"] + b["gflop"] + c["gflop"], read_mb=traffic[stem_row]["alg_rd"], write_mb=c["write_mb"]))
        per_layer = per_layer[3:]
    for i, op in enumerate(per_layer):
        if skip:
            skip = False
            continue
        if op["name"] in fused and i + 1 < len(per_layer):
            nx = per_layer[i + 1]
            op = dict(op, name=fused[op["name"]], ms=op["ms"] + nx["ms"], gflop=op["gflop"] + nx["gflop"], read_mb=op["read_mb"] + nx["read_mb"] - op["write_mb"], write_mb=nx["write_mb"])
            skip = True
        merged.append(op)
    for op in merged:
        t = traffic.get(op["name"])
        if t is None:
            continue
        by = (op["read_mb"] + op["write_mb"]) * 1e6
        bound = "MFMA" if by > 0 and op["gflop"] * 1e9 / by > RIDGE else "HBM"
        r = rows.setdefault((family(t["kernel"]), bound), dict(n=0, ms=0.0, gflop=0.0, alg=0.0, meas=0.0, alg_rd=0.0, rd=0.0, ops=[]))
        r["n"] += 1
        r["ms"] += op["ms"]
        r["gflop"] += op["gflop"]
        r["alg"] += t["alg_rd"] + t["alg_wr"]
        r["meas"] += t["rd"] + t["wr"]
        r["alg_rd"] += t["alg_rd"]
        r["rd"] += t["rd"]
        r["ops"].append((op["ms"], op["name"]))
    tot_ms = sum(r["ms"] for r in rows.values())
    md = "--md" in sys.argv
    hdr = ["kernel family", "bound", "ops", "ms (single stream)", "share", "TFLOP/s", "alg. GB/s", "HBM bytes / alg.", "reads / alg. reads", "largest ops"]
    if md:
        print("| " + " | ".join(hdr) + " |")
        print("|" + "---|" * len(hdr))
    else:
        print("  ".join(hdr))
    for (fam, bound), r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        big = ", ".join(f"`{n}` {ms * 1e3:.0f} us" for ms, n in sorted(r["ops"], reverse=True)[:2])
        cells = [fam, bound, str(r["n"]), f"{r['ms']:.3f}", f"{100 * r['ms'] / tot_ms:.1f} %", f"{r['gflop'] / r['ms']:.0f}" if r["gflop"] else "-",
                 f"{r['alg'] / r['ms']:.0f}", f"{r['meas'] / r['alg']:.2f}" if r["alg"] else "-", f"{r['rd'] / r['alg_rd']:.2f}" if r["alg_rd"] else "-", big]
        print(("| " + " | ".join(cells) + " |") if md else "  ".join(cells))
    tf = sum(r["gflop"] for r in rows.values()) / tot_ms
    print(f"\ntotal: {sum(r['n'] for r in rows.values())} ops, {tot_ms:.3f} ms single stream = {tf:.0f} TFLOP/s; measured HBM bytes {sum(r['meas'] for r in rows.values()) / 1e3:.2f} GB "
          f"= {sum(r['meas'] for r in rows.values()) / sum(r['alg'] for r in rows.values()):.3f} x algorithmic")


if __name__ == "__main__":
    main()
