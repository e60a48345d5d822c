#!/usr/bin/env python3
"""Measure every conv tile configuration on every distinct conv shape of a network (per-op HIP events through
vgh_net_profile) and write the winners to head_detector_amd/tuning/conv_cfg.json (merged with what is there).
Run on the GPU box:  python tools/tune_conv.py --variant vgg_heads_m --batch 32 [--reps 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from head_detector_amd.engine import TUNING_DIR, VGHeadsEngine, tuning_key  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="vgg_heads_m")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--image-size", type=int, default=640)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--split", type=int, default=1, help="measure with the batch split over this many lane streams (vgh_net_set_split)")
    ap.add_argument("--out", default=os.path.join(TUNING_DIR, "conv_cfg.json"))
    ap.add_argument("--report", default=None)
    ap.add_argument("--only", default=None, help="measure only tiles whose name starts with this prefix (e.g. t: the streaming 1x1 tiles), next to the current table")
    ap.add_argument("--no-write", action="store_true", help="report only, leave the table alone")
    ap.add_argument("--precision", default="bf16", help="bf16 (throughput tiles) or fp16x3 / bf16x3 (the split-precision tile set; keys get a precision prefix)")
    args = ap.parse_args()
    eng = VGHeadsEngine(args.variant, image_size=args.image_size, max_batch=args.batch, use_tuning=False, precision=args.precision)
    if args.split > 1:
        eng.set_split(args.split)
    names = eng.cfg_names()
    x = torch.randint(0, 256, (args.batch, args.image_size, args.image_size, 3), dtype=torch.uint8).cuda()
    ops = eng.program.ops
    conv_idx = [i for i, op in enumerate(ops) if op["kind"] == 1]
    best = {}
    times = {i: {} for i in conv_idx}
    if args.only:  # the current table's choice as the reference row of every op
        eng.load_tuning()
        runs = [eng.profile_ops(x) for _ in range(args.reps + 1)][1:]
        for i in conv_idx:
            times[i]["<table>"] = min(r[i]["ms"] for r in runs)
    for c, name in enumerate(names):
        if args.only and not name.startswith(args.only):
            continue
        ok = [i for i in conv_idx if eng.cfg_ok(c, ops[i])]
        if not ok:
            continue
        for i in conv_idx:
            eng.set_cfg(i, c if i in ok else -1)
        runs = [eng.profile_ops(x) for _ in range(args.reps + 1)][1:]
        for i in ok:
            times[i][name] = min(r[i]["ms"] for r in runs)
    table, report = {}, []
    # the winner of a table key is the tile with the smallest SUM over the ops that share the key (r04: it used to be the tile of the single fastest op)
    sums = {}
    for i in conv_idx:
        key = ("" if args.precision == "bf16" else args.precision + ":") + tuning_key(ops[i], args.batch, args.split)
        acc = sums.setdefault(key, {})
        for name, t in times[i].items():
            acc.setdefault(name, []).append(t)
    for key, acc in sums.items():
        n = max(len(v) for v in acc.values())
        full = {name: sum(v) for name, v in acc.items() if len(v) == n}
        w = min(full, key=full.get)
        best[key] = (w, full[w])
    for i in conv_idx:
        key = ("" if args.precision == "bf16" else args.precision + ":") + tuning_key(ops[i], args.batch, args.split)
        w = min(times[i], key=times[i].get)
        fl = 2.0 * ops[i]["macs"] * args.batch
        report.append(dict(name=ops[i]["name"], key=key, best=w, ms=times[i][w], tflops=fl / times[i][w] / 1e9, all={k: round(v, 4) for k, v in sorted(times[i].items(), key=lambda kv: kv[1])}))
    table = {k: v[0] for k, v in best.items() if v[0] != "<table>"}
    if not args.no_write:
        old = json.load(open(args.out)) if os.path.exists(args.out) else {}
        old.update(table)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(old, open(args.out, "w"), indent=0, sort_keys=True)
    if args.only:
        gain = 0.0
        for r in report:
            if r["best"] != "<table>":
                gain += r["all"]["<table>"] - r["ms"]
                print(f"  {r['name']:44s} table {r['all']['<table>'] * 1e3:7.1f} us -> {r['best']:28s} {r['ms'] * 1e3:7.1f} us")
        print(f"  single-stream gain over the table: {gain * 1e3:.1f} us")
    tot = sum(r["ms"] for r in report)
    print(f"{args.variant} B={args.batch} split={args.split}: sum of best conv times {tot:.3f} ms -> {eng.flops_per_image * args.batch / tot / 1e9:.1f} TFLOP/s over convs; {len(table)} shapes")
    if args.report:
        json.dump(report, open(args.report, "w"), indent=0)
    eng.close()


if __name__ == "__main__":
    main()
