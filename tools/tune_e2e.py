#!/usr/bin/env python3
"""Second tuning stage: coordinate descent on the WHOLE forward in the mode the bench runs (batch split over lane streams).
The per-op tuner (tools/tune_conv.py) times every op alone on an idle GPU; with two half-batches in flight the best tile of a
layer can differ (smaller tiles fill the gaps of the other lane, big tiles fight for LDS).  For every distinct conv shape,
heaviest first, the top-K tiles of the per-op report are tried in the real forward and kept when the forward gets faster.
Winners are stored under the split-mode keys ("b32x2_...") of head_detector_amd/tuning/conv_cfg.json."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd.engine import TUNING_DIR, VGHeadsEngine, tuning_key, tuning_lookup  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="vgg_heads_m")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--split", type=int, default=2)
    ap.add_argument("--report", default=os.path.join(ROOT, "profiles", "r01_tune_m32.json"), help="per-op report of tools/tune_conv.py for the same variant / batch")
    ap.add_argument("--topk", type=int, default=4)
    ap.add_argument("--min-gain", type=float, default=0.002)
    ap.add_argument("--out", default=os.path.join(TUNING_DIR, "conv_cfg.json"))
    ap.add_argument("--log", default=None)
    ap.add_argument("--image-size", type=int, default=640)
    ap.add_argument("--precision", default="bf16", help="bf16, or fp16x3 / bf16x3 (the parity modes' own tile table; keys get the precision prefix)")
    args = ap.parse_args()
    eng = VGHeadsEngine(args.variant, image_size=args.image_size, max_batch=args.batch, seed=1, precision=args.precision)
    pre = "" if args.precision == "bf16" else args.precision + ":"
    eng.set_split(args.split)
    names = {n: i for i, n in enumerate(eng.cfg_names())}
    x = torch.randint(0, 256, (args.batch, args.image_size, args.image_size, 3), dtype=torch.uint8).cuda()
    ops = eng.program.ops
    report = {r["name"]: r for r in json.load(open(args.report))}
    groups = {}
    for i, op in enumerate(ops):
        if op["kind"] != 1 or op["name"] not in report:
            continue
        g = groups.setdefault(tuning_key(op, args.batch), dict(ops=[], ms=0.0, all=report[op["name"]]["all"], best=report[op["name"]]["best"]))
        g["ops"].append(i)
        g["ms"] += report[op["name"]]["ms"]

    def measure():
        best = 1e9
        for _ in range(3):
            for _ in range(4):  # back to the steady clock / power state before timing
                eng.forward_net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(16):
                eng.forward_net(x)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 16 * 1e3)
        return best

    table = json.load(open(args.out)) if os.path.exists(args.out) else {}
    cur = {}
    for key, g in groups.items():
        cur[key] = tuning_lookup(table, ops[g["ops"][0]], args.batch, args.split, pre) or g["best"]
        for i in g["ops"]:
            eng.set_cfg(i, names[cur[key]])
    base = measure()
    log = [dict(step="baseline", ms=base)]
    print(f"baseline forward (split {args.split}): {base:.3f} ms")
    for key, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
        cands = [c for c in list(g["all"])[: args.topk + 1] if c != cur[key] and c in names][: args.topk]
        best_c, best_ms = cur[key], base
        for c in cands:
            for i in g["ops"]:
                eng.set_cfg(i, names[c])
            ms = measure()
            if ms < best_ms * (1 - args.min_gain):
                best_c, best_ms = c, ms
        for i in g["ops"]:
            eng.set_cfg(i, names[best_c])
        if best_c != cur[key]:
            check = measure()  # confirm against noise before accepting
            if check < base * (1 - args.min_gain / 2):
                print(f"{key}: {cur[key]} -> {best_c}: {base:.3f} -> {check:.3f} ms")
                log.append(dict(step=key, old=cur[key], new=best_c, ms=check))
                cur[key], base = best_c, check
            else:
                for i in g["ops"]:
                    eng.set_cfg(i, names[cur[key]])
    final = measure()
    print(f"final forward: {final:.3f} ms")
    log.append(dict(step="final", ms=final))
    for key, g in groups.items():
        table[pre + tuning_key(ops[g["ops"][0]], args.batch, args.split)] = cur[key]
    json.dump(table, open(args.out, "w"), indent=0, sort_keys=True)
    if args.log:
        json.dump(log, open(args.log, "w"), indent=1)
    eng.close()


if __name__ == "__main__":
    main()
