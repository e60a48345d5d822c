#!/usr/bin/env python3
"""The w tiles (csrc/ds_b2b.hip: 3x3 / stride-1 convs with 96 / 128 input channels, weights resident in registers) against the shipped tile table, op by op (single stream,
per-op median of 5 profiled forwards) and on the two-lane forward (alternating, HIP events), ONE box and ONE engine.   python tools/ab_w_tile.py [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine, tuning_key  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    picks = {}
    for variant, B, S in (("vgg_heads_l", 64, 640), ("vgg_heads_m", 32, 640), ("vgg_heads_l", 16, 1280)):
        if args.only and args.only != f"{variant}{B}":
            continue
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
        names = eng.cfg_names()
        wt = {96: names.index("w8x8x96_n3"), 128: names.index("w8x8x128_n4")}
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
        ops = eng.program.ops
        elig = [i for i, op in enumerate(ops) if op["kind"] == 1 and op["ksize"] == 3 and op["stride"] == 1 and op["cin"] in wt and eng.cfg_ok(wt[op["cin"]], op)]

        def apply(idx_on):
            for i, op in enumerate(ops):
                if op["kind"] == 1:
                    eng.set_cfg(i, -1)
            eng.load_tuning()
            for i in idx_on:
                eng.set_cfg(i, wt[ops[i]["cin"]])

        eng.set_split(1)
        apply([])
        eng.profile_ops(x)
        base = eng.profile_ops(x, repeats=5)
        apply(elig)
        eng.profile_ops(x)
        new = eng.profile_ops(x, repeats=5)
        better = []
        for i in elig:
            tag = "+" if new[i]["ms"] < base[i]["ms"] else "-"
            print(f"{variant} b{B}@{S} {ops[i]['name']:48s} gemm {ops[i]['gemm']} res={ops[i].get('res_buf', -1) >= 0}: table {base[i]['ms'] * 1e3:7.1f} us ({base[i]['tflops']:6.0f} TF)  w tile {new[i]['ms'] * 1e3:7.1f} us ({new[i]['tflops']:6.0f} TF) {tag}", flush=True)
            if new[i]["ms"] < base[i]["ms"]:
                better.append(i)
        print(f"{variant} b{B}@{S} single stream: sum of all ops {sum(r['ms'] for r in base):.3f} -> {sum(r['ms'] for r in new):.3f} ms; {len(better)} of {len(elig)} eligible ops faster", flush=True)
        eng.set_split(2)
        res80 = [i for i in elig if ops[i].get("res_buf", -1) >= 0 and ops[i]["cin"] == 128 and new[i]["ms"] < 0.92 * base[i]["ms"]]
        res = {"table": [], "w_all": [], "w_better": [], "w_res_big_gain": []}
        for r in range(args.rounds):
            for mode, idx in (("table", []), ("w_all", elig), ("w_better", better), ("w_res_big_gain", res80)):
                apply(idx)
                for _ in range(8):
                    eng.forward_net(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for _ in range(args.steps):
                    eng.forward_net(x)
                e1.record(eng.stream)
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) / args.steps)
        for mode in res:
            v = sorted(res[mode])
            print(f"{variant} b{B}@{S} two lanes, {mode}: min {v[0]:.3f} median {v[len(v) // 2]:.3f} ms/forward = {eng.flops_per_image * B / v[len(v) // 2] / 1e9:.1f} TFLOP/s", flush=True)
        for i in better:
            for ns in (1, 2):
                picks[tuning_key(ops[i], B, ns)] = names[wt[ops[i]["cin"]]]
        eng.close()
    if args.json:
        json.dump(picks, open(args.json, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
