#!/usr/bin/env python3
"""A/B of a process-wide library knob on ONE box and ONE engine: two-lane forwards timed with HIP events on the engine stream, alternating off / on over
several rounds (DVFS and box variance cancel), plus the single-stream per-op table of both settings (JSON) for the layers it moves.

    python tools/ab_knob.py vgh_conv_set_nt_store [--rounds 5] [--json OUT.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("knob")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--json", default=None)
    ap.add_argument("--values", default="0,1")
    args = ap.parse_args()
    lib = _lib.load()
    setter = getattr(lib, args.knob)
    vals = [int(v) for v in args.values.split(",")]
    dev = torch.device("cuda", 0)
    out = {}
    for variant, B, S in (("vgg_heads_l", 64, 640), ("vgg_heads_m", 32, 640), ("vgg_heads_l", 16, 1280)):
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
        eng.set_split(1)
        per = {}
        for v in vals:
            _lib.check(setter(v))
            eng.profile_ops(x)
            per[v] = eng.profile_ops(x)
            print(f"{variant} b{B}@{S} single stream, {args.knob}={v}: sum of ops {sum(t['ms'] for t in per[v]):.3f} ms", flush=True)
        moved = sorted(range(len(per[vals[0]])), key=lambda i: per[vals[-1]][i]["ms"] - per[vals[0]][i]["ms"])
        for i in moved[:8] + moved[-4:]:
            print(f"    {per[vals[0]][i]['name']:44s} {per[vals[0]][i]['ms'] * 1e3:8.1f} -> {per[vals[-1]][i]['ms'] * 1e3:8.1f} us")
        eng.set_split(2)
        res = {v: [] for v in vals}
        for r in range(args.rounds):
            for v in vals:
                _lib.check(setter(v))
                for _ in range(8):
                    eng.forward_net(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for _ in range(args.steps):
                    eng.forward_net(x)
                e1.record(eng.stream)
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / args.steps)
        for v in vals:
            q = sorted(res[v])
            print(f"{variant} b{B}@{S} two lanes, {args.knob}={v}: min {q[0]:.3f} median {q[len(q) // 2]:.3f} ms/forward = {eng.flops_per_image * B / q[len(q) // 2] / 1e9:.1f} TFLOP/s", flush=True)
        out[f"{variant}_b{B}_{S}"] = {"two_lane_ms": res, "single_stream_ops": {v: [(t["name"], t["ms"]) for t in per[v]] for v in vals}}
        _lib.check(setter(vals[0]))
        eng.close()
    if args.json:
        json.dump(out, open(args.json, "w"))


if __name__ == "__main__":
    main()
