#!/usr/bin/env python3
"""A/B of tile tables on ONE box and ONE engine (box-to-box spread is +-3 %, larger than most table differences): alternates the tables,
two-lane forwards, HIP events on the engine stream.   python tools/ab_table.py TABLE_A.json TABLE_B.json [--rounds 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tables", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for variant, B, S in (("vgg_heads_l", 64, 640), ("vgg_heads_m", 32, 640), ("vgg_heads_l", 16, 1280)):
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
        eng.set_split(2)
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
        res = {t: [] for t in args.tables}
        for r in range(args.rounds):
            for t in args.tables:
                for i, op in enumerate(eng.program.ops):  # back to automatic before applying a table (a table may not cover every op)
                    if op["kind"] == 1:
                        eng.set_cfg(i, -1)
                n = eng.load_tuning(t)
                for _ in range(8):
                    eng.forward_net(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for _ in range(args.steps):
                    eng.forward_net(x)
                e1.record(eng.stream)
                torch.cuda.synchronize()
                res[t].append(e0.elapsed_time(e1) / args.steps)
        for t in args.tables:
            v = sorted(res[t])
            print(f"{variant} b{B}@{S} {os.path.basename(t)} ({n} ops): min {v[0]:.3f} median {v[len(v) // 2]:.3f} ms/forward = {eng.flops_per_image * B / v[len(v) // 2] / 1e9:.1f} TFLOP/s", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
