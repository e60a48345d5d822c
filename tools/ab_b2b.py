#!/usr/bin/env python3
"""A/B of the back-to-back GEMM pairs (vgh_net_set_b2b) on ONE box and ONE engine: alternating fused / two launches, two-lane and single-lane forwards, HIP events
on the engine stream; plus the per-op times of the pair's ops in both modes.   python tools/ab_b2b.py [--rounds 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd import arch  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for variant, B, S in (("vgg_heads_l", 64, 640), ("vgg_heads_m", 32, 640), ("vgg_heads_l", 16, 1280)):
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
        pairs = arch.b2b_pairs(eng.program)
        for ns in (2, 1):
            eng.set_split(ns)
            res = {True: [], False: []}
            for r in range(args.rounds):
                for fused in (True, False):
                    eng.set_b2b(fused)
                    for _ in range(8):
                        eng.forward_net(x)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(eng.stream)
                    for _ in range(args.steps):
                        eng.forward_net(x)
                    e1.record(eng.stream)
                    torch.cuda.synchronize()
                    res[fused].append(e0.elapsed_time(e1) / args.steps)
            for fused in (True, False):
                v = sorted(res[fused])
                print(f"{variant} b{B}@{S} lanes {ns} {'b2b fused ' if fused else 'two launches'}: min {v[0]:.3f} median {v[len(v) // 2]:.3f} ms/forward = {eng.flops_per_image * B / v[len(v) // 2] / 1e9:.1f} TFLOP/s", flush=True)
        eng.set_split(1)
        for fused in (True, False):
            eng.set_b2b(fused)
            rows = eng.profile_ops(x)
            rows = eng.profile_ops(x)
            for i in pairs:
                print(f"   per-op, single stream, {'fused' if fused else 'two launches'}: {rows[i]['name']} {rows[i]['ms'] * 1e3:.1f} us + {rows[i + 1]['name']} {rows[i + 1]['ms'] * 1e3:.1f} us = {(rows[i]['ms'] + rows[i + 1]['ms']) * 1e3:.1f} us", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
