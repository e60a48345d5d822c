#!/usr/bin/env python3
"""A/B on ONE box and ONE engine: the stage-1 downsample + conv1|conv2 pair on the persistent "t" tile (csrc/ds_b2b.hip, set_b2b(1)) vs on the implicit-GEMM b2b tile
(set_b2b(2)); two-lane forwards timed with HIP events on the engine stream, alternating, plus the single-stream per-op time of the pair.   python tools/ab_ds_tile.py"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


NAMES = {1: 'u tile (stem + pair, default)', 3: 't tile (pair; stem launch)', 2: 'igemm b2b'}
MODES = (1, 3, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for variant, B, S in (("vgg_heads_l", 64, 640), ("vgg_heads_m", 32, 640), ("vgg_heads_l", 16, 1280)):
        if args.only and args.only != variant + str(B):
            continue
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
        eng.set_split(1)
        for mode in MODES:
            eng.set_b2b(mode)
            eng.profile_ops(x)
            t = eng.profile_ops(x)
            print(f"{variant} b{B}@{S} single stream, {NAMES[mode]}: stem {t[0]['ms'] * 1e3:.1f} us, stage1.downsample + conv1|conv2 {t[1]['ms'] * 1e3:.1f} us", flush=True)
        eng.set_split(2)
        res = {m: [] for m in MODES}
        for r in range(args.rounds):
            for mode in MODES:
                eng.set_b2b(mode)
                for _ in range(8):
                    eng.forward_net(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for _ in range(args.steps):
                    eng.forward_net(x)
                e1.record(eng.stream)
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) / args.steps)
        for mode in MODES:
            v = sorted(res[mode])
            print(f"{variant} b{B}@{S} two lanes, {NAMES[mode]}: min {v[0]:.3f} median {v[len(v) // 2]:.3f} ms/forward = {eng.flops_per_image * B / v[len(v) // 2] / 1e9:.1f} TFLOP/s", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
