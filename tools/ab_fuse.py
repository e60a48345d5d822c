#!/usr/bin/env python3
"""A/B on ONE box and ONE engine: stem + stage-1 downsample fused (csrc/stem_ds.hip) vs as two launches; two-lane forwards timed with HIP events on the
engine stream, alternating, plus the single-stream per-op times of the two ops.   python tools/ab_fuse.py [--rounds 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

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
        eng.set_split(1)
        for fuse in (True, False):
            eng.set_fuse_stem(fuse)
            eng.profile_ops(x)
            t = eng.profile_ops(x)
            print(f"{variant} b{B}@{S} single stream, fuse={fuse}: stem op {t[0]['ms'] * 1e3:.1f} us, stage1.downsample op {t[1]['ms'] * 1e3:.1f} us", flush=True)
        eng.set_split(2)
        res = {True: [], False: []}
        for r in range(args.rounds):
            for fuse in (True, False):
                eng.set_fuse_stem(fuse)
                for _ in range(8):
                    eng.forward_net(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for _ in range(args.steps):
                    eng.forward_net(x)
                e1.record(eng.stream)
                torch.cuda.synchronize()
                res[fuse].append(e0.elapsed_time(e1) / args.steps)
        for fuse in (True, False):
            v = sorted(res[fuse])
            print(f"{variant} b{B}@{S} two lanes, fuse={fuse}: min {v[0]:.3f} median {v[len(v) // 2]:.3f} ms/forward = {eng.flops_per_image * B / v[len(v) // 2] / 1e9:.1f} TFLOP/s", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
