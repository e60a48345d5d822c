#!/usr/bin/env python3
"""Network-part timing of one engine configuration (experiments; run with VGH_LIB_PATH=head_detector_amd/libvgh_exp.so to make the
VGH_STAGGER / VGH_GRID_SHARE / VGH_PATCH_PERSIST knobs live).
  python tools/net_probe.py vgg_heads_l 64 --split 2 --tuning gpurun_out/conv_cfg_q.json"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variant")
    ap.add_argument("batch", type=int)
    ap.add_argument("--split", type=int, default=2)
    ap.add_argument("--tuning", default=None)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--size", type=int, default=640)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    x = torch.randint(0, 256, (args.batch, args.size, args.size, 3), dtype=torch.uint8).to(dev)
    eng = VGHeadsEngine(args.variant, image_size=args.size, max_batch=args.batch, seed=1)
    if args.tuning:
        eng.load_tuning(args.tuning)
    eng.set_split(args.split)
    for _ in range(5):
        eng.forward_net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward_net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    knobs = {k: v for k, v in os.environ.items() if k.startswith("VGH_") and k != "VGH_LIB_PATH"}
    print(f"{args.variant} B={args.batch} split={args.split} tuning={os.path.basename(args.tuning) if args.tuning else 'default'} {knobs}: "
          f"{ms:.3f} ms/forward = {eng.flops_per_image * args.batch / ms / 1e9:.1f} TFLOP/s ({args.batch / ms * 1e3:.0f} img/s net only)")
    eng.close()


if __name__ == "__main__":
    main()
