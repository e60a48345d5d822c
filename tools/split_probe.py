#!/usr/bin/env python3
"""Experiment: network time of one engine with vgh_net_set_split(1..4) (lane streams created inside libvgh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8).to(dev)
eng = VGHeadsEngine("vgg_heads_m", image_size=640, max_batch=B, seed=1)
for ns in (1, 2, 3, 4, 2, 1):
    eng.set_split(ns)
    for _ in range(3):
        eng.forward_net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.forward_net(x)
    torch.cuda.synchronize()
    print(f"split {ns}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
