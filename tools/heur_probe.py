#!/usr/bin/env python3
"""Network time with the measured tile table vs the built-in class heuristic (use_tuning=False), single stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402

dev = torch.device("cuda", 0)
for variant, B, S in (("vgg_heads_m", 32, 640), ("vgg_heads_l", 64, 640), ("vgg_heads_m", 16, 1280), ("vgg_heads_m", 64, 320), ("vgg_heads_l", 4, 640)):
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).to(dev)
    res = []
    for tuned in (True, False):
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1, use_tuning=tuned)
        for _ in range(3):
            eng.forward_net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.forward_net(x)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 10 * 1e3)
        eng.close()
    print(f"{variant} B={B} @{S}: table {res[0]:.3f} ms | heuristic only {res[1]:.3f} ms ({(res[1] / res[0] - 1) * 100:+.1f} %)")
