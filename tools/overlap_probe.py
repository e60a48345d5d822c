#!/usr/bin/env python3
"""What the post-network stages cost the step, on one box, alternating: (a) network forwards only, (b) the bench step with the post stages on the detector's low-priority
side stream (overlap mode: the side stream waits ON THE DEVICE for each forward's end), (c) the same with the post stages on the engine stream (no side stream).
    python tools/overlap_probe.py [variant] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402
from head_detector_amd.flame import FLAMELayer  # noqa: E402
from head_detector_amd.synthetic import synthetic_flame_model  # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "vgg_heads_l"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda", 0)
    flame = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=B * 100)
    eng = VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1)
    images = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    unpad = torch.tensor([[0.0, 0.0, 1.0]], device=dev).expand(B, 3).contiguous()
    _, scores, _ = eng.model(images)
    conf = float(torch.sort(scores.flatten(), descending=True).values[3 * B])
    eng.set_split(2)
    K = 48
    n_heads_all = torch.zeros(K, dtype=torch.int32, device=dev)
    res = {"network forwards only": [], "step, post stages on the low-priority side stream (overlap)": [], "step, post stages on the engine stream": []}
    for rnd in range(4):
        for name in res:
            if name.startswith("network"):
                eng.set_overlap(False)
                fn = lambda i=None: eng.forward_net(images)  # noqa: E731
            else:
                ov = "side stream" in name
                eng.set_overlap(ov)
                fn = bench.make_step(eng, flame, images, unpad, conf, B, None, None, ov, False, n_heads_all)
            for _ in range(10):
                fn()
            eng.join()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                fn(i)
            eng.join()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / K * 1e3)
    for name, v in res.items():
        print(f"{variant} B={B} {name:70s}: min {min(v):7.3f} ms per forward   all: {', '.join(f'{a:.3f}' for a in v)}")
    eng.close()


if __name__ == "__main__":
    main()
