#!/usr/bin/env python3
"""A/B of the lazy FLAME gather (r06, vgh_detector_set_lazy_flame) on ONE box: the bench's step (two-lane network -> candidates -> select with FLAME decode, overlap mode)
with the candidate stage gathering all 1 000 candidates' 413-vectors (eager) or boxes only (lazy), alternating.   python tools/ab_lazy_gather.py [variant batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402
from head_detector_amd.flame import FLAMELayer  # noqa: E402
from head_detector_amd.synthetic import synthetic_flame_model  # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "vgg_heads_l"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda", 0)
    fl = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=B * 64)
    eng = VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1)
    eng.set_split(2)
    eng.set_overlap(True)
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    conf = float(eng.model(x)[1][:, 3, 0].median())
    slot = eng.new_output_slot(fl, B)

    def run(lazy, n):
        for _ in range(n):
            eng.forward_net(x)
            eng.candidates(B, lazy_flame=lazy)
            eng.select(B, confidence_threshold=conf, flame=fl, slot=slot)
        eng.join()
        torch.cuda.synchronize()

    for rnd in range(3):
        for lazy in (False, True):
            run(lazy, 40)
            t0 = time.perf_counter()
            run(lazy, 200)
            print(f"{variant} b{B} {'lazy ' if lazy else 'eager'} gather: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per forward", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
