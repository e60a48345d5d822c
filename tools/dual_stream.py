#!/usr/bin/env python3
"""Experiment: does running the batch as two independent half-batches on two HIP streams (kernel prologues / epilogues / tails of
one stream under the main loops of the other) beat one full-batch stream?  Network part only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "vgg_heads_m"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    nsplit = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev = torch.device("cuda", 0)
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8).to(dev)
    engs = [VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1) for _ in range(nsplit)]
    parts = [p.contiguous() for p in x.chunk(nsplit)]
    K = 20

    def run(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    one = run(lambda: engs[0].forward_net(x))

    def dual():
        for e, p in zip(engs, parts):
            e.forward_net(p)

    two = run(dual)
    half = run(lambda: engs[0].forward_net(parts[0]))
    print(f"{variant} B={B}: one stream {one:.3f} ms | {nsplit} streams x B={B // nsplit}: {two:.3f} ms | single half-batch alone {half:.3f} ms (x{nsplit} = {half * nsplit:.3f})")


if __name__ == "__main__":
    main()
