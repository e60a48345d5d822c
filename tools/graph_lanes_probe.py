#!/usr/bin/env python3
"""Experiment: head levels on side lanes (VGH_OP_FORK) eager vs replayed as a hipGraph with parallel branches, vs the batch split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd import arch  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8).to(dev)


def run(eng, graph):
    for _ in range(3):
        eng.forward_net(x, use_graph=graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.forward_net(x, use_graph=graph)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


orig = arch.build_program
for lanes in (False, True):
    arch.build_program = lambda *a, _l=lanes, **k: orig(*a, head_lanes=_l, **k)
    eng = VGHeadsEngine("vgg_heads_m", image_size=640, max_batch=B, seed=1)
    print(f"head_lanes={lanes}: eager {run(eng, False):.3f} ms, graph {run(eng, True):.3f} ms", end="")
    if not lanes:
        eng.set_split(2)
        print(f", split2 eager {run(eng, False):.3f} ms, split2 graph {run(eng, True):.3f} ms")
    else:
        print()
    eng.close()
