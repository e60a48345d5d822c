#!/usr/bin/env python3
"""Experiment (r05): the batch as two INDEPENDENT half-batch pipelines that never join, a fixed phase apart.

The batch split of vgh_net_forward runs both lanes through the same op at the same time and joins them at the end of every forward
(lane lag k costs the tail: profiles/r03_ab_lane_lag.txt).  Here two engines of B/2 images each run free on their own streams, forward after
forward, and engine 1 starts `phase` of a forward late -- so an HBM-bound op of one pipeline sits next to an MFMA-bound op of the other in
steady state, with no tail.  Network part only; prints ms per B images for the joined two-lane forward and for every phase.

    python tools/phase_pipeline.py [variant] [B] [--inner-split 1|2]
"""
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
    ap.add_argument("variant", nargs="?", default="vgg_heads_l")
    ap.add_argument("batch", nargs="?", type=int, default=64)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--phases", default="0,0.125,0.25,0.375,0.5,0.625,0.75")
    ap.add_argument("--inner-split", default="1,2")
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, K = args.batch, args.iters
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    halves = [h.contiguous() for h in x.chunk(2)]
    flops = None

    def wall(fn, streams):
        for _ in range(6):
            fn(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(True)
        for _ in range(K - 1):
            fn(False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    # sleep calibration: cycles of torch.cuda._sleep per millisecond
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(20_000_000)
    e1.record()
    torch.cuda.synchronize()
    cyc_per_ms = 20_000_000 / e0.elapsed_time(e1)

    full = VGHeadsEngine(args.variant, image_size=640, max_batch=B, seed=1)
    flops = full.flops_per_image * B
    full.set_split(2)
    res = {}
    for rnd in range(args.rounds):
        res.setdefault("joined two-lane forward (vgh_net_set_split 2)", []).append(wall(lambda first: full.forward_net(x), None))
    full.set_split(1)
    res.setdefault("one stream, whole batch", []).append(wall(lambda first: full.forward_net(x), None))
    full.close()

    for isp in [int(v) for v in args.inner_split.split(",")]:
        engs = [VGHeadsEngine(args.variant, image_size=640, max_batch=B // 2, seed=1) for _ in range(2)]
        for e in engs:
            e.set_split(isp)
        one = wall(lambda first: engs[0].forward_net(halves[0]), None)
        res.setdefault(f"half batch alone, inner split {isp} (x2)", []).append(2 * one)
        for rnd in range(args.rounds):
            for ph in [float(v) for v in args.phases.split(",")]:
                def fn(first, ph=ph):
                    if first and ph > 0:
                        with torch.cuda.stream(engs[1].stream):
                            torch.cuda._sleep(int(ph * one * cyc_per_ms))
                    engs[0].forward_net(halves[0])
                    engs[1].forward_net(halves[1])

                res.setdefault(f"two free-running half-batch pipelines, inner split {isp}, phase {ph:.3f}", []).append(wall(fn, None))
        for e in engs:
            e.close()
    for k, v in res.items():
        m = min(v)
        print(f"{args.variant} B={B}: {k:80s} min {m:7.3f} ms  ({', '.join(f'{t:.3f}' for t in v)})  = {flops / (m * 1e-3) / 1e12:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
