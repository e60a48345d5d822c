#!/usr/bin/env python3
"""fp8 mode (e4m3 links) against the bf16 mode on one box: two-lane network time, alternating; per-op table of the fp8 engine; dense-output deviation of both
modes from the fp16x3 parity mode on four seeded images.      python tools/fp8_probe.py [variant] [batch] [--per-layer out.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def iou(a, b):
    lt, rb = torch.maximum(a[..., :2], b[..., :2]), torch.minimum(a[..., 2:], b[..., 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variant", nargs="?", default="vgg_heads_l")
    ap.add_argument("batch", nargs="?", type=int, default=64)
    ap.add_argument("--per-layer", default=None)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--min-px", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.batch
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    engs = {p: VGHeadsEngine(args.variant, image_size=640, max_batch=B, seed=1, precision=p, fp8_min_px=args.min_px) for p in ("bf16", "fp8")}
    for e in engs.values():
        e.set_split(2)
    K = 30

    def t(e):
        for _ in range(5):
            e.forward_net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            e.forward_net(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    res = {p: [] for p in engs}
    for _ in range(args.rounds):
        for p, e in engs.items():
            res[p].append(t(e))
    fl = engs["bf16"].flops_per_image * B
    for p, v in res.items():
        m = min(v)
        print(f"{args.variant} B={B} two lanes {p:5s}: min {m:.3f} ms ({', '.join(f'{a:.3f}' for a in v)}) = {fl / (m * 1e-3) / 1e12:.1f} TFLOP/s algorithmic, {B / m * 1e3:.0f} img/s net only")
    if args.per_layer:
        e = engs["fp8"]
        e.set_split(1)
        tab = e.profile_ops(x)
        json.dump(tab, open(args.per_layer, "w"), indent=0)
        e0 = engs["bf16"]
        e0.set_split(1)
        tab0 = {o["name"]: o for o in e0.profile_ops(x)}
        P = e.program
        print("ops on e4m3 links (single stream, us): bf16 -> fp8")
        for o, op in zip(tab, P.ops):
            from head_detector_amd import arch

            if arch.op_touches_fp8(P, op):
                print(f"  {o['name']:50s} {tab0[o['name']]['ms'] * 1e3:8.1f} -> {o['ms'] * 1e3:8.1f}   ({o['tflops']:.0f} TFLOP/s)")
        print(f"sum of ops: bf16 {sum(o['ms'] for o in tab0.values()):.3f} ms, fp8 {sum(o['ms'] for o in tab):.3f} ms")
    for e in engs.values():
        e.close()
    # deviation on four seeded images against the matrix-core parity mode
    xs = torch.rand(4, 3, 640, 640, generator=torch.Generator().manual_seed(3)).to(dev)
    outs = {}
    for p in ("fp16x3", "bf16", "fp8"):
        e = VGHeadsEngine(args.variant, image_size=640, max_batch=4, seed=1, precision=p, fp8_min_px=args.min_px)
        e.model(xs)
        torch.cuda.synchronize()
        outs[p] = (e.boxes_all[:4].cpu().clone(), e.scores_all[:4].cpu().clone(), e.idx[:4, :100].cpu().long().clone())
        e.close()
    rb, rs, _ = outs["fp16x3"]
    for p in ("bf16", "fp8"):
        b, s, idx = outs[p]
        top = torch.stack([iou(b[i, idx[i]], rb[i, idx[i]]).min() for i in range(4)])
        print(f"{args.variant} {p:5s} vs fp16x3: dense IoU min {float(iou(b, rb).min()):.4f} mean {float(iou(b, rb).mean()):.5f}  top-100 IoU min {float(top.min()):.4f}  score max abs err {float((s - rs).abs().max()):.2e}")


if __name__ == "__main__":
    main()
