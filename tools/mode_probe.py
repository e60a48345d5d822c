#!/usr/bin/env python3
"""Two-lane network time of several precision modes on ONE box, alternating (default: bf16, fp16, fp8), + the dense deviation of each from the fp16x3 parity mode
on four seeded images.     python tools/mode_probe.py [variant] [batch] [--modes bf16,fp16,fp8]"""
import argparse
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
    ap.add_argument("--modes", default="bf16,fp16,fp8")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--per-layer", default=None, help="mode:path -- single-stream per-op table of that mode")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.batch
    modes = args.modes.split(",")
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    engs = {p: VGHeadsEngine(args.variant, image_size=640, max_batch=B, seed=1, precision=p) for p in modes}
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
    fl = engs[modes[0]].flops_per_image * B
    for p, v in res.items():
        m = min(v)
        print(f"{args.variant} B={B} two lanes {p:6s}: min {m:.3f} ms ({', '.join(f'{a:.3f}' for a in v)}) = {fl / (m * 1e-3) / 1e12:.1f} TFLOP/s algorithmic, {B / m * 1e3:.0f} img/s net only")
    if args.per_layer:
        import json

        mode, path = args.per_layer.split(":", 1)
        e = engs[mode]
        e.set_split(1)
        tab = e.profile_ops(x)
        json.dump(tab, open(path, "w"), indent=0)
        base = engs[modes[0]]
        base.set_split(1)
        t0 = {o["name"]: o["ms"] for o in base.profile_ops(x)}
        slow = sorted(tab, key=lambda o: o["ms"] - t0.get(o["name"], 0.0), reverse=True)[:14]
        print(f"single stream: {modes[0]} {sum(t0.values()):.3f} ms, {mode} {sum(o['ms'] for o in tab):.3f} ms; ops that lose most ({modes[0]} -> {mode}, us):")
        for o in slow:
            print(f"   {o['name']:48s} {t0.get(o['name'], 0) * 1e3:8.1f} -> {o['ms'] * 1e3:8.1f}")
    for e in engs.values():
        e.close()
    xs = torch.rand(4, 3, 640, 640, generator=torch.Generator().manual_seed(3)).to(dev)
    outs = {}
    for p in ["fp16x3"] + modes:
        e = VGHeadsEngine(args.variant, image_size=640, max_batch=4, seed=1, precision=p)
        e.model(xs)
        torch.cuda.synchronize()
        outs[p] = (e.boxes_all[:4].cpu().clone(), e.scores_all[:4].cpu().clone(), e.idx[:4, :100].cpu().long().clone())
        e.close()
    rb, rs, _ = outs["fp16x3"]
    for p in modes:
        b, s, idx = outs[p]
        top = torch.stack([iou(b[i, idx[i]], rb[i, idx[i]]).min() for i in range(4)])
        print(f"{args.variant} {p:6s} vs fp16x3: dense IoU min {float(iou(b, rb).min()):.4f} mean {float(iou(b, rb).mean()):.5f}  top-100 IoU min {float(top.min()):.4f}  score max abs err {float((s - rs).abs().max()):.2e}")


if __name__ == "__main__":
    main()
