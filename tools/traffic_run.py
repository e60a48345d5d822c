#!/usr/bin/env python3
"""Exactly N network forwards and nothing else (no threshold calibration, no post-network stages): the command the PMC passes wrap, so
that `counter sum / N` IS bytes (or busy cycles) per forward.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d DIR -o FETCH_SIZE -- python tools/traffic_run.py --forwards 8 --split 2
    python tools/pmc_summary.py DIR vgg_heads_l 64 8 2 OUT_DIR        # -> r03_traffic_l64_x2.json, r03_pmc_net_l64_x2.txt
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="vgg_heads_l")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=640)
    ap.add_argument("--forwards", type=int, default=8)
    ap.add_argument("--split", type=int, default=2)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--b2b", type=int, default=1, help="0: the back-to-back pairs as two launches each (one dispatch per op: what the per-op PMC tools expect)")
    ap.add_argument("--knob", action="append", default=[], help="process-wide library knob, e.g. vgh_conv_set_nt_store=1")
    args = ap.parse_args()
    import torch

    from head_detector_amd import arch
    from head_detector_amd.engine import VGHeadsEngine

    dev = torch.device("cuda", 0)
    for kv in args.knob:
        from head_detector_amd import _lib

        name, val = kv.split("=")
        _lib.check(getattr(_lib.load(), name)(int(val)))
    eng = VGHeadsEngine(args.variant, image_size=args.image_size, max_batch=args.batch, seed=1, precision=args.precision)
    eng.set_split(args.split)
    eng.set_b2b(args.b2b if args.b2b in (2, 3) else bool(args.b2b))
    x = torch.randint(0, 256, (args.batch, args.image_size, args.image_size, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    for _ in range(args.forwards):
        eng.forward_net(x)
    torch.cuda.synchronize()
    alg = arch.program_algorithmic_bytes(eng.program, args.batch, fused_stem=eng.stem_fused, b2b=bool(args.b2b))
    print(json.dumps(dict(variant=args.variant, batch=args.batch, forwards=args.forwards, split=args.split, precision=args.precision,
                          ops_per_forward=sum(1 for op in eng.program.ops if op["kind"] in (0, 1, 2)) - (eng.b2b_pairs if args.b2b else 0) - int(eng.stem_fused), b2b_pairs=eng.b2b_pairs if args.b2b else 0,
                          algorithmic_read_bytes=alg["read"], algorithmic_write_bytes=alg["write"])))
    eng.close()


if __name__ == "__main__":
    main()
