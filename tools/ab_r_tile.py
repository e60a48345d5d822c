#!/usr/bin/env python3
"""Per-op times of the 96-channel stride-2 convs under two tile tables (single stream), e.g. the shipped table vs one with the r tile:  python tools/ab_r_tile.py A.json B.json"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for variant, B, S in (("vgg_heads_l", 64, 640), ("vgg_heads_m", 32, 640), ("vgg_heads_l", 16, 1280)):
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
        eng.set_split(1)
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
        for t in sys.argv[1:]:
            for i, op in enumerate(eng.program.ops):
                if op["kind"] == 1:
                    eng.set_cfg(i, -1)
            eng.load_tuning(t)
            eng.profile_ops(x)
            p = eng.profile_ops(x, repeats=5)
            sel = [(o["name"], r["ms"] * 1e3, r["gbps"]) for o, r in zip(eng.program.ops, p) if o["kind"] == 1 and o["ksize"] == 3 and o["stride"] == 2 and o["cin"] == 96]
            print(f"{variant} b{B}@{S} {os.path.basename(t)}: " + "; ".join(f"{n} {us:.1f} us ({g:.0f} GB/s)" for n, us, g in sel) + f"; sum of all ops {sum(r['ms'] for r in p):.3f} ms", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
