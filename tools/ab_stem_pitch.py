#!/usr/bin/env python3
"""A/B on ONE box of the stem tensor's pitch in the bf16 mode: 48 channels (96-byte pixels, the stage-1 downsample's 64-channel K window overhangs into
the next pixel over zero weight columns) against 64 (16 stored zero channels, r01 - r03).  Two engines alive, alternating rounds, two-lane forwards, HIP
events on the engine stream; then the per-op device time of the stem and the downsample (vgh_net_profile, single lane).

    python tools/ab_stem_pitch.py [--rounds 4] [--steps 40]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd import arch  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for variant, B, S in (("vgg_heads_l", 64, 640), ("vgg_heads_m", 32, 640)):
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
        engs = {}
        for pitch in (64, 48):
            arch.STEM_PITCH_BF16 = pitch
            engs[pitch] = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
            engs[pitch].set_split(2)
            assert engs[pitch].program.bufs[0]["pitch"] == pitch
        arch.STEM_PITCH_BF16 = 48
        res = {p: [] for p in engs}
        for r in range(args.rounds):
            for p, eng in engs.items():
                for _ in range(6):
                    eng.forward_net(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for _ in range(args.steps):
                    eng.forward_net(x)
                e1.record(eng.stream)
                torch.cuda.synchronize()
                res[p].append(e0.elapsed_time(e1) / args.steps)
        for p, eng in engs.items():
            v = sorted(res[p])
            eng.set_split(1)
            ms = [eng.profile_ops(x) for _ in range(3)][-1]
            print(f"{variant} b{B}@{S} stem pitch {p}: two-lane forward min {v[0]:.3f} median {v[len(v) // 2]:.3f} ms; single-lane stem {ms[0]['ms'] * 1e3:.1f} us, "
                  f"stage-1 downsample {ms[1]['ms'] * 1e3:.1f} us", flush=True)
        a, b = sorted(res[64])[len(res[64]) // 2], sorted(res[48])[len(res[48]) // 2]
        print(f"{variant}: 48 vs 64 = {100.0 * (a - b) / a:+.2f} % (positive: the 48-channel pitch is faster)", flush=True)
        same = all(torch.equal(t48, t64) for t48, t64 in zip(engs[48].model(x), engs[64].model(x)))
        print(f"{variant}: engine.model() outputs bit-identical between the two pitches: {same}", flush=True)
        for eng in engs.values():
            eng.close()


if __name__ == "__main__":
    main()
