#!/usr/bin/env python3
"""The stem inside the pair's launch (the default mode of vgh_net_set_b2b) against stem launch + t tile: stem values (a -DVGH_DT_DEBUG_STEM build writes them to the stem
tensor) and the pair's output, in bf16 ulps.   VGH_LIB_PATH=head_detector_amd/libvgh_dbg.so python tools/dbg_stem.py [S B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


def ulps(a, b):
    ia, ib = a.view(torch.int16).int(), b.view(torch.int16).int()
    return (ia - ib).abs()


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    eng = VGHeadsEngine("vgg_heads_l", image_size=S, max_batch=B, seed=11, use_tuning=False)
    P = eng.program
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(S)).to(dev)
    eng.set_split(1)
    eng.set_b2b(3)
    res_ref = [t.clone() for t in eng.model(x)]
    stem_ref = eng.buffer(P.ops[0]["out_buf"], B).clone()
    out_ref = eng.buffer(P.ops[2]["out_buf"], B).clone()
    eng.buffer(P.ops[0]["out_buf"], B).zero_()
    eng.set_b2b(1)
    res = [t.clone() for t in eng.model(x)]
    stem = eng.buffer(P.ops[0]["out_buf"], B).clone()
    out = eng.buffer(P.ops[2]["out_buf"], B).clone()
    if float(stem.float().abs().max()) > 0:
        u = ulps(stem, stem_ref)
        print(f"stem: {int((u > 0).sum())} of {u.numel()} values differ ({float((u > 0).float().mean()):.2e}), max {int(u.max())} bf16 ulps, nan {int(torch.isnan(stem.float()).sum())}")
        if int(u.max()) > 1:
            idx = (u > 1).nonzero()
            print("   > 1 ulp at", idx[:8].tolist(), "got", stem[tuple(idx[0].tolist())].item(), "ref", stem_ref[tuple(idx[0].tolist())].item())
    else:
        print("stem tensor untouched (not a debug build)")
    o2 = P.ops[2]
    own = lambda t: torch.cat([t[..., o2["out_coff"]:o2["out_coff"] + o2["out_split"]], t[..., o2["out_coff2"]:o2["out_coff2"] + o2["cout_store"] - o2["out_split"]]], -1)  # noqa: E731
    u = ulps(own(out), own(out_ref))
    print(f"pair output: {int((u > 0).sum())} of {u.numel()} values differ ({float((u > 0).float().mean()):.2e}), max {int(u.max())} bf16 ulps, max abs diff {float((own(out).float() - own(out_ref).float()).abs().max()):.4f}")
    for a, b, n in zip(res, res_ref, ("boxes", "scores", "flame")):
        print(f"network output {n}: max abs diff {float((a.float() - b.float()).abs().max()):.3e} (max |ref| {float(b.float().abs().max()):.3e})")
    eng.close()


if __name__ == "__main__":
    main()
