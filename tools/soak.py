#!/usr/bin/env python3
"""Soak test of the throughput mode (batch-split lanes + post stages on the side stream): alternate two batches for many
iterations, queueing batch s+1's network before batch s's results are read, and require bit-identical outputs every time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd.engine import VGHeadsEngine  # noqa: E402
from head_detector_amd.flame import FLAMELayer  # noqa: E402
from head_detector_amd.synthetic import synthetic_flame_model  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    variant = sys.argv[2] if len(sys.argv) > 2 else "vgg_heads_m"
    B, S = (int(sys.argv[3]) if len(sys.argv) > 3 else 16), 640
    g = torch.Generator().manual_seed(7)
    xs = [torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(2)]
    fl = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=B * 100)
    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
    conf = float(eng.model(xs[0])[1][:, 40, 0].max())

    def snap(d):
        return [t.clone() for t in (d.boxes, d.scores, d.flame_params, d.counts, d.vertices_3d, d.head_pose)]

    eng.set_split(2)  # before the references: a tuned engine runs the two-lane table's tiles (another summation order than the one-lane table's on some shapes)
    refs = [snap(eng.detect(x, confidence_threshold=conf, flame=fl)) for x in xs]
    assert int(refs[0][3].sum()) > 0
    eng.set_overlap(True)
    eng.forward_net(xs[0])
    eng.candidates(B)
    det = eng.select(B, confidence_threshold=conf, flame=fl)
    bad = 0
    for it in range(1, iters + 1):
        nxt = it % 2
        eng.forward_net(xs[nxt])  # next batch's network is queued before the previous results are consumed
        eng.join()
        got = snap(det)
        for r, q in zip(refs[(it - 1) % 2], got):
            if not torch.equal(r, q):
                bad += 1
                break
        eng.candidates(B)
        det = eng.select(B, confidence_threshold=conf, flame=fl)
    print(f"soak: {iters} iterations, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
