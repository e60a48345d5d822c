#!/usr/bin/env python3
"""Two half-batches as two free-running pipelines: does a phase offset between them (so that the HBM-bound 160x160 layers of one meet the
MFMA-bound 40x40 / 20x20 layers of the other) beat the lock-step two-lane forward?  usage: phase_probe.py [variant] [half_batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "vgg_heads_l"
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
lib = _lib.load()
N = 40

ref = VGHeadsEngine(variant, image_size=640, max_batch=2 * hb, seed=1)
ref.set_split(2)
x2 = torch.randint(0, 256, (2 * hb, 640, 640, 3), dtype=torch.uint8).to(dev)
for _ in range(5):
    ref.forward_net(x2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(ref.stream)
for _ in range(N):
    ref.forward_net(x2)
e1.record(ref.stream)
torch.cuda.synchronize()
t_ref = e0.elapsed_time(e1) / N
print(f"lock-step two-lane forward, batch {2 * hb}: {t_ref:.3f} ms")
ref.close()

a = VGHeadsEngine(variant, image_size=640, max_batch=hb, seed=1)
b = VGHeadsEngine(variant, image_size=640, max_batch=hb, seed=1)
a.load_tuning()
while lib.vgh_streams_overlap(a.stream.cuda_stream, b.stream.cuda_stream) != 1:  # two pool streams on one hardware queue: draw again
    b.stream = torch.cuda.Stream()
xa, xb = x2[:hb].contiguous(), x2[hb:].contiguous()
for _ in range(3):
    a.forward_net(xa)
    b.forward_net(xb)
torch.cuda.synchronize()
single = None
for offset_us in (0, 1000, 2000, 3000, 4000, 5000, 6500):
    torch.cuda.synchronize()
    import time

    t0 = time.perf_counter()
    if offset_us:
        lib.vgh_stream_spin(b.stream.cuda_stream, offset_us)
    for _ in range(N):
        a.forward_net(xa)
        b.forward_net(xb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"two free-running pipelines of {hb}, pipeline B delayed by {offset_us / 1000:.1f} ms: {(dt - offset_us / 1000) / N:.3f} ms per {2 * hb} images (lock-step {t_ref:.3f})")
