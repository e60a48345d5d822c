"""Does the order in which engines are created in one process change the network time?  (M, L, M, pause, M)"""
import sys, time
import torch
from head_detector_amd.engine import VGHeadsEngine

dev = torch.device("cuda", 0)


def run(variant, B, iters=60, split=2, keep=False):
    eng = VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1)
    eng.set_split(split)
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8).to(dev)
    for _ in range(5):
        eng.forward_net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(eng.stream)
    for _ in range(iters):
        eng.forward_net(x)
    e1.record(eng.stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{variant} b{B} split{split}: {ms:.3f} ms  mem {torch.cuda.memory_allocated() >> 20} MiB torch, free {torch.cuda.mem_get_info()[0] >> 20} MiB", flush=True)
    if not keep:
        eng.close()
    return eng


mode = sys.argv[1] if len(sys.argv) > 1 else "close"
if mode == "split":  # lane count with every lane on its own hardware queue
    for v, b in (("vgg_heads_m", 32), ("vgg_heads_l", 64)):
        for sp in (1, 2, 3, 4):
            run(v, b, split=sp)
elif mode == "close":  # every engine is closed before the next is created
    for v, b in (("vgg_heads_m", 32), ("vgg_heads_m", 32), ("vgg_heads_l", 64), ("vgg_heads_m", 32), ("vgg_heads_m", 32), ("vgg_heads_l", 64), ("vgg_heads_l", 64)):
        run(v, b)
else:  # keep every engine alive: no stream is ever destroyed
    held = [run(v, b, keep=True) for v, b in (("vgg_heads_m", 32), ("vgg_heads_m", 32), ("vgg_heads_l", 64), ("vgg_heads_m", 32), ("vgg_heads_l", 64), ("vgg_heads_l", 64))]
