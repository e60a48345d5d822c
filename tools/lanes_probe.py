"""Two-lane vs 1 / 3 / 4 lanes on the network forward, L b64 and M b32, one box (final r06 tree: 12.15 / 11.18 / 12.16 / 11.84 ms and 4.96 / 4.37 / 4.60 / 4.68 ms).   python tools/lanes_probe.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from head_detector_amd.engine import VGHeadsEngine
dev = torch.device("cuda", 0)
for variant, B in (("vgg_heads_l", 64), ("vgg_heads_m", 32)):
    eng = VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1)
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    for rnd in range(2):
        for ns in (1, 2, 3, 4):
            eng.set_split(ns)
            for _ in range(8):
                eng.forward_net(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(eng.stream)
            for _ in range(40):
                eng.forward_net(x)
            e1.record(eng.stream)
            torch.cuda.synchronize()
            print(variant, B, "lanes", ns, f"{e0.elapsed_time(e1) / 40:.3f} ms", flush=True)
    eng.close()
