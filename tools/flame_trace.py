#!/usr/bin/env python3
"""Phase stamps of the FLAME per-head prologue (experiments build: python -m head_detector_amd.build --experiments, VGH_LIB_PATH=.../libvgh_exp.so)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402
from head_detector_amd.flame import FLAMELayer  # noqa: E402
from head_detector_amd.synthetic import synthetic_flame_model  # noqa: E402

dev = torch.device("cuda", 0)
fl = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=2048)
lib = C.CDLL(_lib.LIB_PATH)
tr = torch.zeros(16, dtype=torch.int64, device=dev)
lib.vgh_flame_set_trace.argtypes = [C.c_void_p]
lib.vgh_flame_set_trace(tr.data_ptr())
names = ["start", "params->LDS", "JS.beta + butterfly", "rodrigues", "pose feat + chain", "A pack", "lane-1 block (6D, rpy)"]
for n in (1, 8, 1024):
    p = torch.randn(n, 413, device=dev)
    for _ in range(3):
        fl.decode(p, shape_live=128, expr_live=64)
    torch.cuda.synchronize()
    t = tr.cpu().tolist()
    print(f"n={n}: " + "  ".join(f"{names[i + 1]} {(t[i + 1] - t[i]) / 100:.2f} us" for i in range(6)) + f"  | total {(t[6] - t[0]) / 100:.2f} us")
