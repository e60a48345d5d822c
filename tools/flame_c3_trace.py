#!/usr/bin/env python3
"""Phase stamps of the component-split FLAME kernel (experiments build, VGH_LIB_PATH=.../libvgh_exp.so): block 1's wave 0 -- kernel top, coefficient tile staged,
K loop done (incl. the pose barrier), exchange barrier, end -- next to the prologue's own stamps (head 0)."""
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
pn = ["params->LDS", "JS.beta+butterfly", "rodrigues", "pose feat+chain", "A pack", "lane-1 block"]
cn = ["staged", "K loop", "exchange", "epilogue"]
for mode in (int(x) for x in os.environ.get("FLAME_MODES", "6,7").split(",")):
    lib.vgh_flame_set_matrix_path(mode)
    for n in (int(x) for x in os.environ.get("FLAME_NS", "1,8,32,96").split(",")):
        if mode == 7 and n > 8:
            continue
        for sl, el in ((128, 64), (300, 100)):
            p = torch.randn(n, 413, device=dev)
            p[:, sl:300] = 0
            p[:, 300 + el:400] = 0
            tr.zero_()
            for _ in range(3):
                fl.decode(p, shape_live=sl, expr_live=el, want_vertices=False)
            torch.cuda.synchronize()
            t = tr.cpu().tolist()
            print(f"mode {mode} n={n} live {sl}+{el}: prologue " + " ".join(f"{pn[i]} {(t[i + 1] - t[i]) / 100:.2f}" for i in range(6)) + f" (total {(t[6] - t[0]) / 100:.2f} us)"
                  + " | c3 " + " ".join(f"{cn[i]} {(t[9 + i] - t[8 + i]) / 100:.2f}" for i in range(4)) + f" (total {(t[12] - t[8]) / 100:.2f} us; top - prologue start {(t[8] - t[0]) / 100:.2f} us; K loop {t[15] - t[14]} shader cycles = {(t[15] - t[14]) / max(1, (t[10] - t[9]) * 10):.2f} GHz)")
