#!/usr/bin/env python3
"""FLAME decode alone (SURVEY 8(d) second headline metric): us per head at n in {1, 8, 64, 96, 1024, 8192}, through the C ABI
(vgh_flame_decode via FLAMELayer.decode), synthetic constants seed 3, live coefficients of the M / L heads."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402
from head_detector_amd.flame import FLAMELayer  # noqa: E402
from head_detector_amd.synthetic import synthetic_flame_model  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    fl = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=8192)
    lib = _lib.load()
    # vgh_flame_set_matrix_path: 0 VALU, 1 automatic, 2 register-fed MFMA, 3 / 4 / 5 LDS-staged MFMA, 6 / 7 / 8 component-split tiles; FLAME_MODES = several in one process
    modes = tuple(int(x) for x in os.environ.get("FLAME_MODES", os.environ.get("FLAME_MODE", "1")).split(","))
    ns = tuple(int(x) for x in os.environ.get("FLAME_NS", "1,8,64,96,256,512,1024,8192").split(","))
    lives = {"m": ("M heads 64+32", (64, 32)), "l": ("L heads 128+64", (128, 64)), "all": ("all 300+100", (300, 100))}
    rows = []
    for mode, lk in ((m, k) for k in os.environ.get("FLAME_LIVES", "m,l,all").split(",") for m in modes):
        live, (sl, el) = lives[lk]
        _lib.check(lib.vgh_flame_set_matrix_path(mode))
        for n in ns:
            if mode == 7 and n > 8:
                continue  # modes 8 / 9 differ from 6 / 7 up to 8 heads only
            p = torch.randn(n, 413, device=dev)
            p[:, sl:300] = 0
            p[:, 300 + el:400] = 0
            unpad = torch.tensor([[3.0, 4.0, 1.25]], device=dev).expand(n, 3).contiguous()
            # straight through the C ABI with preallocated outputs: the facade's torch.empty + ctypes marshalling (~20 us of host
            # time per call) would otherwise be what the events see at small n
            proj = torch.empty(n, fl.num_vertices, 3, device=dev)
            rot = torch.empty(n, 3, 3, device=dev)
            h = fl._need_handle()
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: _lib.check(lib.vgh_flame_decode(h, p.data_ptr(), n, sl, el, unpad.data_ptr(), None, rot.data_ptr(), proj.data_ptr(), st))  # noqa: E731
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            it = 200 if n <= 1024 else 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / it
            k = sl + el + 36
            rows.append(dict(mode=mode, live=live, n=n, us_per_call=round(us, 2), us_per_head=round(us / n, 3), gflops=round(n * (2.0 * k * 15069 + 0.12e6 + 0.8e6) / us / 1e3, 1),
                             out_GBps=round(n * 60276 / us / 1e3, 1)))
            print(rows[-1])
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
