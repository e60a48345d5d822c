#!/usr/bin/env python3
"""Phase timeline of the persistent t tile (csrc/ds_b2b.hip; s_memtime marks of the -DVGH_EXPERIMENTS build):
   VGH_LIB_PATH=head_detector_amd/libvgh_exp.so python tools/ds_trace.py [variant batch]
marks per (workgroup, wave, tile): 0 tile top, 1 own patch pieces landed, 2 barrier passed, 3 next patch issued, 4 K loop + first epilogue done, 5 exchange written (after the
second barrier), 6 third barrier passed, 7 second GEMM + stores issued."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402

NAMES = ["wait own patch", "barrier 1", "issue next patch", "K loop + epilogue 1", "barrier 2 + exchange write", "barrier 3", "GEMM 2 + epilogue 2 + stores", "loop tail"]


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "vgg_heads_l"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    lib = _lib.load()
    lib.vgh_conv_set_trace.restype = C.c_int
    lib.vgh_conv_set_trace.argtypes = [C.c_void_p]
    dev = torch.device("cuda", 0)
    eng = VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1)
    eng.set_split(1)
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    for _ in range(3):
        eng.forward_net(x)
    torch.cuda.synchronize()
    rows_other, nblk = 8192 * 16 * 4, 512
    tr = torch.zeros(rows_other + nblk * 3 * 8 * 8, dtype=torch.int64, device=dev)
    lib.vgh_conv_set_trace(tr.data_ptr())
    for _ in range(3):
        eng.forward_net(x)
    torch.cuda.synchronize()
    lib.vgh_conv_set_trace(None)
    t = tr[rows_other:].cpu().numpy().reshape(nblk, 3, 8, 8).astype(np.int64)
    ok = (t[:, :, :, 0] > 0) & (t[:, :, :, 7] > 0)
    print(f"{variant} b{B}: {int(ok[:, 0, :].sum())} (workgroup, tile) records; s_memtime ticks = 10 ns")
    for tile_sel, label in ((slice(0, 1), "first tile of a workgroup"), (slice(2, 8), "tiles 2..7 (steady state)")):
        print(f"-- {label}")
        for k in range(7):
            d = (t[:, :, tile_sel, k + 1] - t[:, :, tile_sel, k])[ok[:, :, tile_sel]]
            print(f"   {NAMES[k]:32s} mean {d.mean() * 10:8.0f} ns   p10 {np.percentile(d, 10) * 10:8.0f}   p90 {np.percentile(d, 90) * 10:8.0f}")
        d = (t[:, :, tile_sel, 7] - t[:, :, tile_sel, 0])[ok[:, :, tile_sel]]
        print(f"   {'tile total':32s} mean {d.mean() * 10:8.0f} ns")
    per = (t[:, :, 7, 7] - t[:, :, 1, 0])[ok[:, :, 7] & ok[:, :, 1]] / 7.0
    print(f"tile period (tiles 1..7): mean {per.mean() * 10:.0f} ns per tile per workgroup")
    eng.close()


if __name__ == "__main__":
    main()
