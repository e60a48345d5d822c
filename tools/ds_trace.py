#!/usr/bin/env python3
"""Phase timeline of the persistent t tile (csrc/ds_b2b.hip; s_memtime marks of the -DVGH_EXPERIMENTS build):
   VGH_LIB_PATH=head_detector_amd/libvgh_exp.so python tools/ds_trace.py [variant batch]
marks per (workgroup, wave, tile) -- compute waves 0-2: 0 tile top, 2 barrier passed, 4 K loop (with the previous tile's second GEMM / epilogue / stores in its slots) done,
5 first epilogue + exchange written, 7 loop tail; loader wave 3: 0 top, 1 patch k landed (vmcnt), 2 barrier passed, 3 patch k + 2 issued.  Ticks are shader-clock cycles / 10 (printed x 10)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402


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
    rows_other, nblk = 8192 * 16 * 4, 256
    tr = torch.zeros(rows_other + nblk * 4 * 8 * 8, dtype=torch.int64, device=dev)
    lib.vgh_conv_set_trace(tr.data_ptr())
    for _ in range(3):
        eng.forward_net(x)
    torch.cuda.synchronize()
    lib.vgh_conv_set_trace(None)
    t = tr[rows_other:].cpu().numpy().reshape(nblk, 4, 8, 8).astype(np.int64)
    comp, load = t[:, :3], t[:, 3]
    okc = (comp[..., 0] > 0) & (comp[..., 7] > 0)
    okl = (load[..., 0] > 0) & (load[..., 3] > 0)
    print(f"{variant} b{B}: {int(okc[:, 0].sum())} (workgroup, tile) records; numbers are shader-clock cycles")

    def stat(name, d):
        d = d * 10.0 if False else d.astype(np.float64)
        print(f"   {name:44s} mean {d.mean():8.0f}   p10 {np.percentile(d, 10):8.0f}   p90 {np.percentile(d, 90):8.0f}")

    sel = slice(2, 8)
    m = okc[:, :, sel]
    print("-- compute waves, tiles 2..7")
    stat("barrier (patch k + exchange k - 1)", (comp[:, :, sel, 2] - comp[:, :, sel, 0])[m])
    stat("K loop k with GEMM 2 / epilogue / stores k - 1", (comp[:, :, sel, 4] - comp[:, :, sel, 2])[m])
    stat("epilogue 1 + exchange write", (comp[:, :, sel, 5] - comp[:, :, sel, 4])[m])
    stat("loop tail", (comp[:, :, sel, 7] - comp[:, :, sel, 5])[m])
    stat("tile total", (comp[:, :, sel, 7] - comp[:, :, sel, 0])[m])
    ml = okl[:, sel]
    print("-- loader wave, tiles 2..7")
    stat("wait for patch k", (load[:, sel, 1] - load[:, sel, 0])[ml])
    stat("barrier", (load[:, sel, 2] - load[:, sel, 1])[ml])
    stat("issue patch k + 2 (34 pieces)", (load[:, sel, 3] - load[:, sel, 2])[ml])
    per = (comp[:, 0, 7, 7] - comp[:, 0, 1, 0])[okc[:, 0, 7] & okc[:, 0, 1]] / 7.0
    tot = (comp[:, 0, 7, 6] - comp[:, 0, 0, 0])[okc[:, 0, 0] & (comp[:, 0, 7, 6] > 0)]
    print(f"workgroup lifetime: mean {tot.mean():.0f} ticks, tiles per workgroup {comp[:, 0, 7, 3].mean():.1f}; span of the launch {comp[:, :3, 7, 6].max() - comp[:, :3, 0, 0][comp[:, :3, 0, 0] > 0].min()} ticks")
    print(f"tile period (tiles 1..7): mean {per.mean():.0f} cycles per tile per workgroup")
    eng.close()


if __name__ == "__main__":
    main()
