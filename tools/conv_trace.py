#!/usr/bin/env python3
"""Per-tile phase timeline of the persistent 3x3 kernels (s_memtime marks compiled into the -DVGH_EXPERIMENTS build only):
   VGH_LIB_PATH=head_detector_amd/libvgh_exp.so python tools/conv_trace.py --shape 64,80,80,128,128,3,1 --cfgs p16x16x64_n4x1,q16x16x64_n4x1
r06: the implicit-GEMM kernel (one tile per block) carries the same four marks -- block top, first stage landed, K loop done, last store issued:
   ... conv_trace.py --shape 64,320,320,64,96,3,2 --cfgs 128x96_w32x96_k1,128x96_w32x96_k1_r4      (the stage-1 downsample)"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402

TILES, MARKS = 16, 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,80,80,128,128,3,1")
    ap.add_argument("--cfgs", default="p16x16x64_n4x1,q16x16x64_n4x1")
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--burst", type=int, default=40)
    ap.add_argument("--pitch", type=int, default=0, help="channel pitch of the input tensor (>= Cin; default Cin): the K windows then use Cin / pitch of every pixel's bytes")
    args = ap.parse_args()
    lib = _lib.load()
    lib.vgh_conv_set_trace.restype = C.c_int
    lib.vgh_conv_set_trace.argtypes = [C.c_void_p]
    B, H, W, Cin, Cout, k, stride = map(int, args.shape.split(","))
    dev = torch.device("cuda", 0)
    rp = (Cout + 31) // 32 * 32
    w = (np.random.default_rng(0).standard_normal((rp, k, k, Cin)) * 0.05).astype(np.float32)
    pack = np.zeros(w.size, dtype=np.uint16)
    _lib.check(lib.vgh_pack_conv_weights(_lib.ptr(w), rp, k, Cin, _lib.ptr(pack)))
    d_pack = torch.from_numpy(pack.view(np.int16)).to(dev)
    d_bias = torch.zeros(rp, device=dev)
    pitch = max(args.pitch, Cin)
    x = torch.randn(B, H, W, pitch, device=dev).to(torch.bfloat16)
    out = torch.empty(B, H, W, rp, device=dev, dtype=torch.bfloat16)
    res = torch.randn(B, H, W, rp, device=dev).to(torch.bfloat16) if args.res else None
    names = [lib.vgh_conv_cfg_name(i).decode() for i in range(lib.vgh_conv_num_cfgs())]
    st = torch.cuda.current_stream().cuda_stream
    for name in args.cfgs.split(","):
        c = names.index(name)
        call = _lib.ConvCall(in_dev=x.data_ptr(), in_pitch=pitch, in_coff=0, cin=Cin, B=B, H=H, W=W, wpack_dev=d_pack.data_ptr(), bias_dev=d_bias.data_ptr(),
                             out_dev=out.data_ptr(), out_pitch=rp, out_coff=0, cout_pad=rp, cout_store=rp, out_split=rp, out_coff2=0, out_f32=0,
                             res_dev=res.data_ptr() if res is not None else None, res_pitch=rp, res_coff=0, alpha=0.5, ksize=k, stride=stride, act=1, shuffle=0, force_cfg=c)
        lib.vgh_conv_set_trace(None)
        for _ in range(3):
            _lib.check(lib.vgh_conv2d(C.byref(call), st))
        torch.cuda.synchronize()
        tr = torch.zeros(8192 * TILES * MARKS, dtype=torch.int64, device=dev)
        # steady state: a burst of launches (the chip settles at its power-limited clock), the LAST one traced
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(args.burst):
            _lib.check(lib.vgh_conv2d(C.byref(call), st))
        e1.record()
        lib.vgh_conv_set_trace(tr.data_ptr())
        _lib.check(lib.vgh_conv2d(C.byref(call), st))
        e2.record()
        torch.cuda.synchronize()
        lib.vgh_conv_set_trace(None)
        t = tr.cpu().numpy().reshape(8192, TILES, MARKS)
        used = t[:, 0, 0] != 0
        xcd = (np.arange(8192) % 8)[used]
        t = t[used]
        nb = t.shape[0]
        ntile = (t[:, :, 3] != 0).sum(1)
        end = np.array([t[i, max(ntile[i] - 1, 0), 3] for i in range(nb)])
        span = max((end[xcd == x].max() - t[xcd == x, 0, 0].min()) for x in range(8) if (xcd == x).any())  # s_memtime bases differ per XCD
        us_burst, us_last = e0.elapsed_time(e1) * 1e3 / max(args.burst, 1), e1.elapsed_time(e2) * 1e3
        print(f"== {name}: {nb} blocks, tiles/block {ntile.min()}..{ntile.max()}; traced launch {us_last:.1f} us (burst mean {us_burst:.1f} us); longest XCD span {span} ticks "
              f"-> {span / us_last / 1e3:.2f} ticks/ns")
        for k_ in range(int(ntile.max())):
            sel = ntile > k_
            tt = t[sel, k_]
            d01, d12, d23 = tt[:, 1] - tt[:, 0], tt[:, 2] - tt[:, 1], tt[:, 3] - tt[:, 2]
            gap = (t[sel & (ntile > k_ + 1), k_ + 1, 0] - t[sel & (ntile > k_ + 1), k_, 3]) if k_ + 1 < ntile.max() else np.array([0])
            print(f"  tile {k_:2d} ({int(sel.sum()):4d} blocks): top->ready {d01.mean():7.0f} (max {d01.max():6d})  K loop {d12.mean():7.0f} (min {d12.min():6d} max {d12.max():6d})  "
                  f"epilogue {d23.mean():7.0f} (max {d23.max():6d})  ->next {gap.mean():6.0f}")
        # blocks in flight: the sum of block lifetimes over the busiest XCD's span / its 32 CUs (one-tile-per-block kernels: how many tiles a CU holds at a time)
        life = np.array([end[i] - t[i, 0, 0] for i in range(nb)], dtype=np.float64)
        per_x = [(life[xcd == x].sum() / max(float(end[xcd == x].max() - t[xcd == x, 0, 0].min()), 1.0)) for x in range(8) if (xcd == x).any()]
        print(f"  traced blocks alive at a time: {np.mean(per_x):.1f} per XCD = {np.mean(per_x) / 32:.2f} per CU (of the {nb} traced blocks; a launch with more than 8192 blocks is traced for its first 8192)")
        # s_memtime ticks are shader cycles (tools/micro/barrier_handoff: 1024.1 ticks per 32 back-to-back 32x32x16 MFMAs per SIMD): the busiest blocks' tick count
        # over the launch's wall time is the clock the kernel actually ran at; for the g / h tiles (marks: tile top, first channel block done, K loop done,
        # epilogue done) the steady K loop gives cycles per barrier slot against the 128 * BC / 32 matrix-pipe cycles a slot holds
        nmax = int(ntile.max())
        blk = t[ntile == nmax]
        cyc = float((blk[:, nmax - 1, 3] - blk[:, 0, 0]).mean())
        line = f"  busiest blocks ({blk.shape[0]}, {nmax} tiles): {cyc:.0f} cycles tile top -> last epilogue = {cyc / us_last / 1e3:.3f} GHz effective over the traced launch"
        if name[0] in "gh" and Cin > 32 and k == 3:
            bc = int(name.split("x")[2].split("_")[0])
            slots = 18 * (Cin // 32 - 1)
            per_slot = float((t[:, 0, 2] - t[:, 0, 1]).mean()) / slots
            tile = float((t[ntile > 1][:, 1, 0] - t[ntile > 1][:, 0, 0]).mean()) if (ntile > 1).any() else cyc
            line += (f"; steady K loop {per_slot:.0f} cycles / slot vs {4 * bc} of MFMA = {4 * bc / per_slot:.2f} busy; whole tile {tile:.0f} cycles, "
                     f"{2 * 9 * (Cin // 32) * 4 * bc / tile:.2f} busy")
        print(line)


if __name__ == "__main__":
    main()
