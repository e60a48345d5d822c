#!/usr/bin/env python3
"""Single-shape conv micro-benchmark through the C ABI (vgh_conv2d): A/B of tile configurations, and the
command rocprofv3 --pmc wraps for counter collection.
  python tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs all --iters 50"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default=["32,80,80,128,128,3,1"], nargs="+", help="B,H,W,Cin,Cout,k,stride (several allowed)")
    ap.add_argument("--cfgs", default="all", help="all | comma-separated indices or names")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--max-blocks", type=int, default=0, help="vgh_conv_set_max_blocks_per_xcd (32 = one workgroup per CU)")
    args = ap.parse_args()
    lib = _lib.load()
    _lib.check(lib.vgh_conv_set_max_blocks_per_xcd(args.max_blocks))
    for shape in args.shape:
        run_shape(lib, args, shape)


def run_shape(lib, args, shape):
    B, H, W, Cin, Cout, k, stride = map(int, shape.split(","))
    dev = torch.device("cuda", 0)
    print(f"== shape {shape}{' +res' if args.res else ''}")
    rp = (Cout + 31) // 32 * 32
    w = (np.random.default_rng(0).standard_normal((rp, k, k, Cin)) * 0.05).astype(np.float32)
    pack = np.zeros(w.size, dtype=np.uint16)
    _lib.check(lib.vgh_pack_conv_weights(_lib.ptr(w), rp, k, Cin, _lib.ptr(pack)))
    d_pack = torch.from_numpy(pack.view(np.int16)).to(dev)
    d_bias = torch.zeros(rp, device=dev)
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(B, Ho, Wo, rp, device=dev, dtype=torch.bfloat16)
    res = torch.randn(B, Ho, Wo, rp, device=dev).to(torch.bfloat16) if args.res else None
    names = [lib.vgh_conv_cfg_name(i).decode() for i in range(lib.vgh_conv_num_cfgs())]
    cfgs = range(len(names)) if args.cfgs == "all" else [int(c) if c.lstrip("-").isdigit() else names.index(c) for c in args.cfgs.split(",") if c.lstrip("-").isdigit() or c in names]
    flops = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    st = torch.cuda.current_stream().cuda_stream
    for c in cfgs:
        if c >= 0 and not lib.vgh_conv_cfg_ok(c, k, stride, rp, 1, 0):
            continue
        call = _lib.ConvCall(in_dev=x.data_ptr(), in_pitch=Cin, in_coff=0, cin=Cin, B=B, H=H, W=W, wpack_dev=d_pack.data_ptr(), bias_dev=d_bias.data_ptr(),
                             out_dev=out.data_ptr(), out_pitch=rp, out_coff=0, cout_pad=rp, cout_store=rp, out_split=rp, out_coff2=0, out_f32=0,
                             res_dev=res.data_ptr() if res is not None else None, res_pitch=rp, res_coff=0, alpha=0.5, ksize=k, stride=stride, act=1, shuffle=0, force_cfg=c)
        for _ in range(3):
            _lib.check(lib.vgh_conv2d(C.byref(call), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            _lib.check(lib.vgh_conv2d(C.byref(call), st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print(f"cfg {c:2d} {names[c] if c >= 0 else 'auto':24s} {ms:.4f} ms  {flops / ms / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
