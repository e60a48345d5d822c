"""GPU (-m gpu): the single-plane fp16 throughput mode (VGH_FMT_F16, r05) -- the reference's own FP16 export
(yolo_head_training/yolo_head/exportable_mesh_model.py:177,299,409: weights and activations in half precision) on the MI355X engine:
one fp16 plane per value, v_mfma_f32_32x32x16_f16, per-op power-of-two weight prescale; same bytes and MFMA count as the bf16 mode.
It rides the fp16 split kernels with one K segment and no lo plane (csrc/conv_split.hip) plus the fp16 variant of the ping-pong 3x3 tiles
(csrc/conv_pp.hip).  Exact-operand parity per conv; the whole network against the unfused fp32 oracle with its level pinned."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
FMT_F16 = 5


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _run_conv_f16(lib, x, W, b, k, stride, act=1, res=None, alpha=0.0, out_f32=False, shuffle=False, cfg=-1, groups=None, in_coff=8, out_coff=8):
    """x [B,H,W,Ct] float, W [rows,k,k,Cin] float -> (engine output float, fp64 reference on the fp16-rounded operands the kernel sees, rows)."""
    from head_detector_amd import _lib

    B, H, Wd, Ct = x.shape
    rows, Cin = W.shape[0], W.shape[3]
    rp = (rows + 31) // 32 * 32
    Wp = torch.zeros(rp, k, k, Cin)
    Wp[:rows] = W
    bp = torch.zeros(rp)
    bp[:rows] = b
    pack = np.zeros(Wp.numel(), dtype=np.uint16)
    osc = C.c_float(0.0)
    _lib.check(lib.vgh_pack_conv_weights_split(_lib.ptr(np.ascontiguousarray(Wp.numpy())), rp, k, Cin, FMT_F16, _lib.ptr(pack), C.byref(osc)))
    scale = 1.0 / osc.value
    assert float(np.log2(scale)) % 1 == 0 and 512 <= float(Wp.abs().max()) * scale < 1024
    w_val = (Wp * scale).half().double() / scale  # what the image holds, in real units
    in_pitch = Ct + in_coff + 8
    xin = torch.zeros(B, H, Wd, in_pitch)
    xin[..., in_coff : in_coff + Ct] = x
    d_x = xin.half().to(_dev()).contiguous()
    x_val = x.half().double()
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (Wd + 2 * (k // 2) - k) // stride + 1
    oc = rp // 4 if shuffle else rp
    oh, ow = (2 * Ho, 2 * Wo) if shuffle else (Ho, Wo)
    out_pitch = oc + out_coff + 8
    d_out = torch.full((B, oh, ow, out_pitch), -768.0, dtype=torch.float32 if out_f32 else torch.float16, device=_dev())
    d_res = res.half().to(_dev()).contiguous() if res is not None else None
    d_pack = torch.from_numpy(pack.view(np.int16)).to(_dev())
    d_bias = bp.to(_dev())
    call = _lib.ConvCall(in_dev=d_x.data_ptr(), in_pitch=in_pitch, in_coff=in_coff, cin=Cin, B=B, H=H, W=Wd, wpack_dev=d_pack.data_ptr(), bias_dev=d_bias.data_ptr(), out_dev=d_out.data_ptr(),
                         out_pitch=out_pitch, out_coff=out_coff, cout_pad=rp, cout_store=rows, out_split=rp, out_coff2=0, out_f32=int(out_f32),
                         res_dev=d_res.data_ptr() if d_res is not None else None, res_pitch=res.shape[-1] if res is not None else 0, res_coff=0, alpha=alpha, ksize=k, stride=stride, act=act,
                         shuffle=int(shuffle), force_cfg=cfg, grp_cout=groups[0] if groups else 0, grp_in_stride=groups[1] if groups else 0, fmt=FMT_F16, out_scale=osc.value, out_fp8=0,
                         gscale_dev=None)
    _lib.check(lib.vgh_conv2d(C.byref(call), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    xr, wr = x_val.permute(0, 3, 1, 2), w_val.permute(0, 3, 1, 2)
    if groups:
        gc, gs = groups
        y = torch.cat([F.conv2d(xr[:, g * gs : g * gs + Cin], wr[g * gc : (g + 1) * gc], None, stride=stride, padding=k // 2) for g in range(rp // gc)], 1)
    else:
        y = F.conv2d(xr[:, :Cin], wr, None, stride=stride, padding=k // 2)
    y = y + bp.double()[None, :, None, None]
    if act == 1:
        y = torch.relu(y)
    y = y.permute(0, 2, 3, 1)
    if shuffle:
        Cc = rp // 4
        z = torch.zeros(B, oh, ow, Cc, dtype=torch.float64)
        for d in range(4):
            z[:, d // 2 :: 2, d % 2 :: 2] = y[..., d * Cc : (d + 1) * Cc]
        y = z
    if res is not None:
        y = y + alpha * res.half().double()[..., : y.shape[-1]]
    return d_out.float().cpu(), y.float(), rows if not shuffle else rp // 4, out_coff


def _close(out, ref, out_f32, where):
    tol = (2e-3 + 1e-4 * ref.abs()) if out_f32 else (1e-3 + 1.0 / 1024 * ref.abs())  # fp16 storage: one ulp (2^-11 relative) + accumulation-order slack
    bad = (out - ref).abs() > tol
    assert not bool(bad.any()), f"{where}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float((out - ref).abs().max()):.5f}, first {bad.nonzero()[:4].tolist()}"


F16_CASES = [
    # (B, H, W, Cin, Cout, k, stride)
    (1, 16, 16, 32, 32, 1, 1),
    (2, 40, 40, 64, 128, 3, 1),   # the fp16 ping-pong tiles take it when left to the library (map >= 40 a side)
    (1, 48, 56, 96, 96, 3, 1),
    (2, 40, 40, 64, 128, 3, 2),
    (1, 32, 32, 128, 256, 3, 1),
    (3, 20, 20, 192, 192, 1, 1),
    (1, 33, 17, 64, 96, 3, 2),
    (1, 5, 5, 32, 13, 1, 1),
]


@pytest.mark.parametrize("case", F16_CASES)
def test_conv_single_plane_fp16_every_tile_vs_exact_operand_reference(gpu_lib, case):
    B, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(hash(case) % 991)
    x = torch.randn(B, H, W, Cin, generator=g)
    Wt = torch.randn(Cout, k, k, Cin, generator=g) * (1.5 / np.sqrt(k * k * Cin)) * (1.0 + 0.5 * torch.arange(Cout).float()[:, None, None, None] / Cout)
    b = torch.randn(Cout, generator=g)
    rp = (Cout + 31) // 32 * 32
    names = [gpu_lib.vgh_conv_split_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_split_num_cfgs())]
    assert names[-3:] == ["g8x8x128_n8", "g8x8x96_n8", "g8x8x64_n8"]
    out_f32 = Cout % 4 != 0
    tested, pp = 0, 0
    for cfg in [-1] + list(range(len(names))):
        if cfg >= 0 and not gpu_lib.vgh_conv_split_cfg_ok(cfg, k, stride, rp, int(Cout % 8 == 0 and not out_f32), 0, 0):
            continue
        if cfg >= len(names) - 3 and (k != 3 or stride != 1 or rp != Cout):
            continue
        out, ref, st, o0 = _run_conv_f16(gpu_lib, x, Wt, b, k, stride, cfg=cfg, out_f32=out_f32)
        _close(out[..., o0 : o0 + st], ref[..., :st], out_f32, f"{case} cfg={names[cfg] if cfg >= 0 else 'auto'}")
        assert float((out[..., :o0] + 768.0).abs().max()) == 0.0 and float((out[..., o0 + st :] + 768.0).abs().max()) < 1.0, "wrote outside its channels"
        tested += 1
        pp += cfg >= len(names) - 3
    assert tested >= 3 and (pp >= 1 or not (k == 3 and stride == 1 and Cout % 64 == 0))


def test_conv_single_plane_fp16_epilogues(gpu_lib):
    """Residual (every 3x3 tile family incl. the ping-pong one, many tiles per workgroup), ConvTranspose pixel-shuffle, grouped launch, fp32 prediction output, saturation."""
    g = torch.Generator().manual_seed(5)
    names = [gpu_lib.vgh_conv_split_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_split_num_cfgs())]
    B, H, W, C0 = 3, 40, 56, 128
    x = torch.randn(B, H, W, 64, generator=g)
    Wt = torch.randn(C0, 3, 3, 64, generator=g) * (1.5 / np.sqrt(9 * 64)) * (1.0 + 0.5 * torch.arange(C0).float()[:, None, None, None] / C0)
    b = torch.randn(C0, generator=g)
    r = torch.randn(B, H, W, C0, generator=g)
    ran = 0
    try:
        for cap in (0, 2):
            assert gpu_lib.vgh_conv_set_max_blocks_per_xcd(cap) == 0
            for cfg, n in enumerate(names):
                if not gpu_lib.vgh_conv_split_cfg_ok(cfg, 3, 1, C0, 1, 0, 0) or not (n.startswith("sp") or n.startswith("g")):
                    continue
                out, ref, st, o0 = _run_conv_f16(gpu_lib, x, Wt, b, 3, 1, res=r, alpha=0.37, cfg=cfg)
                _close(out[..., o0 : o0 + st], ref, False, f"residual {n} cap={cap}")
                ran += 1
    finally:
        gpu_lib.vgh_conv_set_max_blocks_per_xcd(0)
    assert ran >= 8
    # ConvTranspose2d(k=2, s=2) as four pointwise GEMMs + pixel shuffle
    xt = torch.randn(2, 10, 10, 64, generator=g)
    Wg = torch.randn(4 * 32, 1, 1, 64, generator=g) * 0.2
    out, ref, st, o0 = _run_conv_f16(gpu_lib, xt, Wg, torch.randn(128, generator=g), 1, 1, act=0, shuffle=True)
    _close(out[..., o0 : o0 + st], ref, False, "shuffle")
    # grouped: four 32-channel branches in one launch
    xg = torch.randn(2, 24, 24, 128, generator=g)
    Wgr = torch.randn(128, 3, 3, 32, generator=g) * 0.1
    out, ref, st, o0 = _run_conv_f16(gpu_lib, xg, Wgr, torch.randn(128, generator=g), 3, 1, groups=(32, 32))
    _close(out[..., o0 : o0 + st], ref, False, "grouped")
    # saturation instead of infinities: an output beyond fp16's range is stored as +-65504 by every tile family
    xs = torch.full((1, 40, 40, 64), 200.0)
    Ws = torch.full((64, 3, 3, 64), 1.0)
    for cfg in (-1, next(i for i, n in enumerate(names) if n.startswith("sp16x16x64"))):
        out, ref, st, o0 = _run_conv_f16(gpu_lib, xs, Ws, torch.zeros(64), 3, 1, cfg=cfg)
        assert float(ref.max()) > 65504 and float(out[..., o0 : o0 + st].max()) == 65504.0 and bool(torch.isfinite(out[..., o0 : o0 + st]).all())


@pytest.mark.parametrize("variant,S,B", [("vgg_heads_m", 192, 2), ("vgg_heads_l", 320, 1)])
def test_fp16_network_every_op(gpu_lib, variant, S, B):
    """Every op of the fp16 program against an fp64 evaluation on the engine's own inputs (tests/program_ref.py), one fp16 ulp + accumulation slack; at 320 the L
    net's 80- and 40-wide maps run the fp16 ping-pong tiles."""
    import program_ref as pr
    from head_detector_amd import arch
    from head_detector_amd.engine import VGHeadsEngine

    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=7, precision="fp16")
    P = eng.program
    assert all(bf["is_f32"] in (arch.FMT_F16, arch.FMT_F32) for bf in P.bufs)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(2))
    eng.forward_net(x.to(_dev()))
    got = [eng.buffer(i, B).float().cpu() for i in range(len(P.bufs))]
    w_all, b_all = P.arrays()
    w_all = w_all.copy()
    for op in P.ops:  # the weights as the library stores them: per-op power-of-two prescale into [512, 1024), rounded to fp16 (vgh_pack_conv_weights_split, fmt VGH_FMT_F16)
        if op["kind"] != 1:
            continue
        sl = slice(op["w_off"], op["w_off"] + op["cout_pad"] * op["ksize"] ** 2 * op["cin"])
        w = torch.from_numpy(w_all[sl])
        mx = float(w.abs().max())
        sc = 2.0 ** (10 - int(np.frexp(mx)[1])) if mx > 0 else 1.0
        w_all[sl] = ((w * sc).half().float() / sc).numpy()
    for op in P.ops:
        if op["kind"] == 3:
            continue
        ob = op["out_buf"] if op["kind"] != 2 else op["in_buf"]
        exp = list(got)
        exp[ob] = got[ob].clone()
        pr.run_op(P, op, exp, x, False, w_all, b_all, f64=True)
        a, e = got[ob], exp[ob]
        tol = (2e-3 + 2e-4 * e.abs()) if P.bufs[ob]["is_f32"] == arch.FMT_F32 else (1e-3 + 1.0 / 1024 * e.abs())  # exact operands: one fp16 ulp + accumulation-order slack
        bad = (a - e).abs() > tol
        if bool(bad.any()):
            i = tuple(int(v) for v in ((a - e).abs() * bad).flatten().argmax().unsqueeze(0))
            flat = int(((a - e).abs() * bad).flatten().argmax())
            raise AssertionError(f"{variant} S={S} op {op['name']} (k{op['ksize']} s{op['stride']} cin {op['cin']} cout {op['cout_pad']}): {int(bad.sum())} mismatches, worst got {float(a.flatten()[flat])} "
                                 f"want {float(e.flatten()[flat])} at flat index {flat} of shape {tuple(a.shape)}; max |e| {float(e.abs().max()):.2f}")
    eng.close()


@pytest.mark.parametrize("variant,okey,B", [("vgg_heads_m", "m", 2), ("vgg_heads_l", "l", 1)], ids=["m640", "l640"])
def test_fp16_mode_deviation_from_the_oracle_is_pinned_next_to_bf16(gpu_lib, flame_model, variant, okey, B):
    """The fp16 mode against the unfused fp32 oracle by the routine that measures every other mode, next to bf16 on the same images: r04's emulation predicted an
    8 x smaller deviation at the same cost; this is the measurement."""
    from test_gpu_split import network_vs_oracle

    rh = network_vs_oracle(variant, okey, "fp16", 640, B, flame_model)
    r1 = network_vs_oracle(variant, okey, "bf16", 640, B, flame_model)
    assert rh["kept_iou_min"] >= 0.97 and rh["kept_param_max_rel_err"] < 0.05 and rh["vertex_l2_metric_max"] < 3e-3 and rh["dense_score_max_abs_err"] < 1e-4, (rh, r1)
    assert rh["kept_param_max_rel_err"] < 0.5 * r1["kept_param_max_rel_err"] and rh["dense_iou_min"] > r1["dense_iou_min"], (rh, r1)
