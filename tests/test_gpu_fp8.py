"""e4m3 links of the "fp8" throughput mode (csrc/conv_pp.hip, r05).

Exact-operand parity, like the bf16 conv tests: the kernel's operands are e4m3 values, so the reference is an fp64 convolution of
EXACTLY those values (dequantised), and what is left to tolerate is fp32 accumulation order and the one rounding of the output
format.  The reference's own exporter offers quantised / half-precision exports of this network
(yolo_head_training/yolo_head/exportable_mesh_model.py:175-178,398-411); the mode is never the headline (BASELINE: bf16)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

E4M3 = torch.float8_e4m3fn


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _unswizzle(pack: np.ndarray, rp: int, k: int, cin: int) -> np.ndarray:
    """e4m3 weight image [tap * cb64 + cb][cout][64 B, 16-byte chunks at slot = chunk ^ ((cout >> 2) & 3)] -> codes [rp][k][k][cin]."""
    cb64 = cin // 64
    img = pack.reshape(k * k * cb64, rp, 4, 16)
    out = np.zeros((rp, k * k, cin), dtype=np.uint8)
    for co in range(rp):
        sw = (co >> 2) & 3
        for chunk in range(4):
            out[co].reshape(k * k, cb64, 4, 16)[:, :, chunk, :] = img[:, co, chunk ^ sw, :].reshape(k * k, cb64, 16)
    return out.reshape(rp, k, k, cin)


def _pack_fp8(lib, Wp: torch.Tensor):
    from head_detector_amd import _lib

    rp, k, _, cin = Wp.shape
    pack = np.zeros(Wp.numel(), dtype=np.uint8)
    ws = np.zeros(rp, dtype=np.float32)
    w_np = np.ascontiguousarray(Wp.numpy())
    _lib.check(lib.vgh_pack_conv_weights_fp8(_lib.ptr(w_np), rp, k, cin, _lib.ptr(pack), _lib.ptr(ws)))
    return pack, torch.from_numpy(ws)


def test_e4m3_weight_image_is_torch_float8_rounding_with_power_of_two_scales():
    """Host side (no GPU): vgh_pack_conv_weights_fp8 = one power-of-two scale per cout that maps the row's largest weight into (224, 448], every weight rounded ONCE
    to nearest-even e4m3 (bit-identical to torch.float8_e4m3fn), chunks swizzled like the bf16 image; zero rows, subnormals, ties and the 448 edge included."""
    from head_detector_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    rp, k, cin = 64, 3, 128
    W = torch.randn(rp, k, k, cin, generator=g) * torch.logspace(-6, 2, rp)[:, None, None, None]
    W[5] = 0.0  # an all-zero (padding) row
    W[6, 0, 0, :8] = torch.tensor([448.0, -448.0, 464.0, 447.9, 2.0 ** -9, 1.5 * 2.0 ** -9, 0.5 * 2.0 ** -9, -0.0])  # scale 2: ties and subnormals of the scaled grid
    W[7] = 1.0  # exact powers of two: scale 2^-8
    pack, ws = _pack_fp8(lib, W)
    mx = W.abs().flatten(1).max(1).values
    assert float(ws[5]) == 1.0 and bool(((ws.log2() % 1) == 0).all())
    live = mx > 0
    assert bool((mx[live] / ws[live] <= 448).all()) and bool((mx[live] / ws[live] > 224).all())
    codes = torch.from_numpy(_unswizzle(pack, rp, k, cin))
    want = (W / ws[:, None, None, None]).to(E4M3).view(torch.uint8)
    assert torch.equal(codes, want), int((codes != want).sum())
    assert float(ws[7]) == 2.0 ** -8 and int(codes[7, 0, 0, 0]) == 0x78  # 256 = 1.0 * 2^8


FP8_CASES = [
    # (B, H, W, Cin, Cout, in_fp8, out_fp8, res, act)
    (2, 24, 24, 64, 128, True, False, False, 1),
    (3, 40, 40, 128, 96, True, False, True, 1),  # the bottleneck's cv2: e4m3 in, bf16 out + bf16 residual
    (2, 20, 20, 192, 64, True, False, True, 0),  # ragged 8x8 sub-patches, three channel blocks, no activation
    (2, 24, 40, 64, 128, False, True, False, 1),  # the bottleneck's cv1: bf16 in, e4m3 out
    (1, 33, 17, 96, 96, False, True, False, 0),
    (2, 40, 24, 128, 256, True, True, False, 1),  # e4m3 in and out (two cout tiles per pixel group)
    (3, 16, 16, 256, 192, True, True, False, 1),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FP8_CASES)
def test_conv_e4m3_links_vs_exact_operand_reference(gpu_lib, case):
    from head_detector_amd import _lib

    B, H, Wd, Cin, Cout, in8, out8, with_res, act = case
    g = torch.Generator().manual_seed(hash(case) % 997)
    x = torch.randn(B, H, Wd, Cin, generator=g) * (1.0 + torch.arange(Cin).float() / Cin)  # asymmetric over channels
    Wt = torch.randn(Cout, 3, 3, Cin, generator=g) * (1.5 / np.sqrt(9 * Cin)) * (1.0 + 0.5 * torch.arange(Cout).float()[:, None, None, None] / Cout)
    Wt[Cout // 2] *= 1e-3  # a row with a much smaller norm: per-cout scales keep its precision
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, H, Wd, Cout, generator=g).to(torch.bfloat16).float() if with_res else None
    alpha = 0.37 if with_res else 0.0
    names = [gpu_lib.vgh_conv_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_num_cfgs())]
    in_coff, out_coff = 16, 16
    for cfg in [-1] + [i for i, n in enumerate(names) if n[0] == "g" and Cout % gpu_lib.vgh_conv_cfg_cout_tile(i) == 0]:
        for cap in (0, 2):
            # ---- operands exactly as the kernel will see them
            if in8:
                s_in = float(x.abs().max()) / 400.0
                xq = (x / s_in).to(E4M3)
                x_val = xq.float() * s_in
                d_x = torch.zeros(B, H, Wd, Cin + in_coff + 16, dtype=torch.uint8)
                d_x[..., in_coff : in_coff + Cin] = xq.view(torch.uint8)
                d_x = d_x.to(_dev())
                pack, ws = _pack_fp8(gpu_lib, Wt)
                w_val = (Wt / ws[:, None, None, None]).to(E4M3).float() * ws[:, None, None, None]
                d_pack = torch.from_numpy(pack).to(_dev())
                unit = ws * s_in
            else:
                s_in = 1.0
                x_val = x.to(torch.bfloat16).float()
                d_x = torch.zeros(B, H, Wd, Cin + in_coff + 16)
                d_x[..., in_coff : in_coff + Cin] = x_val
                d_x = d_x.to(torch.bfloat16).to(_dev())
                pk = np.zeros(Wt.numel(), dtype=np.uint16)
                _lib.check(gpu_lib.vgh_pack_conv_weights(_lib.ptr(np.ascontiguousarray(Wt.numpy())), Cout, 3, Cin, _lib.ptr(pk)))
                d_pack = torch.from_numpy(pk.view(np.int16)).to(_dev())
                w_val = Wt.to(torch.bfloat16).float()
                unit = torch.ones(Cout)
            y = F.conv2d(x_val.double().permute(0, 3, 1, 2), w_val.double().permute(0, 3, 1, 2), None, padding=1) + b.double()[None, :, None, None]
            if act == 1:
                y = torch.relu(y)
            y = y.permute(0, 2, 3, 1)
            if res is not None:
                y = y + alpha * res.double()
            s_out = float(y.abs().max()) / 300.0 if out8 else 1.0
            gscale = (unit / s_out).to(_dev())
            d_bias = (b / unit).to(_dev())
            out_pitch = Cout + out_coff + 16
            if out8:
                d_out = torch.full((B, H, Wd, out_pitch), 0x5A, dtype=torch.uint8, device=_dev())
            else:
                d_out = torch.full((B, H, Wd, out_pitch), -768.0, dtype=torch.bfloat16, device=_dev())
            d_res = res.to(torch.bfloat16).to(_dev()).contiguous() if res is not None else None
            call = _lib.ConvCall(in_dev=d_x.data_ptr(), in_pitch=d_x.shape[-1], in_coff=in_coff, cin=Cin, B=B, H=H, W=Wd, wpack_dev=d_pack.data_ptr(), bias_dev=d_bias.data_ptr(),
                                 out_dev=d_out.data_ptr(), out_pitch=out_pitch, out_coff=out_coff, cout_pad=Cout, cout_store=Cout, out_split=Cout, out_coff2=0, out_f32=0,
                                 res_dev=d_res.data_ptr() if d_res is not None else None, res_pitch=Cout if d_res is not None else 0, res_coff=0, alpha=alpha, ksize=3, stride=1, act=act,
                                 shuffle=0, force_cfg=cfg, fmt=_lib.VGH_FMT_FP8 if in8 else _lib.VGH_FMT_BF16, out_fp8=int(out8), gscale_dev=gscale.data_ptr())
            try:
                assert gpu_lib.vgh_conv_set_max_blocks_per_xcd(cap) == 0
                _lib.check(gpu_lib.vgh_conv2d(C.byref(call), torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
            finally:
                gpu_lib.vgh_conv_set_max_blocks_per_xcd(0)
            where = f"{case} cfg={names[cfg] if cfg >= 0 else 'auto'} cap={cap}"
            if out8:
                raw = d_out.cpu()
                assert bool((raw[..., :out_coff] == 0x5A).all()) and bool((raw[..., out_coff + Cout :] == 0x5A).all()), f"{where}: wrote outside its channels"
                got = raw[..., out_coff : out_coff + Cout].contiguous().view(E4M3).float()
                lo = 0.0 if act == 1 else -448.0
                want = (y / s_out).clamp(lo, 448.0).float().to(E4M3).float()
                assert bool(torch.isfinite(got).all()), where
                diff = (got - want).abs()
                step = torch.maximum(want.abs(), got.abs()) * 0.125 + 2.0 ** -9  # one e4m3 step at that magnitude
                assert bool((diff <= step).all()), f"{where}: off by more than one e4m3 step at {int((diff > step).sum())} outputs, max {float(diff.max())}"
                # a rounding boundary crossed by the fp32 accumulation order is rare
                assert float((diff > 0).float().mean()) < 0.02, f"{where}: {float((diff > 0).float().mean()):.4f} of the outputs differ from the exact-operand rounding"
            else:
                out = d_out.float().cpu()
                ref = y.float()
                tol = 1e-2 + 1.0 / 128 * ref.abs()  # one bf16 ulp + accumulation-order slack (as test_conv_all_configs_vs_torch)
                bad = (out[..., out_coff : out_coff + Cout] - ref).abs() > tol
                assert not bool(bad.any()), f"{where}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float((out[..., out_coff:out_coff + Cout] - ref).abs().max()):.4f}, first {bad.nonzero()[:4].tolist()}"
                assert float((out[..., :out_coff] + 768.0).abs().max()) == 0.0 and float((out[..., out_coff + Cout :] + 768.0).abs().max()) == 0.0, f"{where}: wrote outside its channels"


@pytest.mark.gpu
def test_e4m3_conv_rejects_what_the_tiles_cannot_do(gpu_lib):
    """Loud failures instead of wrong bytes: a residual with an e4m3 output, cin not a multiple of 64, a non-g tile, a missing factor vector."""
    from head_detector_amd import _lib

    d = _dev()
    x = torch.zeros(1, 16, 16, 64, dtype=torch.uint8, device=d)
    w = torch.zeros(64 * 9 * 64, dtype=torch.uint8, device=d)
    bias = torch.zeros(64, device=d)
    gs = torch.ones(64, device=d)
    out = torch.zeros(1, 16, 16, 64, dtype=torch.bfloat16, device=d)

    def call(**kw):
        base = dict(in_dev=x.data_ptr(), in_pitch=64, in_coff=0, cin=64, B=1, H=16, W=16, wpack_dev=w.data_ptr(), bias_dev=bias.data_ptr(), out_dev=out.data_ptr(), out_pitch=64, out_coff=0,
                    cout_pad=64, cout_store=64, out_split=64, out_coff2=0, out_f32=0, res_dev=None, res_pitch=0, res_coff=0, alpha=0.0, ksize=3, stride=1, act=1, shuffle=0, force_cfg=-1,
                    fmt=_lib.VGH_FMT_FP8, out_fp8=0, gscale_dev=gs.data_ptr())
        base.update(kw)
        return gpu_lib.vgh_conv2d(C.byref(_lib.ConvCall(**base)), torch.cuda.current_stream().cuda_stream)

    assert call() == 0
    assert call(gscale_dev=None) != 0 and b"gscale" in gpu_lib.vgh_last_error()
    assert call(cin=32) != 0
    assert call(ksize=1) != 0
    assert call(out_fp8=1, res_dev=out.data_ptr(), res_pitch=64) != 0 and b"residual" in gpu_lib.vgh_last_error()
    names = [gpu_lib.vgh_conv_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_num_cfgs())]
    assert call(force_cfg=next(i for i, n in enumerate(names) if n[0] == "p" and gpu_lib.vgh_conv_cfg_cout_tile(i) == 64)) != 0
    torch.cuda.synchronize()


def _pow2_row_scales(W: torch.Tensor) -> torch.Tensor:
    """The per-cout scales of vgh_pack_conv_weights_fp8, restated: the smallest power of two s with max|w[c]| / s <= 448 (1 for a zero row)."""
    mx = W.abs().flatten(1).max(1).values.double()
    e = torch.ceil(torch.log2(mx / 448.0))
    s = torch.where(mx > 0, torch.pow(torch.tensor(2.0, dtype=torch.float64), e), torch.ones_like(mx))
    return s.float()


@pytest.mark.gpu
@pytest.mark.parametrize("variant,S,B", [("vgg_heads_m", 192, 2), ("vgg_heads_l", 128, 2)])
def test_fp8_network_every_linked_op(gpu_lib, variant, S, B):
    """The "fp8" program end to end through vgh_net_create: every op that reads or writes an e4m3 link against an fp64 evaluation on the engine's OWN input
    buffer (decoded e4m3 values x the link's scale) with the weights quantised as the library quantises them -- checks the scales, the bias in accumulator
    units, the 64-channel K blocks of a 96-channel link and the separate shape | expression link buffer of the heads; every other op is bit-identical to the
    bf16 engine's (same inputs up to the first link)."""
    from head_detector_amd import arch
    from head_detector_amd.engine import VGHeadsEngine

    sd = arch.random_state_dict(variant, 31)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(2)).to(_dev())
    eng = VGHeadsEngine(variant, state_dict=sd, image_size=S, max_batch=B, precision="fp8", calib_images=x, fp8_min_px=8)
    P = eng.program
    links = [i for i, bf in enumerate(P.bufs) if bf["is_f32"] == arch.FMT_FP8]
    assert len(links) >= 10 and all(eng.fp8_scales[P.bufs[i]["name"]] > 0 for i in links)
    eng.forward_net(x)
    eng.stream.synchronize()
    w_all, b_all = P.arrays()
    checked = 0
    for op in P.ops:
        if not arch.op_touches_fp8(P, op):
            continue
        ib, ob = P.bufs[op["in_buf"]], P.bufs[op["out_buf"]]
        xin = eng.buffer(op["in_buf"], B).float().cpu()[..., op["in_coff"] : op["in_coff"] + op["cin"]]
        W = torch.from_numpy(w_all[op["w_off"] : op["w_off"] + op["cout_pad"] * 9 * op["cin"]].reshape(op["cout_pad"], 3, 3, op["cin"]).copy())
        bias = torch.from_numpy(b_all[op["b_off"] : op["b_off"] + op["cout_pad"]].copy())
        if ib["is_f32"] == arch.FMT_FP8:
            ws = _pow2_row_scales(W)
            Wv = (W / ws[:, None, None, None]).to(E4M3).float() * ws[:, None, None, None]
        else:
            Wv = W.to(torch.bfloat16).float()
        y = F.conv2d(xin.double().permute(0, 3, 1, 2), Wv.double().permute(0, 3, 1, 2), None, padding=1) + bias.double()[None, :, None, None]
        if op["act"] == 1:
            y = torch.relu(y)
        y = y.permute(0, 2, 3, 1)[..., : op["cout_store"]]
        if op["res_buf"] >= 0:
            y = y + op["alpha"] * eng.buffer(op["res_buf"], B).double().cpu()[..., op["res_coff"] : op["res_coff"] + op["cout_store"]]
        got = eng.buffer(op["out_buf"], B).float().cpu()[..., op["out_coff"] : op["out_coff"] + op["cout_store"]]
        if ob["is_f32"] == arch.FMT_FP8:
            sc = ob["scale"]
            want = ((y / sc).clamp(0.0 if op["act"] == 1 else -448.0, 448.0).float().to(E4M3).float() * sc)
            diff = (got - want).abs()
            step = torch.maximum(want.abs(), got.abs()) * 0.125 + sc * 2.0 ** -9
            assert bool((diff <= step).all()) and float((diff > 0).float().mean()) < 0.02, (op["name"], float(diff.max()), float((diff > 0).float().mean()))
            assert float(got.abs().max()) <= 448.0 * sc * 1.0001 and float(y.abs().max()) < 448.0 * sc, (op["name"], "the calibrated scale does not cover the tensor")
            if ob["pitch"] > ob["live"]:  # the 96-channel link: bytes 96 .. 127 of a pixel are never written and stay e4m3 +0
                assert float(eng.buffer(op["out_buf"], B)[..., ob["live"] :].abs().max()) == 0.0, op["name"]
        else:
            tol = 1e-2 + 1.0 / 128 * y.abs().float()
            assert bool(((got - y.float()).abs() <= tol).all()), (op["name"], float((got - y.float()).abs().max()))
        checked += 1
    n_mid = sum(P.bufs[i]["name"].split(".")[-1].startswith("mid") for i in links)
    assert checked == 2 * n_mid + 4 * (len(links) - n_mid)  # cv1 -> cv2 links: one writer, one reader; a head link carries the shape and the expression branch
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant,okey,B", [("vgg_heads_m", "m", 2), ("vgg_heads_l", "l", 1)], ids=["m640", "l640"])
def test_fp8_mode_deviation_from_the_oracle_is_pinned_next_to_bf16(gpu_lib, flame_model, variant, okey, B):
    """The "fp8" mode against the unfused fp32 oracle by the routine that measures every other mode (tests/test_gpu_split.py::network_vs_oracle), next to
    the bf16 mode on the same images: a throughput mode with its deviation on the record, never the headline (BASELINE.json: bf16)."""
    from test_gpu_split import network_vs_oracle

    r8 = network_vs_oracle(variant, okey, "fp8", 640, B, flame_model)
    r1 = network_vs_oracle(variant, okey, "bf16", 640, B, flame_model)
    # measured r05 (gpurun_out/parity_modes.jsonl): M kept IoU min 0.758 / params 0.19 / vertices 1.5e-2 m (bf16: 0.933 / 0.033 / 3.6e-3); L 0.932 / 0.69 / 2.4e-2 (bf16: 0.993 / 0.069 / 4.5e-3):
    # a 3-bit significand on 17 - 24 tensors of a random-weight network costs 5 - 10 x the bf16 mode's deviation.  Pinned with margin so that a regression cannot hide.
    assert r8["kept_iou_min"] >= 0.65 and r8["kept_param_max_rel_err"] < 1.2 and r8["vertex_l2_metric_max"] < 6e-2 and r8["dense_score_max_abs_err"] < 6e-3, (r8, r1)
    assert r8["kept_param_max_rel_err"] > r1["kept_param_max_rel_err"], "the e4m3 links cannot be more exact than the bf16 mode they replace: the comparison is broken"


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp8", "fp16", "int8"])
def test_r05_modes_through_the_pack_and_the_c_context(gpu_lib, flame_model, tmp_path, precision):
    """`pack --precision fp8 | fp16` -> vgh_create -> vgh_ctx_detect (a C client's path to the two r05 throughput modes) equals the Python engine of the same mode bit for
    bit: the pack (version 3) carries the e4m3 buffers with their calibrated scales / the fp16 buffers, the library quantises the weights itself, and the per-op tiles
    of the linked ops are the library's choice on both sides."""
    from head_detector_amd import _lib, arch, pack
    from head_detector_amd.engine import VGHeadsEngine, calibrate_fp8
    from head_detector_amd.flame import FLAMELayer

    variant, S, B = "vgg_heads_m", 320, 2
    sd = arch.random_state_dict(variant, 11)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).to(_dev())
    scales = calibrate_fp8(variant, sd, S, x) if precision in ("fp8", "int8") else None
    eng = VGHeadsEngine(variant, state_dict=sd, image_size=S, max_batch=B, use_tuning=False, precision=precision, fp8_scales=scales)
    P = eng.program
    if precision in ("fp8", "int8"):
        assert sum(bf["is_f32"] == arch.Q8_PRECISIONS[precision] for bf in P.bufs) >= 5  # 80- and 40-wide maps at 320
    pk = str(tmp_path / f"m_{precision}.vghpack")
    pack.write_pack(pk, P, flame_model, {}, B)
    hdr = pack.read_header(pk)
    assert hdr["version"] == 3 and hdr["precision"] == arch.PRECISION_FMT[precision]
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=B * 100)
    _, sc, _ = eng.model(x)
    conf = float(sc[:, 5, 0].min())
    ref = eng.detect(x, confidence_threshold=conf, flame=fl)
    n_ref = ref.num_heads
    assert n_ref >= B
    h = C.c_void_p()
    cfg = _lib.Config(device=torch.cuda.current_device(), pack_path=pk.encode(), max_batch=B)
    _lib.check(gpu_lib.vgh_create(C.byref(cfg), C.byref(h)))
    kk, V = 100, fl.num_vertices
    f32 = dict(dtype=torch.float32, device=_dev())
    ob, os_, of = torch.zeros(B, kk, 4, **f32), torch.zeros(B, kk, **f32), torch.zeros(B, kk, 413, **f32)
    oc, nh, hi = torch.zeros(B, dtype=torch.int32, device=_dev()), torch.zeros(1, dtype=torch.int32, device=_dev()), torch.zeros(B * kk, dtype=torch.int32, device=_dev())
    proj = torch.zeros(B * kk, V, 3, **f32)
    o = _lib.DetectOut(boxes_dev=ob.data_ptr(), scores_dev=os_.data_ptr(), flame_dev=of.data_ptr(), counts_dev=oc.data_ptr(), n_heads_dev=nh.data_ptr(), head_image_dev=hi.data_ptr(),
                       head_capacity=B * kk, unpad_dev=None, verts_dev=None, rot_dev=None, rpy_dev=None, proj_dev=proj.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_lib.vgh_ctx_detect(h, x.data_ptr(), _lib.VGH_IMG_U8_NHWC, B, conf, 0.5, C.byref(o), st) == 0, gpu_lib.vgh_ctx_last_error(h)
    _lib.check(gpu_lib.vgh_ctx_join(h, st))
    torch.cuda.synchronize()
    assert torch.equal(oc, ref.counts) and int(nh) == n_ref
    for b in range(B):
        n = int(oc[b])
        assert torch.equal(ob[b, :n], ref.boxes[b, :n]) and torch.equal(of[b, :n], ref.flame_params[b, :n])
    assert torch.equal(proj[:n_ref], ref.vertices_3d)
    gpu_lib.vgh_destroy(h)
    eng.close()


@pytest.mark.gpu
def test_head_detector_facade_in_the_fp8_and_fp16_modes(gpu_lib, flame_model):
    """The drop-in facade (head_detector/detector.py:97-102 contract) with precision="fp16" / "fp8": same call, same result types; the fp8 detector calibrates its e4m3 links on the
    images it is given (letterboxed like any other input) and warns when it has to fall back to random ones."""
    import warnings

    from head_detector_amd import HeadDetector

    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (300, 420, 3), dtype=np.uint8), rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = HeadDetector("vgg_heads_m", 320, flame_model=flame_model, weights="synthetic", seed=4, precision="fp16x3")
        out = ref(imgs[0], confidence_threshold=0.0)
        sc = sorted((h.score for h in out.heads), reverse=True)
        conf = float(sc[min(5, len(sc) - 1)]) - 1e-3
        base = ref(imgs[0], confidence_threshold=conf)
        dets = {}
        for prec in ("fp16", "fp8", "int8"):
            det = HeadDetector("vgg_heads_m", 320, flame_model=flame_model, weights="synthetic", seed=4, precision=prec, calibration_images=imgs if prec != "fp16" else None)
            if prec != "fp16":
                assert len(det.model.fp8_scales) >= 5 and all(v > 0 for v in det.model.fp8_scales.values())
            dets[prec] = det(imgs[0], confidence_threshold=conf)
    with pytest.warns(UserWarning, match="calibration_images"):
        HeadDetector("vgg_heads_m", 320, flame_model=flame_model, weights="synthetic", seed=4, precision="fp8")
    a = base.heads
    assert len(a) >= 3
    h16 = dets["fp16"].heads
    assert abs(len(h16) - len(a)) <= 1
    for ha, hb in zip(a[:3], h16[:3]):  # the strongest detections agree closely in fp16
        assert abs(ha.bbox.x - hb.bbox.x) <= 2 and abs(ha.bbox.w - hb.bbox.w) <= 3 and abs(ha.score - hb.score) < 1e-3
    for prec in ("fp8", "int8"):
        assert len(dets[prec].heads) >= 1 and all(np.isfinite(np.asarray(h.vertices_3d)).all() for h in dets[prec].heads)


@pytest.mark.gpu
def test_r05_modes_at_1280(gpu_lib):
    """BASELINE configs[4]'s geometry (1280 x 1280: 320- / 160- / 80- / 40-wide maps, 33 600 anchors) in the fp16 and fp8 modes: the e4m3 links and the fp16 ping-pong tiles on
    the 320-wide maps, outputs finite and close to the bf16 mode's on the same image (mean dense IoU; the modes differ by their storage rounding only)."""
    from head_detector_amd.engine import VGHeadsEngine

    x = torch.randint(0, 256, (1, 1280, 1280, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(4)).to(_dev())
    out = {}
    for prec in ("bf16", "fp16", "fp8", "int8"):
        eng = VGHeadsEngine("vgg_heads_l", image_size=1280, max_batch=1, seed=1, precision=prec, calib_images=x if prec in ("fp8", "int8") else None)
        if prec in ("fp8", "int8"):
            assert sum(bf["is_f32"] == (4 if prec == "fp8" else 6) for bf in eng.program.bufs) == 31  # the 40-wide maps of the /32 level qualify at 1280
        eng.model(x)
        torch.cuda.synchronize()
        out[prec] = (eng.boxes_all[:1].cpu().clone(), eng.scores_all[:1].cpu().clone())
        eng.close()

    def iou(a, b):
        lt, rb = torch.maximum(a[..., :2], b[..., :2]), torch.minimum(a[..., 2:], b[..., 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        return inter / ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter)

    assert out["bf16"][0].shape == (1, 33600, 4)
    for prec, lo in (("fp16", 0.95), ("fp8", 0.75), ("int8", 0.75)):
        b, s = out[prec]
        assert bool(torch.isfinite(b).all()) and bool(torch.isfinite(s).all())
        m = float(iou(b, out["bf16"][0]).mean())
        assert m > lo, (prec, m)
        assert float((s - out["bf16"][1]).abs().max()) < (2e-3 if prec == "fp16" else 2e-2), prec
