"""GPU (-m gpu): parity of every HIP kernel, called through the C ABI, against the CPU oracle / golden vectors.
Integer / index results must be bit-exact; floating point within the tolerance written next to each check."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import program_ref as pr
from conftest import golden

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def _sp():
    return torch.cuda.current_stream().cuda_stream


# ======================================================================================================
# FLAME decode (K9)
# ======================================================================================================
@pytest.fixture(scope="module")
def flame_layer(gpu_lib, flame_model):
    from head_detector_amd.flame import FLAMELayer

    return FLAMELayer(model=flame_model, device=_dev(), max_heads=64)


def test_flame_decode_matches_reference_vectors(flame_layer):
    """Golden vectors produced by the reference's FLAMELayer / reproject_spatial_vertices (make_golden.py)."""
    from head_detector_amd.flame import reproject_spatial_vertices

    g = golden("flame_decode.npz")
    params = torch.from_numpy(g["params"]).to(_dev())
    v, R, p = reproject_spatial_vertices(flame_layer, params, to_2d=False)
    # north_star tolerance: FLAME vertices within 1e-4 (metres-scale mesh, |v| ~ 0.2)
    assert np.abs(v.cpu().numpy() - g["vertices"]).max() < 1e-5
    assert np.abs(R.cpu().numpy() - g["R"]).max() < 1e-5
    pg = g["projected"]
    assert np.abs(p.cpu().numpy() - pg).max() < 1e-6 * max(1000.0, np.abs(pg).max())  # pixel space: a few fp32 ulps of the largest coordinate
    _, _, p2 = reproject_spatial_vertices(flame_layer, params, to_2d=True)
    assert p2.shape == (5, 5023, 2) and torch.equal(p2, p[..., :2])
    ev, eR, ep = reproject_spatial_vertices(flame_layer, torch.zeros(0, 413, device=_dev()), to_2d=False)
    assert [list(ev.shape), list(eR.shape), list(ep.shape)] == g["empty_shapes"].tolist()
    with pytest.raises(ValueError):
        reproject_spatial_vertices(flame_layer, torch.zeros(2, 412, device=_dev()))
    sub = [0, 17, 5022]
    _, _, ps = reproject_spatial_vertices(flame_layer, params, to_2d=False, subset_indexes=sub)
    assert torch.equal(ps, p[:, sub])


@pytest.mark.parametrize("n", [1, 2, 3, 7, 24, 25, 64, 65, 150])
def test_flame_decode_vs_oracle_f64(flame_layer, flame_model, n):
    from oracle import flame_oracle as fo

    c64 = fo.FlameConstants(flame_model, torch.float64)
    params = fo.synthetic_params(n, seed=100 + n)
    v64, R64, p64 = fo.reproject(c64, params.double())
    unpad = torch.tensor([[80.0, 0.0, 0.625]]).repeat(n, 1)
    v, R, p = flame_layer.decode(params.to(_dev()), unpad=unpad.to(_dev()), shape_live=300, expr_live=100)
    assert (v.cpu().double() - v64).abs().max() < 2e-6
    assert (R.cpu().double() - R64).abs().max() < 2e-6
    q64 = p64.clone()
    q64[:, :, 0] -= 80.0
    q64 = q64 / 0.625  # detector.py:67-69
    # pixel space (|coords| up to ~1600): a few fp32 ulps of the largest coordinate; metric space is checked at 2e-6 m above
    assert (p.cpu().double() - q64).abs().max() < 1e-6 * max(1000.0, float(q64.abs().max()))
    # structural-zero skipping is bit-exact when the skipped coefficients are exactly zero
    v2, _, p2 = flame_layer.decode(params.to(_dev()), unpad=unpad.to(_dev()), shape_live=128, expr_live=64)
    assert torch.equal(v, v2) and torch.equal(p, p2)


def test_flame_layer_forward_general_pose(flame_layer, flame_model):
    """FLAMELayer.forward with non-zero neck / eyeballs widths (flame.py:141-143 accepts them): general 5-joint LBS."""
    from head_detector_amd.flame import FLAMELayer
    from head_detector_amd.head_info import FlameParams
    from oracle import flame_oracle as fo

    consts = {"shape": 300, "expression": 100, "rotation": 6, "jaw": 3, "eyeballs": 6, "neck": 3, "translation": 3, "scale": 1}
    layer = FLAMELayer(consts=consts, model=flame_model, device=_dev())
    g = torch.Generator().manual_seed(4)
    n = 6
    x = torch.randn(n, sum(consts.values()), generator=g) * 0.3
    fp = FlameParams.from_3dmm(x, consts)
    out = layer.forward(fp, zero_rot=False)
    c64 = fo.FlameConstants(flame_model, torch.float64)
    d = fo.split_3dmm(x.double(), consts)
    ref = fo.flame_forward(c64, d, zero_rot=False)
    assert (out.cpu().double() - ref).abs().max() < 5e-6
    out_zj = layer.forward(fp, zero_rot=True, zero_jaw=True)
    assert (out_zj.cpu().double() - fo.flame_forward(c64, d, zero_rot=True, zero_jaw=True)).abs().max() < 5e-6


@pytest.mark.parametrize("n", [1, 6, 40, 130, 300])
def test_flame_general_lbs_is_bit_identical_across_vertex_kernels(gpu_lib, flame_model, n):
    """vgh_flame_lbs (FLAMELayer.forward's core: betas + a full pose per joint, no 413-vector) through every vertex-kernel family: the component-split tiles read the raw
    betas in place (fused, up to 8 heads) or from the prologue's scratch, with 5-waves-helper, 4-wave and 12 / 15-wave blocks depending on n -- same bits as the VALU kernel."""
    from head_detector_amd.flame import FLAMELayer
    from head_detector_amd.head_info import FlameParams

    consts = {"shape": 300, "expression": 100, "rotation": 6, "jaw": 3, "eyeballs": 6, "neck": 3, "translation": 3, "scale": 1}
    layer = FLAMELayer(consts=consts, model=flame_model, device=_dev(), max_heads=512)
    x = torch.randn(n, sum(consts.values()), generator=torch.Generator().manual_seed(100 + n)) * 0.3
    fp = FlameParams.from_3dmm(x, consts)
    outs = {}
    try:
        for mode in (0, 1, 2, 6, 7):
            assert gpu_lib.vgh_flame_set_matrix_path(mode) == 0
            outs[mode] = layer.forward(fp, zero_rot=False).clone()
    finally:
        gpu_lib.vgh_flame_set_matrix_path(1)
    for mode in (1, 2, 6, 7):
        assert torch.equal(outs[mode], outs[0]), mode


# ======================================================================================================
# top-k (K7), NMS (K8)
# ======================================================================================================
@pytest.mark.parametrize("A,k", [(8400, 1000), (1000, 1000), (33600, 1000), (1500, 7), (64, 64)])
def test_topk_bit_exact(gpu_lib, A, k):
    from head_detector_amd import _lib
    from oracle import postproc_oracle as po

    g = torch.Generator().manual_seed(A + k)
    B = 3
    scores = torch.rand(B, A, generator=g)
    scores[1] = (scores[1] * 50).round() / 50  # massive ties -> index tie-break must match
    scores[2, ::3] = 0.75
    d = scores.to(_dev())
    idx = torch.empty(B, k, dtype=torch.int32, device=_dev())
    out = torch.empty(B, k, device=_dev())
    _lib.check(gpu_lib.vgh_topk(_lib.ptr(d), B, A, k, _lib.ptr(idx), _lib.ptr(out), _sp()))
    for b in range(B):
        ref = po.stable_topk(scores[b], k)
        assert torch.equal(idx[b].cpu().long(), ref), (A, k, b)
        assert torch.equal(out[b].cpu(), scores[b][ref])


@pytest.mark.parametrize("seed,mean_heads,conf", [(1, 3, 0.5), (2, 30, 0.5), (3, 100, 0.02), (4, 3, 0.999), (5, 60, 0.55)])
def test_nms_bit_exact_vs_oracle(gpu_lib, seed, mean_heads, conf):
    from head_detector_amd.utils import nms, nms_batched
    from oracle import postproc_oracle as po

    B = 4
    boxes, scores = po.synthetic_detections(B, seed=seed, mean_heads=mean_heads)
    bb, ss, ff, _ = po.decoding_topk(boxes, scores, torch.randn(B, boxes.shape[1], 413, generator=torch.Generator().manual_seed(seed)), 1000)
    ob, os_, of, counts = nms_batched(bb.to(_dev()), ss.to(_dev()), ff.to(_dev()), confidence_threshold=conf)
    ref = po.postprocess_batched(bb, ss, ff, conf, 0.5)
    for b in range(B):
        n = int(counts[b])
        assert n == ref[b][0].shape[0], (b, n, ref[b][0].shape)
        assert torch.equal(ob[b, :n].cpu(), ref[b][0]) and torch.equal(os_[b, :n].cpu(), ref[b][1]) and torch.equal(of[b, :n].cpu(), ref[b][2])
        assert float(ob[b, n:].abs().sum()) == 0.0
    # reference quirk: nms() returns image 0 only (utils.py:194)
    b0, s0, f0 = nms(bb.to(_dev()), ss.to(_dev()), ff.to(_dev()), confidence_threshold=conf)
    r0 = po.nms_reference(bb, ss, ff, confidence_threshold=conf)
    assert torch.equal(b0.cpu(), r0[0]) and torch.equal(s0.cpu(), r0[1]) and torch.equal(f0.cpu(), r0[2])


def test_nms_reference_golden(gpu_lib):
    from head_detector_amd.utils import nms

    g = golden("nms_glue.npz")
    boxes, scores = torch.from_numpy(g["boxes"]).to(_dev()), torch.from_numpy(g["scores"]).to(_dev())
    flame = torch.randn(2, 1000, 413, generator=torch.Generator().manual_seed(int(g["flame_seed"]))).to(_dev())
    for tag, conf in (("c50", 0.5), ("c02", 0.02), ("c999", 0.999)):
        ob, os_, of = nms(boxes, scores, flame, confidence_threshold=conf)
        assert np.array_equal(ob.cpu().numpy(), g[f"{tag}_boxes"]) and np.array_equal(os_.cpu().numpy(), g[f"{tag}_scores"])
        np.testing.assert_allclose(of.sum(1).cpu().numpy(), g[f"{tag}_flame_rowsum"], rtol=1e-5)


def test_nms_exact_threshold_and_degenerate(gpu_lib):
    from head_detector_amd.utils import nms_batched

    b = torch.tensor([[[0, 0, 2, 2], [0, 0, 2, 1], [0, 0, 0, 0], [0, 0, 0, 0], [5, 5, 6, 6]]], dtype=torch.float32)
    s = torch.tensor([[[0.9], [0.8], [0.7], [0.6], [0.55]]])
    f = torch.zeros(1, 5, 413)
    _, os_, _, counts = nms_batched(b.to(_dev()), s.to(_dev()), f.to(_dev()), 0.5, 0.5)
    assert int(counts[0]) == 5  # IoU exactly 0.5 is NOT suppressed (strict >); 0/0 = nan -> kept
    _, _, _, counts = nms_batched(b.to(_dev()), s.to(_dev()), f.to(_dev()), 0.5, 0.49)
    assert int(counts[0]) == 4


# ======================================================================================================
# head decode (K6) + candidate gather (K6b)
# ======================================================================================================
@pytest.mark.parametrize("S_c,E_c", [(128, 64), (64, 32)])
def test_head_decode_and_gather_vs_oracle(gpu_lib, S_c, E_c):
    from head_detector_amd import _lib
    from oracle import postproc_oracle as po

    torch.manual_seed(S_c)
    B, sizes, strides = 2, [(20, 20), (10, 10), (5, 5)], (8, 16, 32)
    pitch = 72 + S_c + E_c + 13 + 3
    preds, olevels = [], []
    for h, w in sizes:
        t = torch.randn(B, h, w, pitch) * 1.5
        preds.append(t)
        nchw = t.permute(0, 3, 1, 2)
        o = 72  # VGH_PRED_FLAME_OFF
        fl = po.assemble_flame_channels(nchw[:, o : o + S_c], nchw[:, o + S_c : o + S_c + E_c], nchw[:, o + S_c + E_c : o + S_c + E_c + 6],
                                        nchw[:, o + S_c + E_c + 6 : o + S_c + E_c + 9], nchw[:, o + S_c + E_c + 9 : o + S_c + E_c + 12],
                                        nchw[:, o + S_c + E_c + 12 : o + S_c + E_c + 13])
        olevels.append((nchw[:, :68], nchw[:, 68:69], fl))
    rb, rs, rf = po.ndfl_decode(olevels, strides)
    A = rb.shape[1]
    dp = [t.to(_dev()).contiguous() for t in preds]
    lv = (_lib.HeadLevel * 3)(*[_lib.HeadLevel(dp[i].data_ptr(), sizes[i][0], sizes[i][1], pitch, strides[i]) for i in range(3)])
    boxes = torch.empty(B, A, 4, device=_dev())
    scores = torch.empty(B, A, device=_dev())
    _lib.check(gpu_lib.vgh_head_decode(lv, 3, B, _lib.ptr(boxes), _lib.ptr(scores), _sp()))
    assert (boxes.cpu() - rb).abs().max() < 2e-4  # px; IoU >= 0.999 needs ~1e-2 px on a 20 px box
    assert (scores.cpu() - rs[..., 0]).abs().max() < 2e-6
    k = 100
    bb, ss, ff, idx = po.decoding_topk(rb, rs, rf, k)
    didx = idx.to(torch.int32).to(_dev())
    ob = torch.empty(B, k, 4, device=_dev())
    of = torch.empty(B, k, 413, device=_dev())
    _lib.check(gpu_lib.vgh_gather_candidates(lv, 3, B, A, S_c, E_c, _lib.ptr(boxes), _lib.ptr(didx), k, _lib.ptr(ob), _lib.ptr(of), _sp()))
    assert (ob.cpu() - bb).abs().max() < 2e-4
    d = (of.cpu() - ff).abs() / (ff.abs() + 1.0)
    assert d.max() < 1e-5, d.max()  # north_star: FLAME params within 1e-4
    assert float(of[..., S_c:300].abs().max()) == 0.0 and float(of[..., 300 + E_c : 400].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["l", "m"])
def test_post_network_kernels_vs_reference_run_fixture(gpu_lib, flame_model, tag):
    """The HIP post-network kernels against tests/golden/head_decode.npz = outputs of the reference's OWN YoloHeadsDFLHead /
    YoloHeadsNDFLHeads / VGGHeadDecodingModule / YoloHeadsPostPredictionCallback (make_golden_heads.py), fed with the reference's
    raw per-level head outputs: vgh_head_decode, vgh_topk, vgh_gather_candidates (activations, padding, fix-up, permutation),
    the facade's nms_batched (vgh_topk_nms) and the FLAME decode of the survivors."""
    from conftest import golden
    from head_detector_amd import _lib
    from head_detector_amd import utils as hu
    from head_detector_amd.flame import FLAMELayer
    from oracle import flame_oracle as fo

    g = golden("head_decode.npz")
    t = lambda k: torch.from_numpy(g[f"{tag}_{k}"])  # noqa: E731
    B, strides = 2, (8, 16, 32)
    S_c, E_c = g[f"{tag}_raw0_shape"].shape[1], g[f"{tag}_raw0_expression"].shape[1]
    pitch = (72 + S_c + E_c + 13 + 3) // 4 * 4
    preds, sizes = [], []
    for lv in range(3):
        reg = t(f"reg{lv}")
        h, w = reg.shape[2:]
        sizes.append((h, w))
        buf = torch.zeros(B, h, w, pitch)
        parts = [reg, t(f"cls{lv}"), torch.zeros(B, 3, h, w)] + [t(f"raw{lv}_{n}") for n in ("shape", "expression", "rotation", "jaw", "translation", "scale")]
        buf[..., : 72 + S_c + E_c + 13] = torch.cat(parts, 1).permute(0, 2, 3, 1)
        preds.append(buf.to(_dev()).contiguous())
    lv_arr = (_lib.HeadLevel * 3)(*[_lib.HeadLevel(preds[i].data_ptr(), sizes[i][0], sizes[i][1], pitch, strides[i]) for i in range(3)])
    A = sum(h * w for h, w in sizes)
    boxes = torch.empty(B, A, 4, device=_dev())
    scores = torch.empty(B, A, device=_dev())
    _lib.check(gpu_lib.vgh_head_decode(lv_arr, 3, B, _lib.ptr(boxes), _lib.ptr(scores), _sp()))
    assert (boxes.cpu() - t("boxes")).abs().max() < 2e-4  # px
    assert (scores.cpu() - t("scores")[..., 0]).abs().max() < 2e-6
    # row a7: top-k of the reference's scores -> the same candidates in the same order
    k = g[f"{tag}_cand_scores"].shape[1]
    ref_scores = t("scores")[..., 0].contiguous().to(_dev())
    idx = torch.empty(B, k, dtype=torch.int32, device=_dev())
    top = torch.empty(B, k, device=_dev())
    _lib.check(gpu_lib.vgh_topk(_lib.ptr(ref_scores), B, A, k, _lib.ptr(idx), _lib.ptr(top), _sp()))
    assert torch.equal(top.cpu(), t("cand_scores")[..., 0])
    ob = torch.empty(B, k, 4, device=_dev())
    of = torch.empty(B, k, 413, device=_dev())
    _lib.check(gpu_lib.vgh_gather_candidates(lv_arr, 3, B, A, S_c, E_c, _lib.ptr(boxes), _lib.ptr(idx), k, _lib.ptr(ob), _lib.ptr(of), _sp()))
    assert (ob.cpu() - t("cand_boxes")).abs().max() < 2e-4
    d = (of.cpu() - t("cand_flame")).abs() / (t("cand_flame").abs() + 1.0)
    assert d.max() < 1e-5, d.max()
    assert float(of[..., S_c:300].abs().max()) == 0.0 and float(of[..., 300 + E_c : 400].abs().max()) == 0.0
    # batched twin of nms(): the reference's decoded tensors in, its survivors out -- bit-exact decisions, rows copied verbatim
    rb, rs, rf, cnt = hu.nms_batched(t("boxes").to(_dev()), t("scores").to(_dev()), t("flame").to(_dev()), float(g[f"{tag}_post_conf"]), 0.5, top_k=30, keep_top_k=10)
    fl = FLAMELayer(model=fo.synthetic_flame_model(seed=3), device=_dev(), max_heads=64)
    for i in range(B):
        n = int(cnt[i])
        assert n == int(g[f"{tag}_post_counts"][i])
        assert torch.equal(rb[i, :n].cpu(), t(f"post{i}_boxes")) and torch.equal(rs[i, :n].cpu(), t(f"post{i}_scores")) and torch.equal(rf[i, :n].cpu(), t(f"post{i}_params"))
        _, _, proj = fl.decode(rf[i, :n])
        ref = t(f"post{i}_v3d")
        assert (proj[:, ::97].cpu() - ref).abs().max() <= 2e-4 * ref.abs().max()


# ======================================================================================================
# implicit-GEMM conv (K2/K3/K4), per configuration
# ======================================================================================================
def _run_conv(lib, x, W, b, k, stride, act=1, res=None, alpha=0.0, split=None, out_f32=False, shuffle=False, cfg=-1, cout_store=None, in_coff=0, in_pitch=None, out_coff=8):
    """x [B,H,W,Cin] float (bf16-representable); W [rows,k,k,Cin] float; returns engine output + torch reference."""
    from head_detector_amd import _lib

    B, H, Wd, Cin = x.shape
    rows = W.shape[0]
    rp = (rows + 31) // 32 * 32
    Wp = torch.zeros(rp, k, k, Cin)
    Wp[:rows] = W
    bp = torch.zeros(rp)
    bp[:rows] = b
    pack = np.zeros(Wp.numel(), dtype=np.uint16)
    w_np = np.ascontiguousarray(Wp.numpy())
    _lib.check(lib.vgh_pack_conv_weights(_lib.ptr(w_np), rp, k, Cin, _lib.ptr(pack)))
    d_pack = torch.from_numpy(pack.view(np.int16)).to(_dev())
    d_bias = bp.to(_dev())
    in_pitch = in_pitch or (Cin + in_coff)
    xin = torch.zeros(B, H, Wd, in_pitch)
    xin[..., in_coff : in_coff + Cin] = x
    d_x = xin.to(torch.bfloat16).to(_dev()).contiguous()
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (Wd + 2 * (k // 2) - k) // stride + 1
    store = cout_store if cout_store is not None else rows
    oc = rp // 4 if shuffle else rp
    oh, ow = (2 * Ho, 2 * Wo) if shuffle else (Ho, Wo)
    out_pitch = oc + 16
    d_out = torch.full((B, oh, ow, out_pitch), -768.0, dtype=torch.float32 if out_f32 else torch.bfloat16, device=_dev())
    d_res = res.to(torch.bfloat16).to(_dev()).contiguous() if res is not None else None
    call = _lib.ConvCall(
        in_dev=d_x.data_ptr(), in_pitch=in_pitch, in_coff=in_coff, cin=Cin, B=B, H=H, W=Wd, wpack_dev=d_pack.data_ptr(), bias_dev=d_bias.data_ptr(),
        out_dev=d_out.data_ptr(), out_pitch=out_pitch, out_coff=out_coff if not split else split[1], cout_pad=rp, cout_store=store,
        out_split=rp if not split else split[0], out_coff2=0 if not split else split[2], out_f32=int(out_f32),
        res_dev=d_res.data_ptr() if d_res is not None else None, res_pitch=d_res.shape[-1] if d_res is not None else 0, res_coff=0, alpha=alpha,
        ksize=k, stride=stride, act=act, shuffle=int(shuffle), force_cfg=cfg,
    )
    _lib.check(lib.vgh_conv2d(C.byref(call), _sp()))
    torch.cuda.synchronize()
    # reference: fp32 conv on the bf16-rounded operands
    xr = x.to(torch.bfloat16).float().permute(0, 3, 1, 2)
    wr = Wp.to(torch.bfloat16).float().permute(0, 3, 1, 2)
    y = F.conv2d(xr, wr, None, stride=stride, padding=k // 2) + bp[None, :, None, None]
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = F.silu(y)
    y = y.permute(0, 2, 3, 1)
    if shuffle:
        Cc = rp // 4
        z = torch.zeros(B, oh, ow, Cc)
        for d in range(4):
            z[:, d // 2 :: 2, d % 2 :: 2] = y[..., d * Cc : (d + 1) * Cc]
        y = z
    if res is not None:
        y = y + alpha * res.to(torch.bfloat16).float()[..., : y.shape[-1]]
    return d_out.float().cpu(), y, store, out_coff


def _assert_close(out, ref, out_f32, where):
    if out_f32:
        tol = 2e-3 + 1e-4 * ref.abs()
    else:
        tol = 1e-2 + 1.0 / 128 * ref.abs()  # one bf16 ulp (2^-8 relative) + accumulation-order slack
    bad = (out - ref).abs() > tol
    if bad.any():
        idx = bad.nonzero()
        pix = (idx[:, 1] * ref.shape[2] + idx[:, 2]) % 32
        raise AssertionError(f"{where}: {int(bad.sum())}/{bad.numel()} mismatches; max err {float((out - ref).abs().max()):.4f}; "
                             f"first {idx[:5].tolist()}; pixel%32 hist {torch.bincount(pix, minlength=32).tolist()}; "
                             f"chan%32 hist {torch.bincount(idx[:, 3] % 32, minlength=32).tolist()}")


CONV_CASES = [
    # (B, H, W, Cin, Cout, k, stride)
    (1, 16, 16, 32, 32, 1, 1),
    (2, 20, 20, 64, 64, 3, 1),
    (1, 24, 24, 96, 96, 3, 1),
    (2, 40, 40, 64, 128, 3, 2),
    (1, 32, 32, 128, 256, 3, 1),
    (3, 20, 20, 192, 192, 1, 1),
    (1, 33, 17, 64, 96, 3, 2),  # odd sizes, ragged tiles
    (1, 8, 8, 768, 384, 1, 1),
    (1, 5, 5, 32, 13, 1, 1),  # tiny cout (prediction conv)
    (2, 40, 40, 64, 128, 3, 1),  # 40-wide map (row-strip patch tiles)
    (1, 21, 37, 32, 64, 3, 1),  # ragged in both directions for every patch tile shape
    (2, 16, 48, 96, 96, 3, 1),
    (2, 32, 48, 96, 192, 3, 2),  # the r tile's shapes (csrc/ds_b2b.hip: 96 input channels, stride 2, whole 8 x 8 output tiles): two cout halves
    (3, 16, 32, 96, 96, 3, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_all_configs_vs_torch(gpu_lib, case):
    B, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).float()
    Wt = torch.randn(Cout, k, k, Cin, generator=g) * (1.5 / np.sqrt(k * k * Cin))
    # asymmetric structure so that a transposed / permuted fragment layout cannot pass
    Wt = Wt * (1.0 + 0.5 * torch.arange(Cout).float()[:, None, None, None] / Cout)
    b = torch.randn(Cout, generator=g)
    rp = (Cout + 31) // 32 * 32
    ncfg = gpu_lib.vgh_conv_num_cfgs()
    tested = 0
    for cfg in [-1] + list(range(ncfg)):
        out_f32 = Cout % 4 != 0
        for oc0 in (8, 4):  # 8: LDS-transposed 16-byte epilogue (when Cout % 8 == 0); 4: general epilogue
            fast = int(oc0 == 8 and Cout % 8 == 0 and not out_f32)
            if cfg >= 0 and not gpu_lib.vgh_conv_cfg_ok(cfg, k, stride, rp, fast, 0):
                continue
            if cfg >= 0 and gpu_lib.vgh_conv_cfg_name(cfg).decode()[0] == "r" and not (Cin == 96 and H % 16 == 0 and W % 16 == 0):
                continue  # (the r tile's conditions on the input: vgh_conv_cfg_ok sees the output side only)
            if cfg >= 0 and gpu_lib.vgh_conv_cfg_name(cfg).decode()[0] == "w" and not (Cin == gpu_lib.vgh_conv_cfg_cout_tile(cfg) and H % 8 == 0 and W % 8 == 0):
                continue  # (the w tiles: cin = the tile's 96 / 128, whole 8 x 8 tiles)
            out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, k, stride, cfg=cfg, out_f32=out_f32, out_coff=oc0)
            _assert_close(out[..., o0 : o0 + st], ref[..., :st], out_f32, f"{case} cfg={cfg} out_coff={oc0}")
            assert float((out[..., :o0] + 768.0).abs().max()) == 0.0, "wrote outside its channel range"
            assert float((out[..., o0 + st :] + 768.0).abs().max()) < 1.0, "wrote past cout_store"
            tested += 1
    assert tested >= 2


@pytest.mark.parametrize("B,H,W,Cout,split", [(5, 48, 64, 96, None), (3, 32, 32, 192, None), (2, 32, 48, 192, (96, 104, 0)), (40, 16, 16, 96, None)])
def test_conv_r_tile_equals_the_implicit_gemm_tiles(gpu_lib, B, H, W, Cout, split):
    """The r tile (r06, csrc/ds_b2b.hip: a 3x3 / stride-2 conv with 96 input channels on the persistent 4-wave structure -- patch fetched once into parity planes,
    the wave's weights resident in registers, one loader wave): against the torch reference AND bit for bit against an implicit-GEMM tile (same MFMA instruction,
    operand slots, k order, roundings); many tiles per workgroup, image edges on all four sides, both cout halves, an output in two channel segments, an input
    view inside a wider pixel (the neck's concat buffer)."""
    g = torch.Generator().manual_seed(B * 1000 + Cout)
    x = torch.randn(B, H, W, 96, generator=g).to(torch.bfloat16).float()
    Wt = torch.randn(Cout, 3, 3, 96, generator=g) * (1.5 / np.sqrt(9 * 96)) * (1.0 + 0.5 * torch.arange(Cout).float()[:, None, None, None] / Cout)
    b = torch.randn(Cout, generator=g)
    names = [gpu_lib.vgh_conv_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_num_cfgs())]
    r, ig = names.index("r8x8x96_n4"), names.index("128x96_w32x96_k1")
    for in_coff, in_pitch in ((0, None), (32, 160)):
        out_r, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 2, cfg=r, split=split, in_coff=in_coff, in_pitch=in_pitch)
        out_i, _, _, _ = _run_conv(gpu_lib, x, Wt, b, 3, 2, cfg=ig, split=split, in_coff=in_coff, in_pitch=in_pitch)
        assert torch.equal(out_r, out_i), (in_coff, float((out_r.float() - out_i.float()).abs().max()))
        if split is None:
            _assert_close(out_r[..., o0 : o0 + st], ref[..., :st], False, f"r tile {B}x{H}x{W} -> {Cout}")


@pytest.mark.parametrize("B,H,W,Cin,Cout,res,split", [(5, 24, 32, 96, 96, False, None), (3, 16, 24, 128, 128, False, None), (2, 24, 16, 128, 256, False, (128, 136, 0)),
                                                       (7, 16, 16, 96, 96, True, None), (3, 24, 24, 128, 128, True, None), (40, 8, 16, 128, 128, True, None), (2, 16, 16, 96, 192, False, None)])
def test_conv_w_tiles_equal_the_implicit_gemm_tiles(gpu_lib, B, H, W, Cin, Cout, res, split):
    """The w tiles (r06, csrc/ds_b2b.hip: 3x3 / stride-1 convs with 96 / 128 input channels, the wave's 32-cout slice of the weights resident in registers, no barrier
    inside a tile): against the torch reference, and bit for bit against an implicit-GEMM tile -- plain, with a residual (+ alpha * residual in fp32 before the one
    rounding), several cout parts, an output in two channel segments, an input view inside a wider pixel; many tiles per workgroup, image edges on all sides."""
    g = torch.Generator().manual_seed(B * 1000 + Cout + Cin)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).float()
    Wt = torch.randn(Cout, 3, 3, Cin, generator=g) * (1.5 / np.sqrt(9 * Cin)) * (1.0 + 0.5 * torch.arange(Cout).float()[:, None, None, None] / Cout)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).float() if res else None
    names = [gpu_lib.vgh_conv_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_num_cfgs())]
    wt, ig = names.index("w8x8x96_n3" if Cin == 96 else "w8x8x128_n4"), names.index("128x32_w32x32_k1")
    for in_coff, in_pitch in ((0, None), (32, Cin + 64)):
        out_w, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, cfg=wt, split=split, in_coff=in_coff, in_pitch=in_pitch, res=r, alpha=0.75)
        out_i, _, _, _ = _run_conv(gpu_lib, x, Wt, b, 3, 1, cfg=ig, split=split, in_coff=in_coff, in_pitch=in_pitch, res=r, alpha=0.75)
        if split is None:
            _assert_close(out_w[..., o0 : o0 + st], ref[..., :st], False, f"w tile {B}x{H}x{W} {Cin} -> {Cout} res={res}")
        assert torch.equal(out_w, out_i), (in_coff, res, float((out_w.float() - out_i.float()).abs().max()), float((out_w != out_i).float().mean()))


@pytest.mark.parametrize("cin,res", [(64, False), (64, True), (32, True), (96, False)])
def test_conv_persistent_multi_tile(gpu_lib, cin, res):
    """Halo-patch kernels with MANY tiles per workgroup (grid capped to 2 workgroups per XCD): the tile loop of the "p" kernels,
    the operand stream of the 8-wave ping-pong "g" kernels running across tiles (conv_pp.hip) and
    the cross-tile pipeline of the "q" kernels (operands of tile t+1 prefetched under tile t, region parity running across tiles,
    counted store waits, cout-tile changes between consecutive tiles of a workgroup, ragged last tiles)."""
    g = torch.Generator().manual_seed(11 + cin)
    B, H, W, Cout = 3, 40, 56, 128
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).float()
    Wt = torch.randn(Cout, 3, 3, cin, generator=g) * (1.5 / np.sqrt(9 * cin)) * (1.0 + 0.5 * torch.arange(Cout).float()[:, None, None, None] / Cout)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).float() if res else None
    names = [gpu_lib.vgh_conv_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_num_cfgs())]
    tested = 0
    try:
        for cap in (2, 5):
            assert gpu_lib.vgh_conv_set_max_blocks_per_xcd(cap) == 0
            for cfg, name in enumerate(names):
                if name[0] not in "pqghs" or not gpu_lib.vgh_conv_cfg_ok(cfg, 3, 1, Cout, 1, 0):
                    continue
                out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, cfg=cfg, res=r, alpha=0.37 if res else 0.0)
                _assert_close(out[..., o0 : o0 + st], ref[..., :st], False, f"multi-tile cfg={name} cap={cap} cin={cin} res={res}")
                assert float((out[..., :o0] + 768.0).abs().max()) == 0.0 and float((out[..., o0 + st :] + 768.0).abs().max()) == 0.0, f"{name}: wrote outside its channels"
                tested += 1
    finally:
        gpu_lib.vgh_conv_set_max_blocks_per_xcd(0)
    assert tested >= 20


@pytest.mark.parametrize("cin,cout,cap", [(32, 64, 2), (64, 192, 2), (96, 96, 3), (96, 192, 64), (384, 96, 2), (640, 192, 5), (160, 128, 1)])
def test_conv_stream_tiles_multi_tile(gpu_lib, cin, cout, cap):
    """Streaming 1x1 tiles ("t": conv1x1_stream_kernel) with MANY tiles per persistent workgroup: the ring of stages running across tiles (operands of
    tile t+1 in flight under the MFMAs and stores of tile t), the counted waits over loads AND stores (K loops of 1, 2, 3, 5, 12 and 20 steps against
    3- and 4-stage rings), cout-tile changes between consecutive tiles, a ragged last pixel tile, two output segments and a masked channel tail."""
    g = torch.Generator().manual_seed(3 * cin + cout)
    B, H, W = 3, 41, 55  # 6765 pixels: not a multiple of any tile
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).float()
    Wt = torch.randn(cout, 1, 1, cin, generator=g) * (1.5 / np.sqrt(cin)) * (1.0 + 0.5 * torch.arange(cout).float()[:, None, None, None] / cout)
    b = torch.randn(cout, generator=g)
    names = [gpu_lib.vgh_conv_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_num_cfgs())]
    tested = 0
    try:
        assert gpu_lib.vgh_conv_set_max_blocks_per_xcd(cap) == 0
        for cfg, name in enumerate(names):
            if name[0] != "t" or not gpu_lib.vgh_conv_cfg_ok(cfg, 1, 1, cout, 1, 0):
                continue
            for act in (1, 0):
                out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 1, 1, cfg=cfg, act=act)
                _assert_close(out[..., o0 : o0 + st], ref[..., :st], False, f"stream cfg={name} cap={cap} cin={cin} act={act}")
                assert float((out[..., :o0] + 768.0).abs().max()) == 0.0 and float((out[..., o0 + st :] + 768.0).abs().max()) == 0.0, f"{name}: wrote outside its channels"
            # two output segments (CSP conv1|conv2) and a masked tail (cout_store < cout_pad)
            h = cout // 2 // 8 * 8
            out, ref, st, _ = _run_conv(gpu_lib, x, Wt, b, 1, 1, split=(h, cout - h + 8, 0), cfg=cfg)
            _assert_close(out[..., cout - h + 8 : cout + 8], ref[..., :h], False, f"stream split seg0 cfg={name}")
            _assert_close(out[..., 0 : cout - h], ref[..., h:cout], False, f"stream split seg1 cfg={name}")
            out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 1, 1, cfg=cfg, cout_store=cout - 8)
            _assert_close(out[..., o0 : o0 + cout - 8], ref[..., : cout - 8], False, f"stream masked tail cfg={name}")
            assert float((out[..., o0 + cout - 8 :] + 768.0).abs().max()) == 0.0, f"{name}: wrote past cout_store"
            tested += 1
    finally:
        gpu_lib.vgh_conv_set_max_blocks_per_xcd(0)
    assert tested >= 2


def test_conv_silu_epilogue(gpu_lib):
    """VGH_ACT_SILU (north_star's "BN/SiLU fusion"; the VGGHeads graphs themselves are all-ReLU): x * sigmoid(x) with the hardware
    exp (__expf, ~2 ulp fp32 -- far below the bf16 output rounding, and within 1e-6 relative on the fp32 store path) in every
    epilogue family: LDS-transposed and general implicit-GEMM epilogues, fp32 store, residual after the activation, the halo-patch
    kernels ("p") and the pipelined ones ("q")."""
    g = torch.Generator().manual_seed(9)
    B, H, W, Cin, Cout = 2, 24, 24, 64, 128
    x = torch.randn(B, H, W, Cin, generator=g)
    Wt = torch.randn(Cout, 3, 3, Cin, generator=g) * 0.08
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, H, W, Cout, generator=g)
    names = [gpu_lib.vgh_conv_cfg_name(i).decode() for i in range(gpu_lib.vgh_conv_num_cfgs())]
    picks = [-1] + [names.index(n) for n in ("128x128_w64x64_k1", "256x128_w64x64_k1_r3", "p16x16x64_n4x1", "q16x16x64_n4x1", "p8x32x64_n4x2", "q16x16x128_n4x2")]
    for cfg in picks:
        for r, alpha in ((None, 0.0), (res, 0.3)):
            out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, act=2, cfg=cfg, res=r, alpha=alpha)
            _assert_close(out[..., o0 : o0 + st], ref[..., :st], False, f"silu cfg={cfg} res={r is not None}")
    for cfg in picks[:3]:  # general epilogue (8-byte aligned offset) and the fp32 store path
        out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, act=2, cfg=cfg, out_coff=4)
        _assert_close(out[..., o0 : o0 + st], ref[..., :st], False, f"silu general epilogue cfg={cfg}")
    W1 = torch.randn(69, 1, 1, Cin, generator=g) * 0.1
    out, ref, st, o0 = _run_conv(gpu_lib, x, W1, torch.randn(69, generator=g), 1, 1, act=2, out_f32=True, out_coff=8)
    _assert_close(out[..., o0 : o0 + st], ref[..., :st], True, "silu fp32 store")
    assert ((out[..., o0 : o0 + st] - ref[..., :st]).abs() / (ref[..., :st].abs() + 1e-3)).max() < 2e-2  # bf16 inputs dominate; __expf is invisible


def test_conv_epilogues(gpu_lib):
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 12, 12, 64, 128
    x = torch.randn(B, H, W, Cin, generator=g)
    Wt = torch.randn(Cout, 3, 3, Cin, generator=g) * 0.06
    b = torch.randn(Cout, generator=g)
    # residual added AFTER the activation (YoloNASBottleneck)
    res = torch.randn(B, H, W, Cout, generator=g)
    for cfg in range(-1, gpu_lib.vgh_conv_num_cfgs()):
        if cfg >= 0 and (not gpu_lib.vgh_conv_cfg_ok(cfg, 3, 1, 128, 1, 0) or gpu_lib.vgh_conv_cfg_name(cfg).decode()[0] in "rw"):
            continue  # (r / w tiles: conditions on the input side too -- their own tests, test_conv_*_tile*_equal_the_implicit_gemm_tiles, cover residual and split stores)
        out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, res=res, alpha=0.7, cfg=cfg)
        _assert_close(out[..., o0 : o0 + st], ref, False, f"residual cfg={cfg}")
    out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, res=res, alpha=0.7, out_coff=4)
    _assert_close(out[..., o0 : o0 + st], ref, False, "residual general epilogue")
    # no activation
    out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, act=0)
    _assert_close(out[..., o0 : o0 + st], ref, False, "act none")
    assert float(ref.min()) < -0.5 and float(out[..., o0 : o0 + st].min()) < -0.5
    # two-segment output (CSP conv1|conv2): first 64 rows at offset 72, the rest at offset 0
    for cfg in range(-1, gpu_lib.vgh_conv_num_cfgs()):
        if cfg >= 0 and (not gpu_lib.vgh_conv_cfg_ok(cfg, 3, 1, 128, 1, 0) or gpu_lib.vgh_conv_cfg_name(cfg).decode()[0] in "rw"):
            continue  # (r / w tiles: conditions on the input side too -- their own tests, test_conv_*_tile*_equal_the_implicit_gemm_tiles, cover residual and split stores)
        out, ref, st, _ = _run_conv(gpu_lib, x, Wt, b, 3, 1, split=(64, 72, 0), cfg=cfg)
        _assert_close(out[..., 72:136], ref[..., :64], False, f"split seg0 cfg={cfg}")
        _assert_close(out[..., 0:64], ref[..., 64:128], False, f"split seg1 cfg={cfg}")
    # fp32 output with a ragged channel count (prediction convs): 69 channels
    W69 = torch.randn(69, 1, 1, Cin, generator=g) * 0.1
    out, ref, st, o0 = _run_conv(gpu_lib, x, W69, torch.randn(69, generator=g), 1, 1, act=0, out_f32=True, out_coff=5)
    _assert_close(out[..., 5 : 5 + 69], ref[..., :69], True, "f32 out")
    assert float((out[..., 5 + 69 :] + 768.0).abs().max()) == 0.0
    # the same through the transposed float4 epilogue (16-byte aligned channel offset, as the network's prediction buffers):
    # ragged 69 / 13 live channels, every tile config that can run a 1x1 conv; untouched floats keep the sentinel
    for rows_ in (69, 13):
        Wr, br = torch.randn(rows_, 1, 1, Cin, generator=g) * 0.1, torch.randn(rows_, generator=g)
        for cfg in range(-1, gpu_lib.vgh_conv_num_cfgs()):
            if cfg >= 0 and not gpu_lib.vgh_conv_cfg_ok(cfg, 1, 1, 96 if rows_ == 69 else 32, 0, 0):
                continue
            out, ref, st, o0 = _run_conv(gpu_lib, x, Wr, br, 1, 1, act=0, out_f32=True, out_coff=8, cfg=cfg)
            _assert_close(out[..., 8 : 8 + rows_], ref[..., :rows_], True, f"f32 float4 epilogue rows={rows_} cfg={cfg}")
            assert float((out[..., 8 + rows_ :] + 768.0).abs().max()) == 0.0 and float((out[..., :8] + 768.0).abs().max()) == 0.0
    # input view at a channel offset inside a wider (concat) buffer
    for cfg in (-1, 15, 16):
        out, ref, st, o0 = _run_conv(gpu_lib, x, Wt, b, 3, 1, in_coff=32, in_pitch=160, cfg=cfg)
        _assert_close(out[..., o0 : o0 + st], ref, False, f"in_coff cfg={cfg}")
    # ConvTranspose2d(k=2, s=2) as 4 pointwise GEMMs + pixel-shuffle store, checked against torch's own op
    Ct = 96
    xt = torch.randn(B, H, W, Ct, generator=g).to(torch.bfloat16).float()
    Wct = (torch.randn(Ct, Ct, 2, 2, generator=g) * 0.1).to(torch.bfloat16).float()
    bt = torch.randn(Ct, generator=g)
    Wg = torch.zeros(4 * Ct, 1, 1, Ct)
    for dy in range(2):
        for dx in range(2):
            Wg[(dy * 2 + dx) * Ct : (dy * 2 + dx + 1) * Ct, 0, 0] = Wct[:, :, dy, dx].T
    for oc0 in (8, 4):
        out, ref, st, o0 = _run_conv(gpu_lib, xt, Wg, bt.repeat(4), 1, 1, act=0, shuffle=True, out_coff=oc0)
        tref = F.conv_transpose2d(xt.permute(0, 3, 1, 2), Wct, bt, stride=2).permute(0, 2, 3, 1)
        assert (ref - tref).abs().max() < 1e-4
        _assert_close(out[..., o0 : o0 + Ct], tref, False, f"convT shuffle out_coff={oc0}")


# ======================================================================================================
# whole network: every op of the engine against the torch executor on the engine's own inputs
# ======================================================================================================
@pytest.mark.parametrize("variant,S,B", [("vgg_heads_m", 192, 2), ("vgg_heads_l", 128, 1)])
def test_network_every_op(gpu_lib, variant, S, B):
    from head_detector_amd.engine import VGHeadsEngine

    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=7, use_tuning=False)
    eng.set_b2b(False)  # every op as its own launch: every intermediate tensor exists (the fused back-to-back pairs: test_b2b_pairs_equal_their_two_launches)
    P = eng.program
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(2))
    eng.forward_net(x.to(_dev()))
    got = [eng.buffer(i, B).float().cpu() for i in range(len(P.bufs))]
    w_all, b_all = P.arrays()
    worst = []
    for op in P.ops:
        if op["kind"] == 3:
            continue
        ob = op["out_buf"] if op["kind"] != 2 else op["in_buf"]
        exp = list(got)  # engine's own inputs -> no error accumulation across layers
        exp[ob] = got[ob].clone()
        pr.run_op(P, op, exp, x, True, w_all, b_all)
        is_f32 = bool(P.bufs[ob]["is_f32"])
        a, e = got[ob], exp[ob]
        tol = (2e-3 + 1e-4 * e.abs()) if is_f32 else (2e-2 + 1.0 / 64 * e.abs())  # bf16: 2 ulps (accumulation order can cross a rounding boundary)
        bad = (a - e).abs() > tol
        worst.append((float(((a - e).abs() / (e.abs() + 1.0)).max()), op["name"]))
        assert not bad.any(), f"{variant} S={S} op {op['name']}: {int(bad.sum())} mismatches, max abs err {float((a - e).abs().max())}"
    eng.close()


@pytest.mark.parametrize("variant,S,B,tuned", [("vgg_heads_l", 160, 3, False), ("vgg_heads_m", 256, 2, False), ("vgg_heads_l", 640, 2, False), ("vgg_heads_l", 640, 5, True), ("vgg_heads_m", 224, 7, True)])
def test_b2b_pairs_equal_their_two_launches(gpu_lib, variant, S, B, tuned):
    """Back-to-back GEMM (r06, csrc/conv_kernels.inc T2 > 0): a stage's downsample and the CSP layer's conv1|conv2 behind it as ONE launch -- the
    first conv's accumulators become the second GEMM's B operands in registers, the 96-channel tensor between them is never written.  Against the two launches it
    replaces (implicit-GEMM tiles: the untuned engines): the second conv's output tensor, every later tensor and the network's outputs are THE SAME BITS (same k order, same
    roundings: the fragment a lane builds is byte for byte what it would have read back), the tensor in between stays untouched (zeros), for a pixel count that is not a multiple of the 128-pixel tile,
    with and without batch-split lanes, L (192 fused output channels) and M (128; M also has a 1x1 -> 1x1 pair in its neck)."""
    from head_detector_amd import arch
    from head_detector_amd.engine import VGHeadsEngine

    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=11, use_tuning=tuned)
    P = eng.program
    pairs = arch.b2b_pairs(P)
    assert eng.b2b_pairs == len(pairs) >= 1 and P.ops[pairs[0]]["name"] == "backbone.stage1.downsample"
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(S)).to(_dev())
    outs = {}
    for fused in (False, True):
        eng.set_b2b(3 if fused else 0)  # (3: the pairs fused, the stem as its own launch: every bit comparable -- the default mode 1 is measured against it below)
        for ns in (1, 2):
            eng.set_split(ns)
            res = [t.clone() for t in eng.model(x)]
            bufs = {i: eng.buffer(P.ops[i + 1]["out_buf"], B).clone() for i in pairs}
            outs[(fused, ns)] = (res, bufs)
    mid = eng.buffer(P.ops[pairs[0]]["out_buf"], B)
    assert float(mid.float().abs().max()) > 0  # (the unfused runs above wrote it)
    for ns in (1, 2):  # (a tuned engine may run other tiles -- other summation orders -- with two lanes than with one: fused against unfused at the SAME lane count)
        ref, (res, bufs) = outs[(False, ns)], outs[(True, ns)]
        for i in pairs:
            o2 = P.ops[i + 1]  # the second conv's own channels of its output buffer (a CSP concat buffer: later ops write the rest of it)
            seg = lambda t: torch.cat([t[..., o2["out_coff"]:o2["out_coff"] + o2["out_split"]], t[..., o2["out_coff2"]:o2["out_coff2"] + o2["cout_store"] - o2["out_split"]]], -1)  # noqa: E731
            a, b = seg(bufs[i]).float(), seg(ref[1][i]).float()
            assert float(a.abs().max()) > 0
            if tuned:
                # a TUNED engine may run the unfused second conv on a streaming 1x1 tile, whose accumulators START at the bias (bias + sum instead of sum + bias: another fp32
                # rounding, a flipped bf16 ulp in ~2e-5 of the outputs); the fused launch keeps the implicit-GEMM convention.  Only the FIRST pair sees the same input in
                # both runs (the stem's tensor): those rare ulps of its output then travel through fifty layers of a random-weight network, and the later pairs' INPUTS
                # already differ between the fused and the unfused run (measured: 42 % of the neck pair's outputs, 0.9 % in norm).  Every pair's bit-identity is what the
                # untuned cases above prove; here: the first pair to one bf16 ulp, almost everywhere exactly.
                if i != pairs[0]:
                    continue
                d = (a - b).abs()
                assert not (d > 5e-2 + b.abs() / 32).any(), (ns, P.ops[i]["name"], float(d.max()))
                assert float((a != b).float().mean()) < 1e-3, (ns, P.ops[i]["name"], float((a != b).float().mean()))
            else:
                assert torch.equal(bufs[i], ref[1][i]), (ns, P.ops[i]["name"])
        if not tuned:
            for a, b in zip(res, ref[0]):
                assert torch.equal(a, b), ns
    ref = outs[(True, 1)]
    # the stage-1 pair ran on the persistent "t" tile above (csrc/ds_b2b.hip: patch fetched once into parity planes, both convs' weights resident in registers); the same
    # pair on the implicit-GEMM b2b tile (set_b2b(2)): the same bits again -- two independent implementations of one summation order
    eng.set_b2b(2)
    for ns in (1, 2):
        eng.set_split(ns)
        res = eng.model(x)
        for i in pairs:
            assert torch.equal(eng.buffer(P.ops[i + 1]["out_buf"], B), outs[(True, ns)][1][i]), (ns, P.ops[i]["name"])
        assert all(torch.equal(a, b) for a, b in zip(res, outs[(True, ns)][0])), ns
    # the DEFAULT mode (1) with u8 images: the stem conv runs inside the stage-1 pair's launch as a bf16 x 3 split GEMM ("u" tile: a u8 pixel is an exact bf16, the weights / 255
    # are three bf16 values; exact products, fp32 accumulation in another order than the stem kernel's fmaf chain).  Not the same bits: a flipped bf16 ulp in ~4e-5 of the stem
    # values, which shows in < 2e-3 of the pair's outputs, by one ulp of the value's magnitude (later layers of a random-weight network amplify it: only this pair is compared)
    eng.set_b2b(1)
    o2 = P.ops[pairs[0] + 1]
    seg = lambda t: torch.cat([t[..., o2["out_coff"]:o2["out_coff"] + o2["out_split"]], t[..., o2["out_coff2"]:o2["out_coff2"] + o2["cout_store"] - o2["out_split"]]], -1)  # noqa: E731
    for ns in (1, 2):
        eng.set_split(ns)
        eng.model(x)
        a, b = seg(eng.buffer(o2["out_buf"], B)).float(), seg(outs[(True, ns)][1][pairs[0]]).float()
        d = (a - b).abs()
        assert not (d > 2e-2 + b.abs() / 64).any(), (ns, float(d.max()))
        assert float((a != b).float().mean()) < 2e-3, (ns, float((a != b).float().mean()))
    eng.close()
    # fresh engines that only ever ran fused: mode 3 = the same bits as the fused runs above and the tensor between the two convs is never written; the default mode: neither is
    # the stem's tensor
    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=11, use_tuning=tuned)
    eng.set_b2b(3)
    res = eng.model(x)
    assert all(torch.equal(a, b) for a, b in zip(res, ref[0]))
    assert float(eng.buffer(P.ops[pairs[0]]["out_buf"], B).float().abs().max()) == 0.0
    eng.close()
    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=11, use_tuning=tuned)
    eng.model(x)
    assert float(eng.buffer(P.ops[pairs[0]]["out_buf"], B).float().abs().max()) == 0.0 and float(eng.buffer(P.ops[0]["out_buf"], B).float().abs().max()) == 0.0
    assert float(seg(eng.buffer(o2["out_buf"], B)).float().abs().max()) > 0
    eng.close()


def test_stem_tensor_at_its_48_channel_pitch(gpu_lib, monkeypatch):
    """bf16 mode (r04): the stem tensor is stored as 96-byte pixels and the stage-1 downsample reads 64-channel K windows over it -- the last 16 channels of
    a window are the next pixel's first 16 and meet all-zero weight columns.  (i) the stem buffer holds the 48 channels of the fp32 reference, nothing else;
    (ii) the downsample equals the torch conv on the 48 real channels (also at the last pixel of the last image, whose window runs into the arena's slack)
    for a batch below the arena batch after a larger one has left its data behind; (iii) vgh_net_create refuses a non-zero weight in a padded column."""
    from head_detector_amd import _lib, arch
    from head_detector_amd.engine import VGHeadsEngine

    S, B = 160, 3
    sd = arch.random_state_dict("vgg_heads_l", 5)
    eng = VGHeadsEngine("vgg_heads_l", state_dict=sd, image_size=S, max_batch=B, use_tuning=False)
    eng.set_b2b(False)  # (ii) reads the stage-1 downsample's own output tensor
    P = eng.program
    assert P.bufs[0]["pitch"] == 48 and P.ops[0]["cout_store"] == 48 and P.ops[1]["cin"] == 64
    g = torch.Generator().manual_seed(11)
    big = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g)
    x = torch.randint(0, 256, (2, S, S, 3), dtype=torch.uint8, generator=g)
    eng.forward_net(big.to(_dev()))  # image 2's stem pixels stay in the arena behind the 2-image batch
    eng.forward_net(x.to(_dev()))
    stem = eng.buffer("stem", 2).float().cpu()
    assert stem.shape == (2, S // 2, S // 2, 48)
    bufs = pr.alloc(P, 2)
    pr.run_op(P, P.ops[0], bufs, x, True)
    assert torch.equal(stem, bufs[0])
    pr.run_op(P, P.ops[1], bufs, x, True)
    ds = eng.buffer("backbone.stage1.ds", 2).float().cpu()
    ref = bufs[P.ops[1]["out_buf"]]
    assert float(ref.abs().max()) > 0 and not ((ds - ref).abs() > 2e-2 + 1.0 / 64 * ref.abs()).any(), "stage-1 downsample over the 48-channel pitch"  # bf16: 2 ulps
    eng.close()

    real = arch.build_program

    def poisoned(*a, **k):
        Q = real(*a, **k)
        Q.weights[1].reshape(-1, 64)[5, 50] = 0.25  # stage-1 downsample: a weight on input channel 50, which the 48-channel tensor does not have
        return Q

    monkeypatch.setattr(arch, "build_program", poisoned)
    with pytest.raises(_lib.VghError, match="non-zero weight at input channel 50"):
        VGHeadsEngine("vgg_heads_l", state_dict=sd, image_size=S, max_batch=1, use_tuning=False)


def test_u8_nhwc_input_equals_f32_nchw(gpu_lib):
    """The fused /255 (detector.py:51) must reproduce the float path exactly."""
    from head_detector_amd.engine import VGHeadsEngine

    S = 128
    eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=1, seed=3, use_tuning=False)
    eng.set_b2b(False)
    u8 = torch.randint(0, 256, (1, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    f = (u8.permute(0, 3, 1, 2).float() / 255.0).contiguous()
    for name in ("stem", "backbone.stage1.ds"):  # the stem kernel's own output, and the first conv behind it
        eng.forward_net(u8.to(_dev()))
        a = eng.buffer(name, 1).float().cpu()
        eng.forward_net(f.to(_dev()))
        b = eng.buffer(name, 1).float().cpu()
        assert float(a.abs().max()) > 0 and torch.equal(a, b), name
    eng.close()


def test_end_to_end_vs_fp32_oracle_and_bf16_emulation(gpu_lib, flame_model):
    """engine.model() against (i) the torch executor with bf16 storage emulation: same candidates / tight tolerances;
    (ii) the unfused fp32 oracle network: reported deviation of the bf16 throughput mode (loose bound)."""
    from head_detector_amd import arch
    from head_detector_amd.engine import VGHeadsEngine
    from oracle import net_oracle, postproc_oracle as po

    variant, S, B = "vgg_heads_m", 256, 1
    sd = arch.random_state_dict(variant, 11)
    eng = VGHeadsEngine(variant, state_dict=sd, image_size=S, max_batch=B, use_tuning=False)
    P = eng.program
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(3))
    boxes, scores, flame = eng.model(x.to(_dev()))
    assert boxes.shape == (B, 1000, 4) and scores.shape == (B, 1000, 1) and flame.shape == (B, 1000, 413)
    # (i) bf16-emulating executor -> oracle post-network stages
    bufs = pr.run_program(P, x, bf16=True)
    lv = [(reg, cls, po.assemble_flame_channels(raw["shape"], raw["expr"], raw["rot"], raw["jaw"], raw["trans"], raw["scale"])) for reg, cls, raw in pr.head_outputs(P, bufs)]
    rb, rs, rf = po.ndfl_decode(lv)
    s_gpu = scores[..., 0].cpu()
    assert (s_gpu[:, :-1] >= s_gpu[:, 1:]).all(), "candidates must be sorted by descending score"
    # layer-to-layer bf16 rounding flips make the two nets differ slightly -> compare the dense score field statistically
    dense = eng.scores_all[:B].cpu()
    assert (dense - rs[..., 0]).abs().max() < 3e-2
    # (ii) fp32 unfused oracle
    oracle = net_oracle.YoloHeadsOracle("m")
    oracle.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ob, os_, of = oracle.dense(x)
    rel = float((eng.boxes_all[:B].cpu() - ob).abs().mean() / ob.abs().mean())
    print(f"[bf16 vs fp32 oracle] mean rel box err {rel:.4f}, max score err {float((dense - os_[..., 0]).abs().max()):.4f}")
    assert rel < 5e-2
    eng.close()


def test_detect_pipeline_and_facade(gpu_lib, flame_model):
    """HeadDetector()(image).heads: API surface, types and the post-network stages against the oracle, given the engine's own
    network output (the random-weight network's scores are arbitrary, so the threshold is lowered to get detections)."""
    from head_detector_amd.detector import HeadDetector
    from head_detector_amd.head_info import Bbox, FlameParams, HeadMetadata, RPY
    from oracle import flame_oracle as fo
    from oracle import postproc_oracle as po

    det = HeadDetector("vgg_heads_m", 640, flame_model=flame_model, weights="synthetic", seed=4)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (480, 600, 3), dtype=np.uint8)  # letterboxed: scale 640/600, pad (0, 64)
    image, cache = det._preprocess(img)
    assert image.shape == (1, 640, 640, 3) and cache["padding"] == (0, (640 - int(480 * 640 / 600)) // 2) and abs(cache["scale"] - 640 / 600) < 1e-12
    b, s, f = det._process(image)
    conf = float(s[0, 20, 0])  # keep roughly the 20 best candidates before NMS
    res = det(img, confidence_threshold=conf)
    assert len(res.heads) >= 1 and all(isinstance(h, HeadMetadata) for h in res.heads)
    rb, rs, rf = po.nms_reference(b.cpu(), s.cpu(), f.cpu(), confidence_threshold=conf)
    c32 = fo.FlameConstants(flame_model, torch.float32)
    xywh, verts, pout = fo.parse_predictions(c32, rb.numpy(), rf, cache["padding"], cache["scale"])
    assert len(res.heads) == rb.shape[0]
    for i, h in enumerate(res.heads):
        assert isinstance(h.bbox, Bbox) and isinstance(h.head_pose, RPY) and isinstance(h.flame_params, FlameParams)
        assert (h.bbox.x, h.bbox.y, h.bbox.w, h.bbox.h) == tuple(int(v) for v in xywh[i])
        assert h.vertices_3d.shape == (5023, 3) and h.vertices_3d.dtype == np.float32
        assert np.isfinite(verts[i]).all() and np.isfinite(h.vertices_3d).all()
        assert np.abs(h.vertices_3d - verts[i]).max() < 2e-6 * max(1000.0, float(np.abs(verts[i]).max()))
        assert abs(float(h.score) - float(rs[i])) == 0.0
        np.testing.assert_allclose(h.flame_params.scale.numpy(), pout[i : i + 1, 412:413].numpy(), rtol=1e-6)
        np.testing.assert_allclose(h.flame_params.translation.numpy(), rf[i : i + 1, 409:412].numpy())  # NOT un-padded (detector.py:78-79)
        rpy = fo.calculate_rpy(rf[i, 403:409])
        np.testing.assert_allclose(np.array(h.head_pose), np.array(rpy), atol=1e-3)
    # batched twin on the engine: every image, slabs + counts, vertices for every valid head
    eng = det.model
    d = eng.detect(image, confidence_threshold=conf, flame=det._flame)
    assert int(d.counts[0]) == len(res.heads) and d.vertices_3d.shape == (len(res.heads), 5023, 3)
    assert torch.equal(d.boxes[0, : len(res.heads)].cpu(), rb)
    # empty result path (detector must not fail when nothing passes the threshold; flame.py:186-189)
    empty = det(img, confidence_threshold=1.1)
    assert empty.heads == []


def test_batch_chunking_is_invisible(gpu_lib):
    """Batches whose tensors would cross the 2 GiB / 32-bit-offset limit run through the network in arena-sized chunks;
    forced here with a tiny arena: results must be bit-identical to the unchunked run."""
    from head_detector_amd.engine import VGHeadsEngine

    S, B = 128, 5
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).to(_dev())
    a = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=2, use_tuning=False)
    b = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=2, use_tuning=False, arena_batch=2)
    assert a.arena_batch == 5 and b.arena_batch == 2
    ra = [t.clone() for t in a.model(x)]
    rb = [t.clone() for t in b.model(x)]
    for u, v in zip(ra, rb):
        assert torch.equal(u, v)
    da, db = a.detect(x, confidence_threshold=float(ra[1][:, 5, 0].min())), b.detect(x, confidence_threshold=float(ra[1][:, 5, 0].min()))
    assert torch.equal(da.counts, db.counts) and torch.equal(da.boxes, db.boxes)
    a.close()
    b.close()


def test_1280_crowd_config_shapes(gpu_lib, flame_model):
    """BASELINE config 5 geometry: 1280x1280 -> 33 600 anchors, top-k 1000, crowd NMS keeps up to 100 heads per image."""
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer
    from oracle import postproc_oracle as po

    eng = VGHeadsEngine("vgg_heads_l", image_size=1280, max_batch=2, seed=1)
    assert eng.A == 33600
    x = torch.randint(0, 256, (2, 1280, 1280, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).to(_dev())
    boxes, scores, flame = eng.model(x)
    assert boxes.shape == (2, 1000, 4) and torch.isfinite(flame).all()
    ref_idx = po.stable_topk(eng.scores_all[0].cpu(), 1000)
    assert torch.equal(eng.idx[0].cpu().long(), ref_idx)
    conf = float(scores[:, 400, 0].max())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=256)
    det = eng.detect(x, confidence_threshold=conf, flame=fl)
    ref = po.postprocess_batched(boxes.cpu(), scores.cpu(), flame.cpu(), conf, 0.5)
    for b in range(2):
        n = int(det.counts[b])
        assert n == ref[b][0].shape[0] and torch.equal(det.boxes[b, :n].cpu(), ref[b][0])
    assert det.vertices_3d.shape[0] == int(det.counts.sum())
    eng.close()


def _box_iou(a, b):
    x1, y1 = torch.maximum(a[..., 0], b[..., 0]), torch.maximum(a[..., 1], b[..., 1])
    x2, y2 = torch.minimum(a[..., 2], b[..., 2]), torch.minimum(a[..., 3], b[..., 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    ua = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
    return inter / ua


@pytest.mark.parametrize("variant,okey", [("vgg_heads_m", "m"), ("vgg_heads_l", "l")])
def test_fp32_parity_mode_meets_north_star_tolerances(gpu_lib, variant, okey):
    """precision='fp32' (no bf16 anywhere): every op against the torch executor at fp32 round-off, and the whole network against
    the UNFUSED fp32 oracle at BASELINE.json's bar: bbox IoU >= 0.999, scores / FLAME params within 1e-4."""
    from head_detector_amd import arch
    from head_detector_amd.engine import VGHeadsEngine
    from oracle import net_oracle

    S, B = 160, 2
    sd = arch.random_state_dict(variant, 21)
    eng = VGHeadsEngine(variant, state_dict=sd, image_size=S, max_batch=B, precision="fp32")
    P = eng.program
    assert all(bf["is_f32"] for bf in P.bufs)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(5))
    boxes, scores, flame = eng.model(x.to(_dev()))
    got = [eng.buffer(i, B).cpu() for i in range(len(P.bufs))]
    w_all, b_all = P.arrays()
    for op in P.ops:
        if op["kind"] == 3:
            continue
        ob = op["out_buf"] if op["kind"] != 2 else op["in_buf"]
        exp = list(got)
        exp[ob] = got[ob].clone()
        pr.run_op(P, op, exp, x, False, w_all, b_all)
        err = (got[ob] - exp[ob]).abs() / (exp[ob].abs() + 1.0)
        assert float(err.max()) < 2e-5, (op["name"], float(err.max()))
    oracle = net_oracle.YoloHeadsOracle(okey)
    oracle.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ob_, os_, of_ = oracle.dense(x)
    A = ob_.shape[1]
    dense_b, dense_s = eng.boxes_all[:B].cpu(), eng.scores_all[:B].cpu()
    iou = _box_iou(dense_b, ob_)
    assert float(iou.min()) >= 0.999, float(iou.min())
    assert float((dense_s - os_[..., 0]).abs().max()) < 1e-5
    # candidates: the k best scores match the oracle's decoding module (anchors with scores closer than fp32 round-off may swap
    # places, so rows are compared at the anchor the engine picked), FLAME vectors within 1e-4, boxes IoU >= 0.999
    rb, rs, rf = oracle(x, k=min(1000, A))
    k = rb.shape[1]
    assert (scores[:, :k, 0].cpu() - rs[..., 0]).abs().max() < 1e-5
    idx = eng.idx[:B, :k].cpu().long()
    of_at = torch.stack([of_[b, idx[b]] for b in range(B)])
    ob_at = torch.stack([ob_[b, idx[b]] for b in range(B)])
    rel = (flame[:, :k].cpu() - of_at).abs() / (of_at.abs() + 1.0)
    ch_err = rel.amax(dim=(0, 1))
    worst = int(ch_err.argmax())
    wi = (rel[..., worst]).flatten().argmax()
    print(f"[fp32 parity mode] worst channel {worst}: rel {float(ch_err[worst]):.3e}; got {float(flame[:, :k, worst].cpu().flatten()[wi]):.6g} ref {float(of_at[..., worst].flatten()[wi]):.6g}; "
          f"max rel excluding scale(412): {float(ch_err[:412].max()):.3e}")
    # scale = exp(x)/0.05*stride amplifies the fp32 round-off of its logit by |x| ~ 10: judged on the logit (1e-4 absolute per unit)
    assert float(ch_err[:412].max()) < 1e-4, float(ch_err[:412].max())
    dlog = (torch.log(flame[:, :k, 412].cpu()) - torch.log(of_at[..., 412])).abs()
    assert float(dlog.max()) < 2e-3, float(dlog.max())
    assert float(_box_iou(boxes[:, :k].cpu(), ob_at).min()) >= 0.999
    swapped = float((idx != torch.stack([po_idx for po_idx in [torch.sort(os_[b, :, 0], descending=True, stable=True).indices[:k] for b in range(B)]])).float().mean())
    print(f"[fp32 parity mode] min IoU {float(iou.min()):.6f}, max FLAME rel err {float(rel.max()):.2e}, near-tie order swaps {swapped:.4f}")
    eng.close()


def test_fused_detect_matches_staged_pipeline(gpu_lib, flame_model):
    """vgh_detect (one asynchronous call, head count on the device) against the staged pipeline: oracle post-processing of the
    engine's own candidates, direct FLAME decode of the survivors with the per-image un-pad, and scipy's calculate_rpy.
    Also through the chunked path (arena smaller than the batch) and with a head capacity smaller than the survivors."""
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer
    from oracle import flame_oracle as fo
    from oracle import postproc_oracle as po

    S, B = 256, 5
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(11)).to(_dev())
    unpad = torch.tensor([[0.0, 12.0, 0.8], [7.0, 0.0, 1.25], [0.0, 0.0, 1.0], [3.0, 5.0, 0.5], [10.0, 20.0, 2.0]], device=_dev())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=1024)
    for arena in (None, 2):
        eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=5, arena_batch=arena)
        boxes, scores, flame = [t.clone() for t in eng.model(x)]
        conf = float(scores[:, 30, 0].max())
        det = eng.detect(x, confidence_threshold=conf, flame=fl, unpad=unpad)
        ref = po.postprocess_batched(boxes.cpu(), scores.cpu(), flame.cpu(), conf, 0.5)
        counts = det.counts.cpu().tolist()
        assert counts == [r[0].shape[0] for r in ref] and sum(counts) >= B
        assert det.num_heads == sum(counts)
        assert det.head_image.cpu().tolist() == [b for b, c in enumerate(counts) for _ in range(c)]
        params = torch.cat([r[2] for r in ref])
        for b in range(B):
            n = counts[b]
            assert torch.equal(det.boxes[b, :n].cpu(), ref[b][0]) and torch.equal(det.flame_params[b, :n].cpu(), ref[b][2])
        # FLAME: the indirect device-count path must be bit-identical to the direct decode of the same rows
        _, rot, proj = fl.decode(params.to(_dev()), unpad=unpad[det.head_image], shape_live=eng.program.shape_c, expr_live=eng.program.expr_c, want_vertices=False)
        assert torch.equal(det.vertices_3d, proj)
        # and equal to the f64 oracle within the fp32 bar
        _, _, q = fo.reproject(fo.FlameConstants(flame_model, torch.float64), params.double())
        up = unpad[det.head_image].cpu().double()
        q[:, :, 0] -= up[:, None, 0]
        q[:, :, 1] -= up[:, None, 1]
        q = q / up[:, None, 2:3]
        assert float((det.vertices_3d.cpu().double() - q).abs().max()) < 2e-6 * max(1000.0, float(q.abs().max()))
        # head pose: closed form in the kernel vs scipy (utils.py:146-151)
        want = np.array([list(fo.calculate_rpy(p[403:409])) for p in params])
        got = det.head_pose.cpu().numpy()
        d = np.abs(((got - want) + 180.0) % 360.0 - 180.0)
        assert d.max() < 2e-3, d.max()
        eng.close()
    # capacity smaller than the number of survivors: the list is truncated, never overrun
    eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=5)
    small = FLAMELayer(model=flame_model, device=_dev(), max_heads=4)
    det = eng.detect(x, confidence_threshold=conf, flame=small, unpad=unpad)
    assert det.num_heads == 4 and torch.equal(det.vertices_3d, proj[:4])
    eng.close()


def test_select_as_one_launch_with_many_images_and_empty_ones(gpu_lib, flame_model):
    """r06: NMS + compaction + head list are ONE launch (nms_select_kernel: block per image, the LAST block to finish builds the head list behind a device-memory
    ticket).  70 images -- more blocks than one round of anything -- with a threshold that leaves about half of them without a survivor, twice through the same
    detector (the ticket is back at zero): counts, the image-major head list and every kept row against the oracle's post-processing of the engine's candidates."""
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer
    from oracle import postproc_oracle as po

    S, B = 160, 70
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(23)).to(_dev())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=B * 100)
    eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=8)
    boxes, scores, flame = [t.clone() for t in eng.model(x)]
    conf = float(scores[:, 0, 0].median())  # the best score of about half of the images falls below it
    ref = po.postprocess_batched(boxes.cpu(), scores.cpu(), flame.cpu(), conf, 0.5)
    want = [r[0].shape[0] for r in ref]
    assert 0 in want and max(want) >= 1
    for _ in range(2):
        det = eng.detect(x, confidence_threshold=conf, flame=fl)
        assert det.counts.cpu().tolist() == want and det.num_heads == sum(want)
        assert det.head_image.cpu().tolist() == [b for b, c in enumerate(want) for _ in range(c)]
        for b in range(B):
            n = want[b]
            assert torch.equal(det.boxes[b, :n].cpu(), ref[b][0]) and torch.equal(det.flame_params[b, :n].cpu(), ref[b][2])
            assert not bool(det.boxes[b, n:].any()) and not bool(det.flame_params[b, n:].any())  # rows behind the count are zeros
    eng.close()


def test_detect_batch_facade_equals_single_image_calls(gpu_lib, flame_model):
    """HeadDetector.detect_batch (fused device path for every image) returns, per image, what HeadDetector.__call__ returns."""
    from head_detector_amd.detector import HeadDetector

    det = HeadDetector("vgg_heads_m", 320, flame_model=flame_model, weights="synthetic", seed=4, max_batch=3)
    rng = np.random.default_rng(1)
    imgs = [rng.integers(0, 256, shape, dtype=np.uint8) for shape in ((240, 300, 3), (320, 320, 3), (400, 250, 3))]
    image, _ = det._preprocess(imgs[0])
    conf = float(det._process(image)[1][0, 15, 0])
    singles = [det(im, confidence_threshold=conf) for im in imgs]
    batch = det.detect_batch(imgs, confidence_threshold=conf)
    assert len(batch) == 3 and sum(len(r.heads) for r in batch) >= 1
    for one, many in zip(singles, batch):
        assert len(one.heads) == len(many.heads)
        for a, b in zip(one.heads, many.heads):
            assert (a.bbox.x, a.bbox.y, a.bbox.w, a.bbox.h) == (b.bbox.x, b.bbox.y, b.bbox.w, b.bbox.h)
            assert float(a.score) == float(b.score)
            assert np.array_equal(a.vertices_3d, b.vertices_3d)
            assert torch.equal(a.flame_params.to_3dmm_tensor(), b.flame_params.to_3dmm_tensor())
            np.testing.assert_allclose(np.array(a.head_pose), np.array(b.head_pose), atol=2e-3)
    assert det.detect_batch([]) == []
    with pytest.raises(ValueError):
        det.detect_batch(imgs + imgs)


# ---- result-side consumers on the device: Sim3DR rasteriser, PNCC, refined_head_bbox -----------------------
def test_rasterizer_bit_exact_vs_reference_vectors_and_oracle(gpu_lib):
    """csrc/raster.hip through the C ABI (vgh_rasterize) against (i) images produced by the reference's own C++ (golden (f)),
    (ii) the pinned oracle on fresh meshes incl. heavy clipping, degenerate triangles and a 1-channel image, (iii) the live
    oracle/_ref build when present.  Bit-exact: integer/byte output."""
    from head_detector_amd.pncc import rasterize
    from oracle import build_ref
    from oracle import raster_oracle as ro

    g = golden("raster_ref.npz")
    for i in range(3):
        out = rasterize(g[f"m{i}_ver"], g[f"m{i}_tri"], g[f"m{i}_col"], bg=g[f"m{i}_bg"].copy(), reverse=bool(g[f"m{i}_rev"]))
        assert np.array_equal(out, g[f"m{i}_out"]), f"golden mesh {i}"
    ref = build_ref.load()
    for seed in (41, 42, 43, 44):
        ver, tri, col = ro.random_mesh(seed, n_side=8 + seed % 7, size=150 + 40 * (seed % 3), centre=(10 + 40 * (seed % 4), 30 + 20 * (seed % 3)), depth_scale=25)
        tri = np.concatenate([tri, np.array([[0, 0, 1], [2, 2, 2]], dtype=np.int32)])  # degenerate triangles: zero area -> inverDeno = 0
        bg = np.random.default_rng(seed).integers(0, 256, (90, 140, 3), dtype=np.uint8)
        for rev in (False, True):
            want = ro.rasterize(ver, tri, col, bg, reverse=rev)
            assert np.array_equal(rasterize(ver, tri, col, bg=bg.copy(), reverse=rev), want), (seed, rev)
            if ref is not None:
                img, zb = bg.copy(), np.zeros((90, 140), dtype=np.float32) - 1e8
                ref.ref_rasterize(img.ctypes.data, ver.ctypes.data, tri.ctypes.data, col.ctypes.data, zb.ctypes.data, tri.shape[0], 90, 140, 3, 1.0, int(rev))
                assert np.array_equal(want, img)
    ver, tri, col = ro.random_mesh(50)
    out1 = rasterize(ver, tri, col[:, :1].copy(), height=128, width=128, channel=1)
    assert np.array_equal(out1, ro.rasterize(ver, tri, col[:, :1], np.zeros((128, 128, 1), np.uint8)))
    assert np.array_equal(rasterize(ver, tri[:0], col, bg=np.full((8, 8, 3), 7, np.uint8)), np.full((8, 8, 3), 7, np.uint8))  # empty mesh


def test_pncc_processor_and_head_bbox_vs_reference_vectors(gpu_lib):
    """PNCCProcessor (vgh_pncc_render) + refined_head_bbox (vgh_refined_head_bbox) against the outputs of the reference's own
    pncc_processor.py / utils.py (golden (f)), including the in-place z negation and the empty-heads case."""
    from types import SimpleNamespace

    from head_detector_amd.pncc import MeshAssets, PNCCProcessor, compute_ncc_color_codes, refined_head_bbox

    g = golden("raster_ref.npz")
    assets = MeshAssets(g["full_faces"], g["v_template"], g["head_w_ears"], g["head_indices"])
    proc = PNCCProcessor(assets)
    assert np.array_equal(proc.triangles, g["pncc_triangles"]) and np.array_equal(proc.colors, g["pncc_colors"])
    assert np.array_equal(compute_ncc_color_codes(g["v_template"], g["head_w_ears"]), g["pncc_colors"])
    image = np.zeros(tuple(g["image_shape"]), dtype=np.uint8)
    heads = [SimpleNamespace(vertices_3d=v.copy()) for v in g["heads"]]
    img = proc(image, heads)
    assert img.dtype == np.uint8 and np.array_equal(img, g["pncc"])
    assert np.array_equal(np.stack([h.vertices_3d for h in heads]), g["heads_after"])
    assert not proc(image, []).any()
    boxes = refined_head_bbox(g["heads"], g["head_indices"])
    assert [(b.x, b.y, b.w, b.h) for b in boxes] == [tuple(int(q) for q in bb) for bb in g["bboxes"]]
    one = refined_head_bbox(g["heads"][1], g["head_indices"])
    assert (one.x, one.y, one.w, one.h) == tuple(int(q) for q in g["bboxes"][1])


def test_get_pncc_through_the_facade(gpu_lib, flame_model):
    """HeadDetector(..., mesh_assets=...)(image).get_pncc() = the oracle's PNCC of the heads the detector returned; without
    assets it raises FileNotFoundError (the reference would fail to np.load its bundled files the same way)."""
    from head_detector_amd.detector import HeadDetector
    from head_detector_amd.pncc import MeshAssets
    from oracle import raster_oracle as ro

    V = 5023
    rng = np.random.default_rng(3)
    faces = np.asarray(flame_model["f"]).astype(np.int64)
    subset = np.sort(rng.choice(V, 3000, replace=False))
    assets = MeshAssets(faces, np.asarray(flame_model["v_template"], dtype=np.float64), subset, subset[:500])
    det = HeadDetector("vgg_heads_m", 320, flame_model=flame_model, weights="synthetic", seed=4, mesh_assets=assets)
    img = rng.integers(0, 256, (300, 320, 3), dtype=np.uint8)
    image, _ = det._preprocess(img)
    conf = float(det._process(image)[1][0, 6, 0])
    res = det(img, confidence_threshold=conf)
    assert len(res.heads) >= 1
    before = [h.vertices_3d.copy() for h in res.heads]
    got = res.get_pncc()
    tri = ro.pncc_triangles(faces, subset)
    col = ro.compute_ncc_color_codes(np.asarray(flame_model["v_template"], dtype=np.float64), subset)
    want = ro.pncc_image(img.shape, [v.copy() for v in before], tri, col)
    assert got.shape == img.shape and np.array_equal(got, want)
    assert all(np.array_equal(h.vertices_3d[:, 2], -b[:, 2]) for h, b in zip(res.heads, before))
    with pytest.raises(FileNotFoundError):
        HeadDetector("vgg_heads_m", 320, flame_model=flame_model, weights="synthetic", seed=4)(img, confidence_threshold=conf).get_pncc()


def test_overlap_mode_is_race_free_and_identical(gpu_lib, flame_model):
    """Throughput mode (vgh_detector_set_overlap): the select half of batch s runs on the detector's side stream while the network
    of batch s+1 runs on the engine stream.  Results must be bit-identical to the stream-ordered mode, also when the next batch's
    network + candidate stages are queued before the previous select is consumed."""
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer

    S, B = 256, 4
    g = torch.Generator().manual_seed(23)
    xa = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(_dev())
    xb = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(_dev())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=1024)
    eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=5)
    conf = float(eng.model(xa)[1][:, 25, 0].max())

    def snap(det):
        return [t.clone() for t in (det.boxes, det.scores, det.flame_params, det.counts, det.vertices_3d, det.head_pose, det.head_image)]

    ref_a, ref_b = snap(eng.detect(xa, confidence_threshold=conf, flame=fl)), snap(eng.detect(xb, confidence_threshold=conf, flame=fl))
    assert int(ref_a[3].sum()) > 0 and not torch.equal(ref_a[0], ref_b[0])
    # the references again through the EAGER candidate stage (detect() gathers lazily since r06: the survivors' FLAME vectors straight from the prediction buffers)
    for x, ref in ((xa, ref_a), (xb, ref_b)):
        eng.forward_candidates(x)
        det = eng.select(B, confidence_threshold=conf, flame=fl)
        eng.join()  # (select() is queued on the engine's stream and returns at once: the caller's stream waits here -- detect() does that itself)
        for r, q in zip(ref, snap(det)):
            assert torch.equal(r, q)
    eng.set_overlap(True)
    for it in range(6):
        lazy = bool(it & 1)  # lazy: select(a) reads the prediction buffers on the side stream while the next forward is already queued -- the guard has to sit behind it
        eng.forward_net(xa)
        eng.candidates(B, lazy_flame=lazy)
        da = eng.select(B, confidence_threshold=conf, flame=fl)  # side stream
        eng.forward_net(xb)                                      # next batch's network queued while select(a) may still run
        eng.candidates(B, lazy_flame=lazy)                       # waits for select(a) before refilling the candidate buffers
        eng.join()
        got_a = snap(da)
        db = eng.select(B, confidence_threshold=conf, flame=fl)
        eng.join()
        got_b = snap(db)
        for r, q in zip(ref_a, got_a):
            assert torch.equal(r, q)
        for r, q in zip(ref_b, got_b):
            assert torch.equal(r, q)
    # detect() keeps its stream-ordered contract in overlap mode
    for r, q in zip(ref_a, snap(eng.detect(xa, confidence_threshold=conf, flame=fl))):
        assert torch.equal(r, q)
    # tune_overlap: both settings timed on this engine, the faster left on, results unchanged
    t = eng.tune_overlap(xa, flame=fl, forwards=3, confidence_threshold=conf)
    assert set(t) == {"overlapped", "serial"} and all(v > 0 for v in t.values()) and eng._overlap == (t["overlapped"] <= t["serial"])
    for r, q in zip(ref_b, snap(eng.detect(xb, confidence_threshold=conf, flame=fl))):
        assert torch.equal(r, q)
    eng.set_overlap(False)
    eng.close()


def test_output_slots_and_gatherer_pipeline_without_host_syncs(gpu_lib, flame_model):
    """The N>1 step of bench.py on one GPU (a one-rank gatherer copies instead of calling RCCL; everything else is the same code):
    two engine output slots, select on the side stream, the communication stream joined to it (not the engine stream), the previous
    batch read only after the next one has been queued.  Every batch must come out bit-identical to a stream-ordered detect()."""
    from head_detector_amd.dist import DetectionGatherer
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer

    S, B, V = 256, 4, 5023
    g = torch.Generator().manual_seed(29)
    xs = [torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(_dev()) for _ in range(3)]
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=1024)
    eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=5)
    conf = float(eng.model(xs[0])[1][:, 25, 0].max())

    def snap(det):
        n = det.num_heads
        return [t.clone() for t in (det.boxes, det.scores, det.flame_params, det.counts)], det.vertices_cap[:n].clone(), n

    refs = [snap(eng.detect(x, confidence_threshold=conf, flame=fl)) for x in xs]
    assert all(r[2] > 0 for r in refs)
    rows = max(r[2] for r in refs) + 3
    for overlap in (True, False):
        eng.set_overlap(overlap)
        slots = [eng.new_output_slot(fl) for _ in range(2)]
        gat = DetectionGatherer(B, eng.keep_k, fl.num_vertices, vertex_rows=rows, device=_dev())
        ready = [torch.cuda.Event() for _ in range(2)]
        seen = []

        def collect(step):
            out = gat.result(step & 1)
            seen.append((step, [t.clone() for t in (out.boxes, out.scores, out.flame_params, out.counts)], out.vertex_slabs[0].clone(), int(out.n_heads_per_rank[0])))

        for step in range(7):
            s = step & 1
            gat.wait_slot_free(s, eng.stream)
            eng.forward_net(xs[step % 3])
            eng.candidates(B)
            det = eng.select(B, confidence_threshold=conf, flame=fl, slot=slots[s])
            if overlap:
                eng.join_into(gat.stream)
                ev = None
            else:
                ev = ready[s]
                ev.record(eng.stream)
            gat.submit(s, det.boxes, det.scores, det.flame_params, det.counts, det.n_heads, det.vertices_cap, ev)
            if step >= 1:
                collect(step - 1)  # read late: batch `step` is already queued behind it
        collect(6)
        torch.cuda.synchronize()
        assert len(seen) == 7
        for step, slabs, verts, n in seen:
            (rs, rv, rn) = refs[step % 3]
            assert n == rn
            for a, b in zip(rs, slabs):
                assert torch.equal(a, b), (overlap, step)
            assert torch.equal(verts[:n], rv), (overlap, step)
    eng.set_overlap(False)
    eng.close()


def test_results_are_owned_by_the_caller_and_flame_scratch_is_stream_safe(gpu_lib, flame_model):
    """detect()/model() hand out fresh tensors (the reference's TorchScript module does): a result held across the next call keeps
    its contents.  And one FLAME handle driven from two streams back to back (its coefficient scratch is per handle) must give
    each stream its own result."""
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer

    S, B = 256, 2
    g = torch.Generator().manual_seed(31)
    xa = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(_dev())
    xb = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(_dev())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=1024)
    eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=5)
    ma = eng.model(xa)
    keep = [t.clone() for t in ma]
    conf = float(ma[1][:, 25, 0].max())
    da = eng.detect(xa, confidence_threshold=conf, flame=fl)
    snap = [t.clone() for t in (da.boxes, da.scores, da.flame_params, da.counts, da.vertices_3d, da.head_pose)]
    mb = eng.model(xb)
    db = eng.detect(xb, confidence_threshold=conf, flame=fl)
    assert not torch.equal(mb[0], ma[0]) and not torch.equal(db.boxes, da.boxes)
    for a, b in zip(keep, ma):
        assert torch.equal(a, b)
    for a, b in zip(snap, (da.boxes, da.scores, da.flame_params, da.counts, da.vertices_3d, da.head_pose)):
        assert torch.equal(a, b)
    va = eng.detect(xa, confidence_threshold=conf, flame=fl, reuse_outputs=True)  # the aliasing variant says so
    vb = eng.detect(xb, confidence_threshold=conf, flame=fl, reuse_outputs=True)
    assert va.boxes.data_ptr() == vb.boxes.data_ptr() and torch.equal(va.boxes, db.boxes)
    eng.close()

    pa, pb = torch.randn(700, 413, device=_dev()) * 0.3, torch.randn(700, 413, device=_dev()) * 0.3
    ref_a, ref_b = fl.decode(pa)[2].clone(), fl.decode(pb)[2].clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(5):
        with torch.cuda.stream(s1):
            oa = fl.decode(pa)[2]
        with torch.cuda.stream(s2):
            ob = fl.decode(pb)[2]
        torch.cuda.synchronize()
        assert torch.equal(oa, ref_a) and torch.equal(ob, ref_b)


def test_lane_and_side_streams_are_measured_to_overlap(gpu_lib):
    """HIP maps streams onto 4 hardware queues; two streams on one queue serialise (the same 2-lane forward: 5.1 vs 7.1 ms).  The
    net's lane stream and the detector's side stream are picked by measurement, so they must overlap with the engine's stream (and
    with each other) for the 1st, 2nd, 3rd ... engine of a process alike, whatever was created and destroyed before."""
    import ctypes as C

    from head_detector_amd.engine import VGHeadsEngine

    lib = gpu_lib
    junk = []
    for round_ in range(4):
        eng = VGHeadsEngine("vgg_heads_m", image_size=128, max_batch=4, seed=1)
        eng.set_overlap(True)
        eng.set_split(2)
        x = torch.randint(0, 256, (4, 128, 128, 3), dtype=torch.uint8).to(_dev())
        eng.forward_candidates(x)
        eng.join()
        torch.cuda.synchronize()
        out = (C.c_void_p * 4)()
        assert lib.vgh_detector_streams(eng._det, eng._sp(), out) == 0
        main, lane1, side = eng._sp(), out[0], out[3]
        assert lane1 and side
        assert lib.vgh_streams_overlap(main, lane1) == 1, round_
        assert lib.vgh_streams_overlap(main, side) == 1, round_
        assert lib.vgh_streams_overlap(lane1, side) == 1, round_
        comm = eng.acquire_stream()
        assert lib.vgh_streams_overlap(main, comm.cuda_stream) == 1 and lib.vgh_streams_overlap(lane1, comm.cuda_stream) == 1
        junk.append(torch.cuda.Stream())  # perturb the runtime's queue assignment between rounds
        eng.close()


def test_gatherer_collectives_on_rccl_single_rank(gpu_lib):
    """The exact torch.distributed calls of DetectionGatherer (async all_gather into views, gather, work.wait() on a communication
    stream acquired from the library) on the real RCCL backend -- a one-rank group is all a 1-GPU box offers, but it goes through the
    same ProcessGroupNCCL code as N ranks."""
    import os
    import socket

    import torch.distributed as dist

    from head_detector_amd.dist import DetectionGatherer

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        B, keep, V = 4, 100, 5023
        import ctypes as C

        got = C.c_void_p()
        main = torch.cuda.current_stream().cuda_stream
        avoid = (C.c_void_p * 1)(main)
        assert gpu_lib.vgh_stream_acquire(0, avoid, 1, C.byref(got)) == 0
        comm = torch.cuda.ExternalStream(got.value, device=_dev())
        # RCCL's own stream comes from torch's pool: steer it off the compute stream's hardware queue before the first collective,
        # then the measurement must agree with itself (a stream that shares the queue is seen as sharing it)
        from head_detector_amd.dist import collective_shares_queue_with, steer_collective_stream

        clear = steer_collective_stream([torch.cuda.current_stream()])
        assert clear == (not collective_shares_queue_with(torch.cuda.current_stream()))
        g = DetectionGatherer(B, keep, V, vertex_rows=16, device=_dev(), stream=comm, always_collective=True)
        assert g.collective
        gen = torch.Generator().manual_seed(5)
        for step in range(5):
            slot = step & 1
            g.wait_slot_free(slot)
            boxes, scores, flame = torch.rand(B, keep, 4, generator=gen).to(_dev()), torch.rand(B, keep, generator=gen).to(_dev()), torch.rand(B, keep, 413, generator=gen).to(_dev())
            counts = torch.randint(0, 4, (B,), generator=gen).int().to(_dev())
            verts = torch.rand(16, V, 3, generator=gen).to(_dev())
            ev = torch.cuda.Event()
            ev.record()
            g.submit(slot, boxes, scores, flame, counts, None, verts, ev)
            out = g.result(slot)
            assert torch.equal(out.boxes, boxes) and torch.equal(out.scores, scores) and torch.equal(out.flame_params, flame) and torch.equal(out.counts, counts)
            assert int(out.n_heads_per_rank[0]) == int(counts.sum()) and torch.equal(out.vertex_slabs[0], verts)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant,B,prec", [("vgg_heads_l", 1, "bf16"), ("vgg_heads_m", 2, "bf16"), ("vgg_heads_m", 1, "fp32"), ("vgg_heads_l", 1, "fp16x3")])
def test_latency_lanes_are_invisible(gpu_lib, flame_model, variant, B, prec):
    """Single-image engines (r06, arch.schedule_latency): each head's ops right behind its pyramid level, on the executor's lane streams with exact one-op
    dependencies (vgh_op_desc.lane bits 8+).  Against the same engine built WITHOUT the pass: every network output and every detection bit for bit, eagerly and as a
    replayed hipGraph (whose capture follows the cross-lane events), call after call; the reordered program is still a valid serial order (the per-op profiler runs it)."""
    from head_detector_amd import arch
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer

    S = 320
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(B + 40)).to(_dev())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=256)
    ref_eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=9, precision=prec, latency_lanes=False)
    assert all((op.get("lane", 0) >> 8) == 0 for op in ref_eng.program.ops)
    ref = [t.clone() for t in ref_eng.model(x)]
    conf = float(ref[1][:, 6, 0].max())
    d0 = ref_eng.detect(x, confidence_threshold=conf, flame=fl)
    ref_det = [t.clone() for t in (d0.boxes, d0.counts, d0.vertices_3d)]
    ref_eng.close()
    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=9, precision=prec)
    P = eng.program
    deps = [(i, (op["lane"] >> 8) - 1) for i, op in enumerate(P.ops) if op.get("lane", 0) >> 8]
    assert len(deps) >= 9 and all(0 <= d < i and (P.ops[d].get("lane", 0) & 255) != (P.ops[i]["lane"] & 255) for i, d in deps)  # every wait names an EARLIER op on ANOTHER lane
    assert {op.get("lane", 0) & 255 for op in P.ops} == {0, 1, 2, 3}
    names = [op["name"] for op in P.ops]
    assert names.index("heads.head1.pose_stem|bbox_stem") == names.index("neck.neck2.blocks.conv3") + 1 < names.index("neck.neck3.conv")
    for graph in (False, True, False):
        for _ in range(3):
            got = eng.model(x, use_graph=graph)
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), graph
        d = eng.detect(x, confidence_threshold=conf, flame=fl, use_graph=graph)
        assert all(torch.equal(a, b) for a, b in zip((d.boxes, d.counts, d.vertices_3d), ref_det)), graph
    rows = eng.profile_ops(x)
    assert len(rows) == len(P.ops) and all(r["ms"] >= 0 for r in rows)
    assert all(torch.equal(a, b) for a, b in zip(eng.model(x), ref))
    eng.close()


def test_batch_split_lanes_are_invisible(gpu_lib, flame_model):
    """vgh_net_set_split: the batch as 2 / 3 / 4 independent sub-batches on the net's lane streams (uneven sizes included) must give
    bit-identical activations, candidates and detections to the single-stream run -- also combined with overlap mode."""
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer

    S, B = 192, 7
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(31)).to(_dev())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=1024)
    eng = VGHeadsEngine("vgg_heads_m", image_size=S, max_batch=B, seed=6, use_tuning=False)  # same tile choice in every mode
    ref = [t.clone() for t in eng.model(x)]
    conf = float(ref[1][:, 12, 0].max())
    d0 = eng.detect(x, confidence_threshold=conf, flame=fl)
    ref_det = [t.clone() for t in (d0.boxes, d0.counts, d0.vertices_3d, d0.head_pose)]
    names = [bf["name"] for bf in eng.program.bufs]
    probe = [names[len(names) // 3], names[-1]]
    ref_bufs = [eng.buffer(nm, B) for nm in probe]
    for ns in (2, 3, 4):
        eng.set_split(ns)
        for overlap in (False, True):
            eng.set_overlap(overlap)
            got = eng.model(x)
            for r, q in zip(ref, got):
                assert torch.equal(r, q), (ns, overlap)
            for nm, r in zip(probe, ref_bufs):
                assert torch.equal(eng.buffer(nm, B), r), (ns, nm)
            d = eng.detect(x, confidence_threshold=conf, flame=fl)
            for r, q in zip(ref_det, (d.boxes, d.counts, d.vertices_3d, d.head_pose)):
                assert torch.equal(r, q), (ns, overlap)
    # a batch smaller than the number of lanes
    eng.set_split(4)
    assert all(torch.equal(r[:2], q) for r, q in zip(ref, eng.model(x[:2].contiguous())))
    eng.set_overlap(False)
    eng.close()


def test_letterbox_kernel_vs_oracle(gpu_lib):
    """vgh_letterbox (LANCZOS4 fixed-point resize + constant border on the GPU) against oracle/letterbox_oracle.py, bit-exact:
    landscape / portrait / square, up- and down-scaling, the identity size, a 4-channel source, extreme aspect ratios.
    (Both restate OpenCV's published algorithm; cv2 itself is absent => this stage is parity-unpinned against the reference.)"""
    from head_detector_amd.letterbox import geometry, letterbox
    from oracle import letterbox_oracle as lo

    rng = np.random.default_rng(5)
    for (h, w), S in (((480, 600), 640), ((600, 480), 640), ((640, 640), 640), ((97, 1300), 640), ((1080, 1920), 640), ((333, 200), 320), ((64, 64), 256),
                      ((720, 1280), 1280)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if (h, w) == (333, 200):
            img[50:80, 20:90] = 255  # saturated block next to a black one: the negative Lanczos lobes must clamp, not wrap
            img[80:110, 20:90] = 0
        want, pad, scale = lo.transform_image(img, S)
        got, gpad, gscale = letterbox(img, S, _dev())
        assert gpad == pad and gscale == scale and geometry(h, w, S)[2:4] == pad
        assert np.array_equal(got.cpu().numpy(), want), (h, w, S)
    rgba = rng.integers(0, 256, (100, 150, 4), dtype=np.uint8)
    got, _, _ = letterbox(rgba, 128, _dev())
    assert np.array_equal(got.cpu().numpy(), lo.transform_image(rgba, 128)[0])
    same = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
    assert np.array_equal(letterbox(same, 128, _dev())[0].cpu().numpy(), same)  # identity size: exact copy
    with pytest.raises(ValueError):
        letterbox(same.astype(np.float32), 128, _dev())


@pytest.mark.parametrize("variant,B,probe,S", [("vgg_heads_m", 32, (0, 13, 31), 640), ("vgg_heads_l", 64, (0, 31, 63), 640), ("vgg_heads_l", 8, (0, 5, 7), 640),
                                               ("vgg_heads_l", 16, (0, 9, 15), 1280), ("vgg_heads_l", 32, (0, 26, 27, 31), 1280), ("vgg_heads_l", 256, (0, 27, 255), 1280)],
                         ids=["m32", "l64", "l8", "l16_1280", "l32_1280_chunked", "l256_1280_configs4_stated_batch"])
def test_full_size_batch_independence_property(gpu_lib, flame_model, variant, B, probe, S):
    """BASELINE configs[1] (VGGHeads_M, B = 32) and configs[2] (VGGHeads_L, B = 64 + FLAME decode: the benchmark line) at 640x640
    with the tuned tile tables, two lanes + overlap, plus the b8 tile bucket; configs[4] (VGGHeads_L @ 1280x1280 CROWD images:
    33 600 anchors, the confidence threshold calibrated so that >= 32 heads per image survive NMS and are decoded through the
    device-side head count) in its own tile bucket (b8, m102400 tiles) at B = 16 and -- B = 32 -- through the chunked arena (a
    1280 activation tensor passes 2 GiB beyond 27 images: the batch runs as 27 + 5), and (r06) at configs[4]'s STATED batch, 256 images
    on one GPU = ten arena chunks (9 x 27 + 13).  The oracle cannot run these sizes in seconds,
    so parity rests on a size-independent property -- images are independent, hence every image's candidates and detections in
    the full batch equal those of the same image run alone (same engine, same tile choices), bit for bit."""
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer

    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(77)).to(_dev())
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=B * 100)
    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
    if S == 1280:
        assert eng.A == 33600 and eng.arena_batch == min(B, 27)
    eng.set_split(2)
    eng.set_overlap(True)
    boxes, scores, flame = [t.clone() for t in eng.model(x)]
    assert torch.isfinite(boxes).all() and torch.isfinite(flame).all()
    conf = float(scores[:, 3, 0].min())
    if S == 1280:  # crowd: bisect the threshold until ~48 heads per image survive
        lo, hi = float(scores.min()), float(scores.max())
        for _ in range(30):
            conf = 0.5 * (lo + hi)
            n = float(eng.detect(x, confidence_threshold=conf).counts.float().mean())
            if abs(n - 48.0) < 3.0:
                break
            lo, hi = (conf, hi) if n > 48.0 else (lo, conf)
    det = eng.detect(x, confidence_threshold=conf, flame=fl)
    counts = det.counts.cpu().tolist()
    full = (det.boxes.clone(), det.flame_params.clone(), det.vertices_3d.clone(), det.head_image.clone())
    assert min(counts) >= 1
    if S == 1280:
        print(f"[1280 crowd] B={B} heads per image: mean {sum(counts) / B:.1f} min {min(counts)} max {max(counts)}; decoded {det.num_heads}")
        assert sum(counts) / B >= 32.0 and det.num_heads == sum(counts) and torch.isfinite(det.vertices_3d).all()
    for i in probe:
        b1, s1, f1 = eng.model(x[i : i + 1].contiguous())
        assert torch.equal(b1[0], boxes[i]) and torch.equal(s1[0], scores[i]) and torch.equal(f1[0], flame[i])
        d1 = eng.detect(x[i : i + 1].contiguous(), confidence_threshold=conf, flame=fl)
        n = counts[i]
        assert int(d1.counts[0]) == n and torch.equal(d1.boxes[0, :n], full[0][i, :n]) and torch.equal(d1.flame_params[0, :n], full[1][i, :n])
        assert torch.equal(d1.vertices_3d, full[2][full[3] == i])
    eng.set_overlap(False)
    eng.close()


def test_bf16_mode_accuracy_vs_fp32_mode(gpu_lib, flame_model):
    """North_star states "bbox IoU >= 0.999, FLAME params / vertices within 1e-4" for the reference's fp32 path; the engine meets it
    in fp32 parity mode (test_fp32_parity_mode_meets_north_star_tolerances).  The bf16 THROUGHPUT mode -- the one bench.py times:
    tuned tiles, two lane streams -- cannot (bf16 storage through ~190 layers), so its deviation from the fp32 mode is measured
    on the benchmark geometry and held to explicit numbers (random weights; matched by anchor on the fp32 mode's NMS survivors)."""
    from head_detector_amd.accuracy import bf16_vs_fp32
    from head_detector_amd.flame import FLAMELayer

    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=256)
    for variant, B, split in (("vgg_heads_m", 2, 2), ("vgg_heads_l", 1, 1)):
        r = bf16_vs_fp32(variant, 640, B, fl, split=split)
        print(f"[bf16 vs fp32 mode] {r}")
        assert r["kept_fp32"] >= 4 and r["missing_in_bf16_topk"] == 0
        assert r["iou_median"] >= 0.999 and r["iou_min"] >= 0.90  # median meets the bar; a near-tie DFL bin can move one box by a few %
        assert r["dense_score_max_abs_err"] < 2e-3
        assert r["param_max_abs_err_live"] < 0.5  # translation is in pixels (|t| ~ 640): 0.5 px
        assert r["log_scale_max_abs_err"] < 5e-2
        assert r["vertex_l2_metric_mean"] < 2e-3 and r["vertex_l2_metric_max"] < 2e-2  # metres in FLAME space (|v| ~ 0.2)
        # the share of the fp32 mode's NMS survivors the bf16 mode also keeps at the same threshold, pinned WITH A TIE MARGIN: survivors whose
        # fp32 score clears the threshold by less than 3 x the dense score error are left out -- the random-weight net's scores sit within a few 1e-4 of each
        # other around any calibrated threshold, so the raw share moved between 0.44 and 0.88 with the summation order of ONE retuned layer (r02 / r03 tables)
        assert 0.0 < r["kept_by_both_frac"] <= 1.0
        if r["kept_clear_of_threshold"] >= 3:
            assert r["kept_by_both_clear_frac"] >= 0.75, r


def test_c_only_create_and_detect(gpu_lib, flame_model, tmp_path):
    """the pipeline from a pack file behind `vgh_create(config)` with no Python in the loop.  tests/c_abi_smoke.c is
    compiled by plain gcc against include/vgh.h + libvgh.so, builds a context from a .vghpack written by head_detector_amd.pack and
    runs one vgh_ctx_detect; its raw outputs must equal the Python engine's (same folded weights, same tile choices) bit for bit.
    Also: the context created through ctypes, a missing pack and an oversize batch fail with messages (per-context error)."""
    import subprocess

    from conftest import ROOT
    from head_detector_amd import _lib, arch, pack
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer

    variant, S, B = "vgg_heads_m", 320, 3
    sd = arch.random_state_dict(variant, 13)
    P = arch.build_program(variant, sd, S)
    # (the grouped transform-branch op among them cannot run a 128-cout tile: the executor falls back to its automatic tile, in both paths alike)
    names = {i: "256x128_w64x64_k1_r3" for i, op in enumerate(P.ops) if op["kind"] == 1 and op["cout_pad"] % 128 == 0 and i % 2 == 0}
    path = str(tmp_path / "m320.vghpack")
    pack.write_pack(path, P, flame_model, names, B)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    x.numpy().tofile(str(tmp_path / "images.u8"))
    unpad = torch.tensor([[3.0 * i, 5.0, 1.0 + 0.25 * i] for i in range(B)], device=_dev())
    # Python path with the same tile choices
    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=B * 100)
    eng = VGHeadsEngine(variant, state_dict=sd, image_size=S, max_batch=B, use_tuning=False)
    for i, n in names.items():
        eng.set_cfg(i, eng.cfg_names().index(n))
    _, sc, _ = eng.model(x.to(_dev()))
    conf = float(sc[:, 6, 0].min())
    det = eng.detect(x.to(_dev()), confidence_threshold=conf, flame=fl, unpad=unpad)
    n_py = det.num_heads
    assert n_py >= B
    # C path
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.join(ROOT, "head_detector_amd")
    cc = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_smoke.c"),
          "-o", exe, "-L" + libdir, "-lvgh", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cc, check=True, capture_output=True, text=True)
    r = subprocess.run([exe, path, str(tmp_path / "images.u8"), str(B), repr(conf), str(tmp_path / "c")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)
    rd = lambda name, dt: np.fromfile(str(tmp_path / f"c.{name}"), dtype=dt)  # noqa: E731
    counts = rd("counts", np.int32)
    assert np.array_equal(counts, det.counts.cpu().numpy()) and int(rd("n_heads", np.int32)[0]) == n_py
    kk = eng.keep_k
    cb, cs, cf = rd("boxes", np.float32).reshape(B, kk, 4), rd("scores", np.float32).reshape(B, kk), rd("flame", np.float32).reshape(B, kk, 413)
    for i in range(B):
        n = int(counts[i])
        assert np.array_equal(cb[i, :n], det.boxes[i, :n].cpu().numpy()) and np.array_equal(cs[i, :n], det.scores[i, :n].cpu().numpy())
        assert np.array_equal(cf[i, :n], det.flame_params[i, :n].cpu().numpy())
    V = fl.num_vertices
    assert np.array_equal(rd("head_image", np.int32)[:n_py], det.head_image.cpu().numpy().astype(np.int32))
    assert np.array_equal(rd("proj", np.float32).reshape(-1, V, 3)[:n_py], det.vertices_3d.cpu().numpy())
    assert np.array_equal(rd("rpy", np.float32).reshape(-1, 3)[:n_py], det.head_pose.cpu().numpy())
    eng.close()
    # the same context through ctypes + per-context error text
    cfg = _lib.Config(device=torch.cuda.current_device(), pack_path=path.encode(), max_batch=B)
    h = C.c_void_p()
    _lib.check(gpu_lib.vgh_create(C.byref(cfg), C.byref(h)))
    info = _lib.CtxInfo()
    _lib.check(gpu_lib.vgh_ctx_get_info(h, C.byref(info)))
    assert (info.variant.decode(), info.image_size, info.num_vertices, info.keep_top_k, info.shape_live) == (variant, S, V, 100, 64)
    o = _lib.DetectOut()
    assert gpu_lib.vgh_ctx_detect(h, x.to(_dev()).data_ptr(), _lib.VGH_IMG_U8_NHWC, B + 5, 0.5, 0.5, C.byref(o), None) != 0
    assert b"batch 8 outside 1..3" in gpu_lib.vgh_ctx_last_error(h)
    gpu_lib.vgh_destroy(h)
    bad = _lib.Config(device=0, pack_path=str(tmp_path / "images.u8").encode(), max_batch=1)
    assert gpu_lib.vgh_create(C.byref(bad), C.byref(h)) != 0 and b"not a readable" in gpu_lib.vgh_last_error()


@pytest.mark.parametrize("n,live", [(1, (128, 64)), (2, (300, 100)), (3, (64, 32)), (4, (128, 64)), (7, (300, 100)), (8, (128, 64)), (5, (128, 64)), (12, (64, 32)), (16, (128, 64)), (17, (300, 100)), (32, (64, 32)), (33, (128, 64)), (100, (300, 100)), (128, (128, 64)), (191, (128, 64)), (192, (300, 100)), (255, (128, 64)), (256, (300, 100)), (257, (128, 64)), (1023, (128, 64)),
                                    (1024, (300, 100)), (1300, (64, 32)), (1100, (300, 100))])
def test_flame_matrix_core_kernel_is_bit_identical_to_valu_kernel(gpu_lib, flame_model, n, live):
    """The FP32-MFMA vertex kernels (v_mfma_f32_32x32x2_f32 = an exact k-ordered fmaf chain; register-fed for small / medium batches,
    LDS-staged 128 x 128 tiles at crowd scale) and the VALU kernel produce the SAME bits for every vertex (unrotated and projected /
    un-padded), for partial head tiles, every live-coefficient split -- so which one runs is a pure speed choice -- and all stay
    within the f64-oracle bar.  Modes of vgh_flame_set_matrix_path: 0 VALU, 1 automatic, 2 register-fed, 3 / 4 / 5 LDS-staged (128- / 64- / 32-head blocks),
    6 / 7 / 8 the component-split tiles (one / two head tiles per block; 8: prologue waves inside the block up to 8 heads)."""
    from head_detector_amd.flame import FLAMELayer
    from oracle import flame_oracle as fo

    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=max(512, n))
    p = fo.synthetic_params(n, seed=n, live_shape=live[0], live_expr=live[1]).to(_dev())
    unpad = torch.tensor([[3.0, 7.0, 1.3]], device=_dev()).expand(n, 3).contiguous()
    outs = {}
    try:
        for mode in (1, 0, 2, 3, 4, 5, 6, 7, 8, 9):  # 8 / 9 (r06): the 16 x 16 x 4 quad tiles forced on (experiments build; measured slower, profiles/r06_flame_quad_tiles.txt) / off; = 1 in the product library
            assert gpu_lib.vgh_flame_set_matrix_path(mode) == 0
            outs[mode] = [t.clone() for t in fl.decode(p, unpad=unpad, shape_live=live[0], expr_live=live[1])]
    finally:
        gpu_lib.vgh_flame_set_matrix_path(1)
    for mode in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        for a, b in zip(outs[mode], outs[0]):
            assert torch.equal(a, b), mode
    _, _, q = fo.reproject(fo.FlameConstants(flame_model, torch.float64), p.cpu().double())
    q[:, :, 0] -= 3.0
    q[:, :, 1] -= 7.0
    q = q / 1.3
    assert float((outs[1][2].cpu().double() - q).abs().max()) < 2e-6 * max(1000.0, float(q.abs().max()))


def test_flame_decode_large_n_equals_chunks(gpu_lib, flame_model):
    """FLAME decode at crowd scale (n = 8192 heads, BASELINE config 5) equals the same heads decoded in chunks of 1000 (different
    HT tile variants and launch shapes must not change a single bit), and every vertex is finite."""
    from head_detector_amd.flame import FLAMELayer

    fl = FLAMELayer(model=flame_model, device=_dev(), max_heads=8192)
    n = 8192
    p = torch.randn(n, 413, generator=torch.Generator().manual_seed(9)).to(_dev())
    p[:, 128:300] = 0
    p[:, 364:400] = 0
    unpad = torch.tensor([[5.0, 7.0, 1.5]], device=_dev()).expand(n, 3).contiguous()
    _, rot, proj = fl.decode(p, unpad=unpad, shape_live=128, expr_live=64, want_vertices=False)
    assert torch.isfinite(proj).all()
    for a in range(0, n, 1000):
        _, r2, q2 = fl.decode(p[a : a + 1000].contiguous(), unpad=unpad[a : a + 1000].contiguous(), shape_live=128, expr_live=64, want_vertices=False)
        assert torch.equal(q2, proj[a : a + 1000]) and torch.equal(r2, rot[a : a + 1000])
    _, _, q1 = fl.decode(p[4242:4243].contiguous(), unpad=unpad[:1].contiguous(), shape_live=128, expr_live=64, want_vertices=False)
    assert torch.equal(q1[0], proj[4242])  # HT = 1 variant
