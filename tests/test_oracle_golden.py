"""CPU: pin the oracle (oracle/) against vectors produced by running the reference's own Python
(tests/golden/make_golden.py) and against the reference's only known-answer fixture (tests/1.json)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import flame_oracle as fo
from oracle import postproc_oracle as po


def test_param_layout_matches_reference():
    g = golden("layout.npz")
    x = torch.arange(413, dtype=torch.float32)[None]
    d = fo.split_3dmm(x)
    for k in ("shape", "expression", "jaw", "rotation", "eyeballs", "neck", "translation", "scale"):
        assert np.array_equal(d[k].numpy().astype(np.int32)[0], g[f"read_{k}"]), k
    perm = fo.join_3dmm(d).numpy().astype(np.int32)[0]
    assert np.array_equal(perm, g["perm"])
    # SURVEY.md 8a row a6': perm = [0..399, 403..408, 400..402, 409..412]
    assert np.array_equal(perm, np.r_[0:400, 403:409, 400:403, 409:413])
    assert bool(g["raised"])
    try:
        fo.split_3dmm(torch.zeros(1, 412))
        assert False, "must raise on a wrong width (head_info.py:51-52)"
    except ValueError:
        pass


def test_rotation_helpers_match_reference():
    g = golden("rotation.npz")
    v6 = torch.from_numpy(g["v6"])
    R = fo.rot_mat_from_6dof(v6).numpy()
    np.testing.assert_allclose(R, g["R"], atol=1e-6)
    # proper rotations
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", R, R), np.broadcast_to(np.eye(3), R.shape), atol=1e-5)
    for i in range(1, 16):  # row 0 is the gimbal-lock case scipy warns about
        np.testing.assert_allclose(np.array(fo.calculate_rpy(v6[i])), g["rpy"][i], atol=1e-3)
    np.testing.assert_allclose([fo.limit_angle(a) for a in g["lim_in"]], g["lim_out"])


def test_flame_decode_matches_reference_glue(flame_model):
    g = golden("flame_decode.npz")
    params = torch.from_numpy(g["params"])
    c32 = fo.FlameConstants(flame_model, torch.float32)
    v, R, p = fo.reproject(c32, params)
    np.testing.assert_allclose(v.numpy(), g["vertices"], atol=2e-6)
    np.testing.assert_allclose(R.numpy(), g["R"], atol=1e-6)
    np.testing.assert_allclose(p.numpy(), g["projected"], rtol=1e-5, atol=1e-3)
    fp = fo.split_3dmm(params)
    np.testing.assert_allclose(fo.flame_forward(c32, fp, zero_rot=False).numpy(), g["forward_rot"], atol=2e-6)
    np.testing.assert_allclose(fo.flame_forward(c32, fp, zero_rot=True, zero_jaw=True).numpy(), g["forward_zero_jaw"], atol=2e-6)
    ev, eR, ep = fo.reproject(c32, torch.zeros(0, 413))
    assert [list(ev.shape), list(eR.shape), list(ep.shape)] == g["empty_shapes"].tolist()
    # float64 arbiter agrees with the float32 reference path to fp32 round-off
    c64 = fo.FlameConstants(flame_model, torch.float64)
    v64, _, p64 = fo.reproject(c64, params.double())
    assert (v64 - v.double()).abs().max() < 5e-6
    rel = ((p64 - p.double()).abs() / (p64.abs() + 1.0)).max()
    assert rel < 1e-5
    # scale clamp case (flame.py:198): row 2 has scale 1e-9 -> clamped to 1e-8
    assert abs(float(params[2, 412]) - 1e-9) < 1e-12
    np.testing.assert_allclose(p[2].numpy(), (torch.matmul(R[2], v[2].T).T * 1e-8 + params[2, 409:412]).numpy(), rtol=1e-5, atol=1e-3)


def test_jaw_only_pose_collapses_to_one_joint(flame_model):
    """with global = neck = eyes = 0 the general lbs equals a single-joint closed form."""
    c = fo.FlameConstants(flame_model, torch.float64)
    params = fo.synthetic_params(3, seed=9, dtype=torch.float64)
    fp = fo.split_3dmm(params)
    v = fo.flame_forward(c, fp, zero_rot=True)
    betas = torch.cat([fp["shape"], fp["expression"]], 1)
    v_shaped = c.v_template[None] + torch.einsum("bl,mkl->bmk", betas, c.shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, c.J_regressor)
    Rj = fo.batch_rodrigues(fp["jaw"])
    pf = torch.zeros(3, 36, dtype=torch.float64)
    pf[:, 9:18] = (Rj - torch.eye(3, dtype=torch.float64)).reshape(3, 9)
    v_posed = v_shaped + (pf @ c.posedirs).view(3, -1, 3)
    w = c.lbs_weights[:, 2][None, :, None]
    Jj = J[:, 2][:, None]
    closed = v_posed + w * (torch.einsum("bij,bvj->bvi", Rj, v_posed - Jj) + Jj - v_posed)
    closed[:, :, 2] += 0.05
    assert (closed - v).abs().max() < 1e-12


def test_nms_glue_matches_reference():
    g = golden("nms_glue.npz")
    boxes, scores = torch.from_numpy(g["boxes"]), torch.from_numpy(g["scores"])
    flame = torch.randn(2, 1000, 413, generator=torch.Generator().manual_seed(int(g["flame_seed"])))
    for tag, conf in (("c50", 0.5), ("c02", 0.02), ("c999", 0.999)):
        ob, os_, of = po.nms_reference(boxes, scores, flame, confidence_threshold=conf)
        assert np.array_equal(ob.numpy(), g[f"{tag}_boxes"]), tag
        assert np.array_equal(os_.numpy(), g[f"{tag}_scores"]), tag
        np.testing.assert_allclose(of.sum(1).numpy(), g[f"{tag}_flame_rowsum"], rtol=1e-6)
    assert g["c50_boxes"].shape[0] >= 1 and g["c999_boxes"].shape[0] == 0 and g["c02_boxes"].shape[0] == 100
    # the batched twin returns one result per image and agrees with nms() on image 0
    res = po.postprocess_batched(boxes, scores, flame, 0.5, 0.5)
    assert len(res) == 2 and np.array_equal(res[0][0].numpy(), g["c50_boxes"])


def test_nms_torchvision_semantics():
    # strict '>' on IoU: two boxes with IoU exactly 0.5 both survive at thr 0.5
    b = np.array([[0, 0, 2, 1], [1, 0, 3, 1], [0, 0, 2, 1.0001]], dtype=np.float32)  # IoU(0,1) = 1/3; IoU(0,2) ~ 1
    keep = po.nms_torchvision(b, np.array([0.9, 0.8, 0.7], np.float32), 0.5)
    assert keep.tolist() == [0, 1]
    b2 = np.array([[0, 0, 2, 2], [0, 0, 2, 1]], dtype=np.float32)  # IoU exactly 0.5
    assert po.nms_torchvision(b2, np.array([0.9, 0.8], np.float32), 0.5).tolist() == [0, 1]
    assert po.nms_torchvision(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).tolist() == []
    # degenerate zero-area boxes: 0/0 = nan -> comparison false -> kept (CPU kernel behaviour)
    z = np.zeros((2, 4), np.float32)
    assert po.nms_torchvision(z, np.array([0.9, 0.8], np.float32), 0.5).tolist() == [0, 1]


def test_fixture_1json_pins_layout_and_rigid_stage():
    """The reference's only numeric fixture (yolo_head_training/tests/1.json). Without the licensed FLAME pickle it pins:
    (i) which 6 of the 413 numbers are the rotation (from_3dmm layout, not the to_3dmm one), and
    (ii) rot_mat_from_6dof + the rigid stage: R^T * 3d_vertices - 0.05 z is the un-posed mesh, which must sit within a few mm of
    v_template (blendshape offsets are mm-scale), and
    (iii) the DAD-style projection of dataset_parsing.py:183-188 reproduces projected_vertices from 3d_vertices."""
    g = golden("fixture_1json.npz")
    vt = golden("flame_decode.npz")["v_template"].astype(np.float64)
    p = torch.from_numpy(g["params"])[None]
    v3 = g["vertices_3d"].astype(np.float64)
    fp = fo.split_3dmm(p)
    R = fo.rot_mat_from_6dof(fp["rotation"]).numpy()[0]
    unposed = v3 @ R - np.array([0, 0, 0.05])  # R^T v
    err_read = np.abs(unposed - vt).mean()
    R_wrong = fo.rot_mat_from_6dof(p[:, 400:406]).numpy()[0]  # to_3dmm layout would put rotation first
    err_wrong = np.abs(v3 @ R_wrong - np.array([0, 0, 0.05]) - vt).mean()
    assert err_read < 5e-3 and err_wrong > 10 * err_read, (err_read, err_wrong)
    s, t = float(fp["scale"][0, 0]), fp["translation"][0].numpy()
    proj = ((v3 * (s + 1.0) + np.array([t[0], t[1], 0.0])) + 1.0) / 2.0 * 256.0
    assert np.abs(proj[:, :2] - g["projected_vertices"]).max() < 1e-3


def test_ndfl_decode_anchor_order_and_permutation():
    torch.manual_seed(0)
    B = 2
    levels = []
    for s in (4, 2, 1):
        levels.append((torch.randn(B, 68, s, s), torch.randn(B, 1, s, s), torch.randn(B, 413, s, s)))
    boxes, scores, flame = po.ndfl_decode(levels, strides=(8, 16, 32))
    A = 16 + 4 + 1
    assert boxes.shape == (B, A, 4) and scores.shape == (B, A, 1) and flame.shape == (B, A, 413)
    # anchor 5 of level 0 = (y=1, x=1): centre (1.5, 1.5) * 8
    reg = levels[0][0][0, :, 1, 1].view(4, 17)
    d = (torch.softmax(reg, 1) * torch.arange(17.0)).sum(1)
    exp = torch.stack([1.5 - d[0], 1.5 - d[1], 1.5 + d[2], 1.5 + d[3]]) * 8
    assert torch.allclose(boxes[0, 5], exp, atol=1e-5)
    O = levels[0][2][0, :, 1, 1]
    T = flame[0, 5]
    assert torch.equal(T[:400], O[:400]) and torch.equal(T[400:406], O[403:409]) and torch.equal(T[406:409], O[400:403])
    assert torch.allclose(T[409:411], O[409:411] + 12.0) and T[411] == O[411] and torch.allclose(T[412], O[412] * 8)
    # level 2 single anchor index 20: stride 32 centre 16
    assert torch.allclose(flame[1, 20, 409:411], levels[2][2][1, 409:411, 0, 0] + 16.0)


# ---- Sim3DR rasteriser / PNCC / refined_head_bbox ------------------------------------------------------
def test_raster_oracle_reproduces_reference_outputs():
    """oracle/raster_oracle.py against vectors produced by the reference's own C++ rasteriser and its own PNCCProcessor /
    refined_head_bbox Python (tests/golden/make_golden.py (f)): bit-exact images, triangle filter, colour codes, boxes, and the
    in-place z negation side effect."""
    from oracle import raster_oracle as ro

    g = golden("raster_ref.npz")
    for i in range(3):
        out = ro.rasterize(g[f"m{i}_ver"], g[f"m{i}_tri"], g[f"m{i}_col"], g[f"m{i}_bg"], reverse=bool(g[f"m{i}_rev"]))
        assert np.array_equal(out, g[f"m{i}_out"]), f"mesh case {i}"
        assert (out != g[f"m{i}_bg"]).any()
    tri = ro.pncc_triangles(g["full_faces"], g["head_w_ears"])
    assert np.array_equal(tri, g["pncc_triangles"]) and 0 < tri.shape[0] < g["full_faces"].shape[0]
    col = ro.compute_ncc_color_codes(g["v_template"], g["head_w_ears"])
    assert np.array_equal(col, g["pncc_colors"])
    heads = [v.copy() for v in g["heads"]]
    img = ro.pncc_image(tuple(g["image_shape"]), heads, tri, col)
    assert np.array_equal(img, g["pncc"]) and img.any()
    assert np.array_equal(np.stack(heads), g["heads_after"])  # z *= -1 in place, like the reference
    for v, bb in zip(g["heads"], g["bboxes"]):
        assert ro.refined_head_bbox(v, g["head_indices"]) == tuple(int(q) for q in bb)


def test_raster_oracle_matches_live_reference_build():
    """The same restatement against oracle/_ref/libsim3dr_ref.so (the reference's rasterize_kernel.cpp compiled by oracle/build_ref.py)
    on fresh random meshes, clipped and unclipped, both row orders."""
    from oracle import build_ref
    from oracle import raster_oracle as ro

    lib = build_ref.load()
    if lib is None:
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/libsim3dr_ref.so is available")
    for seed in (31, 32, 33):
        ver, tri, col = ro.random_mesh(seed, n_side=9 + seed % 5, size=100 + 20 * (seed % 3), centre=(20 + 30 * (seed % 4), 70), depth_scale=25)
        bg = np.random.default_rng(seed).integers(0, 256, (96, 128, 3), dtype=np.uint8)
        for rev in (False, True):
            img = bg.copy()
            zb = np.zeros((96, 128), dtype=np.float32) - 1e8
            lib.ref_rasterize(img.ctypes.data, ver.ctypes.data, tri.ctypes.data, col.ctypes.data, zb.ctypes.data, tri.shape[0], 96, 128, 3, 1.0, int(rev))
            assert np.array_equal(ro.rasterize(ver, tri, col, bg, reverse=rev), img)


def test_letterbox_oracle_sanity_against_pil_lanczos():
    """oracle/letterbox_oracle.py is parity-unpinned against cv2 (absent).  As an independent sanity check of the sampling geometry
    (pixel-centre alignment, tap order, fixed-point scaling) it must stay within 2 grey levels of PIL's Lanczos filter on a smooth
    image -- a different kernel (a = 3 vs OpenCV's 8-tap a = 4), so not a parity claim -- and geometry / border must match
    detector.py:41-50."""
    PIL_Image = pytest.importorskip("PIL.Image")
    from oracle import letterbox_oracle as lo

    yy, xx = np.mgrid[0:200, 0:300]
    img = np.stack([127 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0), xx * 255 / 299, yy * 255 / 199], -1).astype(np.uint8)
    for nw, nh in ((450, 300), (640, 427), (150, 100)):
        a = lo.resize_lanczos4(img, nw, nh).astype(int)
        b = np.array(PIL_Image.fromarray(img).resize((nw, nh), PIL_Image.LANCZOS)).astype(int)
        assert np.abs(a - b).max() <= 2 and np.abs(a - b).mean() < 0.6
    canvas, pad, scale = lo.transform_image(img, 640)
    assert canvas.shape == (640, 640, 3) and pad == (0, (640 - int(200 * 640 / 300)) // 2) and scale == 640 / 300
    assert tuple(canvas[0, 0]) == (127, 0, 0) and tuple(canvas[-1, -1]) == (127, 0, 0)
    assert np.array_equal(lo.resize_lanczos4(img, 300, 200), img)
    tab = lo.resize_tables(300, 640)[1].astype(int).sum(1)
    assert tab.min() >= 2044 and tab.max() <= 2052  # weights sum to ~2048 (no sum correction in cv::resize)


# ======================================================================================================
# rows a5 / a6 / a7 / batched twin: the oracle against vectors produced by RUNNING the reference's own
# yolo_head_dfl_head.py / yolo_head_ndfl_heads.py / yolo_heads.py / yolo_heads_post_prediction_callback.py
# (tests/golden/make_golden_heads.py; super_gradients' conv blocks substituted by the oracle's, weights from the shared seed)
# ======================================================================================================
def _head_fixture(tag):
    import torch

    g = golden("head_decode.npz")
    variant = {"l": "vgg_heads_l", "m": "vgg_heads_m"}[tag]
    t = lambda k: torch.from_numpy(g[f"{tag}_{k}"])  # noqa: E731
    return g, variant, t


@pytest.mark.parametrize("tag", ["l", "m"])
def test_head_wiring_and_activations_vs_reference_run(tag):
    """oracle.net_oracle.DFLHead (+ assemble_flame_channels) == the reference's YoloHeadsDFLHead.forward on the same weights and
    features: stems, towers, branch order, tanh*3 / exp/0.05, zero padding to 300 / 100, concat order (row a5)."""
    import torch

    from head_detector_amd import arch
    from oracle import net_oracle, postproc_oracle as po

    g, variant, t = _head_fixture(tag)
    v = net_oracle.VARIANTS[tag]
    heads = net_oracle.Heads(v, [v["neck"][1][0], v["neck"][2][0], v["neck"][3][0]]).eval()
    sd = {k[len("heads."):]: torch.from_numpy(a) for k, a in arch.random_state_dict(variant, int(g[f"{tag}_seed"])).items() if k.startswith("heads.")}
    heads.load_state_dict(sd, strict=False)
    with torch.no_grad():
        out = heads([t(f"feat{lv}") for lv in range(3)])
    names = {"shape": "shape", "expr": "expression", "rot": "rotation", "jaw": "jaw", "trans": "translation", "scale": "scale"}
    for lv, (reg, cls, flame, raw) in enumerate(out):
        assert torch.allclose(reg, t(f"reg{lv}"), atol=1e-5, rtol=1e-5) and torch.allclose(cls, t(f"cls{lv}"), atol=1e-5, rtol=1e-5)
        for k, ref_k in names.items():
            assert torch.allclose(raw[k], t(f"raw{lv}_{ref_k}"), atol=1e-5, rtol=1e-5), (lv, k)
        assert torch.allclose(flame, t(f"flame{lv}"), atol=1e-5, rtol=1e-5)
        # the activation tail alone, fed with the REFERENCE's raw branch outputs: exact same torch ops -> bit-identical
        fl = po.assemble_flame_channels(*[t(f"raw{lv}_{names[k]}") for k in ("shape", "expr", "rot", "jaw", "trans", "scale")])
        assert torch.equal(fl, t(f"flame{lv}"))
        assert flame.shape[1] == 413 and float(flame[:, v["head"]["shape_out"] : 300].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["l", "m"])
def test_ndfl_decode_topk_postprocess_vs_reference_run(tag):
    """ndfl_decode == YoloHeadsNDFLHeads.forward (row a6 incl. the a6' permutation), decoding_topk == VGGHeadDecodingModule.forward
    (row a7), postprocess_batched == YoloHeadsPostPredictionCallback.__call__ (batched twin of nms), and the reference's own
    reproject_spatial_vertices on the survivors == oracle reproject -- all fed with the reference's per-level head outputs."""
    import torch

    from oracle import flame_oracle as fo
    from oracle import postproc_oracle as po

    g, variant, t = _head_fixture(tag)
    levels = [(t(f"reg{lv}"), t(f"cls{lv}"), t(f"flame{lv}")) for lv in range(3)]
    b, s, f = po.ndfl_decode(levels)
    assert torch.allclose(b, t("boxes"), atol=1e-5, rtol=1e-6) and torch.allclose(s, t("scores"), atol=1e-7) and torch.allclose(f, t("flame"), atol=1e-5, rtol=1e-6)
    # the permutation quirk is visible in the fixture itself: output rows 400:403 hold the head's rot[3:6]... (head order [rot6, jaw3])
    A = b.shape[1]
    head_order = torch.cat([lv[2].flatten(2) for lv in levels], dim=-1).permute(0, 2, 1)  # [B,A,413] in HEAD order, before the fix-up
    assert torch.equal(t("flame")[..., 400:403], head_order[..., 403:406]) and torch.equal(t("flame")[..., 403:406], head_order[..., 406:409])
    assert A == 84
    k = g[f"{tag}_cand_scores"].shape[1]
    cb, cs, cf, _ = po.decoding_topk(t("boxes"), t("scores"), t("flame"), k)
    assert torch.equal(cb, t("cand_boxes")) and torch.equal(cs, t("cand_scores")) and torch.equal(cf, t("cand_flame"))
    res = po.postprocess_batched(t("boxes"), t("scores"), t("flame"), float(g[f"{tag}_post_conf"]), 0.5, pre_nms_max=30, post_nms_max=10)
    consts = fo.FlameConstants(fo.synthetic_flame_model(seed=3), torch.float32)
    for i, (rb_, rs_, rf_) in enumerate(res):
        assert len(rs_) == int(g[f"{tag}_post_counts"][i]) and len(rs_) > 0
        assert torch.equal(rb_, t(f"post{i}_boxes")) and torch.equal(rs_, t(f"post{i}_scores")) and torch.equal(rf_, t(f"post{i}_params"))
        _, _, proj = fo.reproject(consts, rf_)
        ref = t(f"post{i}_v3d")
        assert (proj[:, ::97] - ref).abs().max() <= 2e-4 * ref.abs().max()  # fp32 summation order in lbs
