#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own Python in the build container.

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only mount); the GPU box never runs this -- only the committed vectors travel.

What is real reference code here and what is stubbed:
  * head_detector/head_info.py   imported as-is   -> FlameParams.from_3dmm / to_3dmm_tensor (layout pin)
  * head_detector/utils.py       imported with `cv2` and `torchvision` stubbed in sys.modules
                                  -> real rot_mat_from_6dof, calculate_rpy, limit_angle, nms() glue
                                  (torchvision.ops.boxes.nms := oracle.postproc_oracle.nms_torchvision)
  * head_detector/flame.py       imported with `smplx.lbs.lbs` := oracle.flame_oracle.lbs
                                  -> real FLAMELayer.__init__/forward and reproject_spatial_vertices run on a
                                  synthetic FLAME pickle (the licensed generic_model.pkl is absent)
  * yolo_head_training/tests/1.json  the reference's only numeric fixture, re-packed as float32 .npz
  * head_detector/assets/v_template.npy  data file, re-packed as float32
  * head_detector/pncc_processor.py  imported with `head_detector.Sim3DR` := a module whose ``rasterize`` restates the 20-line
                                  Python wrapper (Sim3DR.py:17-38) and calls the reference's OWN C++ `_rasterize`
                                  (oracle/_ref/libsim3dr_ref.so, built from Sim3DR/lib/rasterize_kernel.cpp by oracle/build_ref.py)
                                  -> real pncc(), compute_ncc_color_codes, PNCCProcessor.__init__/__call__ on SYNTHETIC mesh
                                  assets (np.load patched inside that module; the reference's licensed mesh assets are not packed)
  * head_detector/utils.py:refined_head_bbox  real function with HEAD_INDICES := synthetic subset
  * head_detector/detection_result.py  imported as-is (cv2 stubbed: only the draw helpers touch it) -> real MeshSaver /
                                  PredictionResult.save_meshes on the synthetic faces: the OBJ bytes are the fixture (mesh_obj.npz)
No reference *source* is copied; the vectors are inputs + outputs only.
"""
import importlib.util
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import flame_oracle as fo  # noqa: E402
from oracle import postproc_oracle as po  # noqa: E402


def _stub_modules():
    cv2 = types.ModuleType("cv2")
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.boxes = types.ModuleType("torchvision.ops.boxes")
    tv.ops.boxes.nms = lambda boxes, scores, iou_threshold: torch.from_numpy(po.nms_torchvision(boxes.numpy(), scores.numpy(), iou_threshold))
    sys.modules.update({"torchvision": tv, "torchvision.ops": tv.ops, "torchvision.ops.boxes": tv.ops.boxes})
    smplx = types.ModuleType("smplx")
    smplx.lbs = types.ModuleType("smplx.lbs")
    smplx.lbs.lbs = fo.lbs
    smplx.utils = types.ModuleType("smplx.utils")

    class Struct:
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

    def to_tensor(a, dtype=torch.float32):
        return a.clone().detach().to(dtype) if torch.is_tensor(a) else torch.tensor(a, dtype=dtype)

    def to_np(a, dtype=np.float32):
        if "scipy.sparse" in str(type(a)):
            a = a.todense()
        return np.array(a, dtype=dtype)

    smplx.utils.Struct, smplx.utils.to_tensor, smplx.utils.to_np = Struct, to_tensor, to_np
    sys.modules.update({"smplx": smplx, "smplx.lbs": smplx.lbs, "smplx.utils": smplx.utils})
    # package shell so `from head_detector.x import y` resolves to the reference files without running __init__.py
    pkg = types.ModuleType("head_detector")
    pkg.__path__ = [os.path.join(REF, "head_detector")]
    sys.modules["head_detector"] = pkg


def _load(name):
    spec = importlib.util.spec_from_file_location(f"head_detector.{name}", os.path.join(REF, "head_detector", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"head_detector.{name}"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    _stub_modules()
    head_info = _load("head_info")
    utils = _load("utils")
    flame = _load("flame")
    torch.manual_seed(0)

    # ---- (a) parameter layout -------------------------------------------------------------------------
    x = torch.arange(413, dtype=torch.float32)[None]
    fp = head_info.FlameParams.from_3dmm(x)
    layout = {k: getattr(fp, k).numpy().astype(np.int32)[0] for k in ("shape", "expression", "jaw", "rotation", "eyeballs", "neck", "translation", "scale")}
    perm = fp.to_3dmm_tensor().numpy().astype(np.int32)[0]
    try:
        head_info.FlameParams.from_3dmm(torch.zeros(1, 412))
        raised = False
    except ValueError:
        raised = True
    np.savez_compressed(os.path.join(OUT, "layout.npz"), perm=perm, raised=np.array(raised), **{f"read_{k}": v for k, v in layout.items()})

    # ---- (b) rot_mat_from_6dof / calculate_rpy -----------------------------------------------------
    g = torch.Generator().manual_seed(11)
    v6 = torch.randn(16, 6, generator=g)
    v6[0] = torch.tensor([1.0, 0, 0, 0, 1.0, 0])
    v6[1] = torch.tensor([0.0, 0, 2.0, 0, -3.0, 0])
    R = utils.rot_mat_from_6dof(v6).numpy()
    rpy = []
    for i in range(16):
        p = head_info.FlameParams.from_3dmm(torch.zeros(1, 413))
        p.rotation = v6[i : i + 1]
        rpy.append(list(utils.calculate_rpy(p)))
    lim_in = np.array([-725.0, -540.0, -181.0, -180.0, -10.0, 0.0, 179.9, 180.0, 181.0, 359.0, 540.0, 900.5])
    lim_out = np.array([utils.limit_angle(a) for a in lim_in])
    np.savez_compressed(os.path.join(OUT, "rotation.npz"), v6=v6.numpy(), R=R, rpy=np.array(rpy), lim_in=lim_in, lim_out=lim_out)

    # ---- (c) FLAMELayer + reproject_spatial_vertices on a synthetic pickle ---------------------------
    v_template = np.load(os.path.join(REF, "head_detector", "assets", "v_template.npy"))
    model = fo.synthetic_flame_model(seed=3, v_template=v_template)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "synthetic_flame.pkl")
        with open(path, "wb") as f:
            pickle.dump(model, f)
        layer = flame.FLAMELayer(flame_path=path)
    params = fo.synthetic_params(5, seed=2, live_shape=128, live_expr=64)
    params[1, 400:403] = 0.0  # zero jaw -> batch_rodrigues(0) path
    params[2, 412] = 1e-9  # scale below the clamp (flame.py:198)
    params[3, :300] = 3 * torch.tanh(torch.randn(300, generator=g))  # all 300+100 betas live (dataset-style)
    params[3, 300:400] = 3 * torch.tanh(torch.randn(100, generator=g))
    params[4, 400:403] = torch.tensor([0.6, -0.2, 0.1])  # large jaw rotation
    with torch.no_grad():
        verts, Rm, proj = flame.reproject_spatial_vertices(layer, params, to_2d=False)
        _, _, proj2d = flame.reproject_spatial_vertices(layer, params, to_2d=True)
        fpz = head_info.FlameParams.from_3dmm(params)
        fwd_rot = layer.forward(fpz, zero_rot=False, zero_jaw=False)
        fwd_zero_jaw = layer.forward(fpz, zero_rot=True, zero_jaw=True)
        e_v, e_R, e_p = flame.reproject_spatial_vertices(layer, torch.zeros(0, 413), to_2d=False)
    assert torch.equal(proj2d, proj[..., :2])
    np.savez_compressed(
        os.path.join(OUT, "flame_decode.npz"), v_template=v_template.astype(np.float32), seed=np.array(3), params=params.numpy(), vertices=verts.numpy(),
        R=Rm.numpy(), projected=proj.numpy(), forward_rot=fwd_rot.numpy(), forward_zero_jaw=fwd_zero_jaw.numpy(),
        empty_shapes=np.array([list(e_v.shape), list(e_R.shape), list(e_p.shape)]),
    )

    # ---- (d) nms() glue (image 0 only!) ----------------------------------------------------------------
    boxes, scores = po.synthetic_detections(2, seed=5, mean_heads=4)
    gidx = torch.Generator().manual_seed(7)
    # the reference feeds nms() the decoding module's output: 1000 candidates sorted by score
    idx = torch.stack([po.stable_topk(scores[b, :, 0], 1000) for b in range(2)])
    cb = torch.stack([boxes[b, idx[b]] for b in range(2)])
    cs = torch.stack([scores[b, idx[b]] for b in range(2)])
    cf = torch.randn(2, 1000, 413, generator=gidx)
    outs = {}
    for tag, conf in (("c50", 0.5), ("c02", 0.02), ("c999", 0.999)):
        ob, os_, of = utils.nms(cb, cs, cf, confidence_threshold=conf)
        outs[f"{tag}_boxes"], outs[f"{tag}_scores"], outs[f"{tag}_flame_rowsum"] = ob.numpy(), os_.numpy(), of.sum(1).numpy()
    np.savez_compressed(os.path.join(OUT, "nms_glue.npz"), boxes=cb.numpy(), scores=cs.numpy(), flame_seed=np.array(7), **outs)

    # ---- (e) the reference's own known-answer fixture --------------------------------------------------
    d = json.load(open(os.path.join(REF, "yolo_head_training", "tests", "1.json")))[0]
    np.savez_compressed(
        os.path.join(OUT, "fixture_1json.npz"), params=np.array(d["3dmm_params"], dtype=np.float64), vertices_3d=np.array(d["3d_vertices"], dtype=np.float32),
        projected_vertices=np.array(d["projected_vertices"], dtype=np.float32), bbox=np.array(d["bbox"]), extended_bbox=np.array(d["extended_bbox"]),
    )
    # ---- (f) Sim3DR rasteriser + PNCC composition + refined_head_bbox -------------------------
    from oracle import build_ref
    from oracle import raster_oracle as ro

    ref = build_ref.load()
    assert ref is not None, "oracle/_ref could not be built"

    def sim3dr_rasterize(vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False):
        if bg is not None:
            height, width, channel = bg.shape
        else:
            bg = np.zeros((height, width, channel), dtype=np.uint8)
        buffer = np.zeros((height, width), dtype=np.float32) - 1e8
        if colors.dtype != np.float32:
            colors = colors.astype(np.float32)
        assert bg.flags.c_contiguous and vertices.flags.c_contiguous and vertices.dtype == np.float32 and triangles.dtype == np.int32
        ref.ref_rasterize(bg.ctypes.data, vertices.ctypes.data, triangles.ctypes.data, colors.ctypes.data, buffer.ctypes.data, triangles.shape[0], height, width, channel, 1.0,
                          int(reverse))
        return bg

    sim = types.ModuleType("head_detector.Sim3DR")
    sim.rasterize = sim3dr_rasterize
    sys.modules["head_detector.Sim3DR"] = sim
    sys.modules["head_detector"].Sim3DR = sim
    pp = _load("pncc_processor")
    # single-mesh cases straight through the reference C++
    cases = {}
    for i, (seed, rev) in enumerate(((0, False), (5, True), (9, False))):
        ver, tri, col = ro.random_mesh(seed, n_side=10 + seed, size=60 + 10 * seed, centre=(64 + 3 * seed, 50), depth_scale=30)
        bg = np.random.default_rng(seed).integers(0, 256, (128, 160, 3), dtype=np.uint8)
        cases.update({f"m{i}_ver": ver, f"m{i}_tri": tri, f"m{i}_col": col, f"m{i}_bg": bg, f"m{i}_rev": np.array(rev),
                      f"m{i}_out": sim3dr_rasterize(ver, tri, col, bg=bg.copy(), reverse=rev)})
    # PNCCProcessor on synthetic assets: a 14x14 grid "template", subset = all vertices but a border strip
    rng = np.random.default_rng(21)
    tver, ttri, _ = ro.random_mesh(21, n_side=14, size=1.0, centre=(0.0, 0.0), depth_scale=0.3)
    subset = np.array([k for k in range(tver.shape[0]) if (k % 14) not in (0, 13)], dtype=np.int64)
    fake = {"full_faces.npy": ttri.astype(np.int64), "v_template.npy": tver.astype(np.float64), "head_w_ears.npy": subset}
    real_load = np.load
    pp.np.load = lambda path, *a, **k: fake[os.path.basename(str(path))]
    try:
        proc = pp.PNCCProcessor()
    finally:
        pp.np.load = real_load
    heads_v = []
    for hseed, (cx, cy, sc) in enumerate(((60.0, 50.0, 70.0), (95.0, 60.0, 55.0), (70.0, 75.0, 40.0))):
        v = tver.copy()
        v[:, 0] = cx + sc * v[:, 0]
        v[:, 1] = cy + sc * v[:, 1]
        v[:, 2] = sc * v[:, 2] + rng.normal(0, 0.5, v.shape[0])
        heads_v.append(v.astype(np.float32))
    image = rng.integers(0, 256, (120, 150, 3), dtype=np.uint8)
    heads = [types.SimpleNamespace(vertices_3d=v.copy()) for v in heads_v]
    pncc_img = proc(image, heads)
    hidx = np.array(sorted(rng.choice(tver.shape[0], 40, replace=False).tolist()))
    utils.HEAD_INDICES = hidx
    bb = [utils.refined_head_bbox(v) for v in heads_v]
    np.savez_compressed(
        os.path.join(OUT, "raster_ref.npz"), full_faces=fake["full_faces.npy"], v_template=fake["v_template.npy"], head_w_ears=subset,
        pncc_triangles=proc.triangles, pncc_colors=proc.colors, heads=np.stack(heads_v), heads_after=np.stack([h.vertices_3d for h in heads]),
        image_shape=np.array(image.shape), pncc=pncc_img, head_indices=hidx, bboxes=np.array([[b.x, b.y, b.w, b.h] for b in bb]), **cases,
    )
    # ---- (g) PredictionResult.save_meshes / MeshSaver (detection_result.py:22-35,73-78): the OBJ text the reference writes ------
    _load("draw_utils")  # cv2 is only touched inside the draw functions
    det = _load("detection_result")
    det.np.load = lambda path, *a, **k: fake[os.path.basename(str(path))]
    try:
        pr = det.PredictionResult(image, [types.SimpleNamespace(vertices_3d=v.copy()) for v in heads_v[:2]])
    finally:
        det.np.load = real_load
    with tempfile.TemporaryDirectory() as d:
        pr.save_meshes(os.path.join(d, "meshes"))
        names = sorted(os.listdir(os.path.join(d, "meshes")))
        texts = [open(os.path.join(d, "meshes", n), "rb").read() for n in names]
    np.savez_compressed(os.path.join(OUT, "mesh_obj.npz"), faces=fake["full_faces.npy"], heads=np.stack(heads_v[:2]), names=np.array(names),
                        obj0=np.frombuffer(texts[0], dtype=np.uint8), obj1=np.frombuffer(texts[1], dtype=np.uint8))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
