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
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
