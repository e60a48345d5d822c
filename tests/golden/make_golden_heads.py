#!/usr/bin/env python3
"""Generate tests/golden/head_decode.npz by RUNNING THE REFERENCE's own head / decoding / post-prediction code.

    python tests/golden/make_golden_heads.py          (build container only: needs /root/reference)

Real reference code that runs here, imported unmodified from /root/reference/yolo_head_training/yolo_head/:
  * yolo_head_dfl_head.py   YoloHeadsDFLHead.__init__ / forward   (rows a5: stems, towers, six FLAME branches, tanh*3, exp/0.05,
                            zero-padding to 300 / 100, concat order)
  * yolo_head_ndfl_heads.py YoloHeadsNDFLHeads.forward / _generate_anchors  (row a6: DFL softmax-expectation, anchors, strides,
                            FLAME fix-up, the from_3dmm -> to_3dmm_tensor permutation)
  * flame.py                FLAME_CONSTS, FlameParams, FLAMELayer, reproject_spatial_vertices (yolo_head twin of head_detector/flame.py)
  * yolo_heads.py           VGGHeadDecodingModule.forward          (row a7: top-k(1000) + flat gather)
  * yolo_heads_post_prediction_callback.py  YoloHeadsPostPredictionCallback.__call__  (the batched twin of utils.nms)
What is absent from this image and therefore substituted (sys.modules), exactly as tests/golden/make_golden.py does for smplx:
  * super_gradients (>=3.7): ConvBNReLU / QARepVGGBlock / width_multiplier := oracle/net_oracle.py's restatements,
    batch_distance2bbox := the two-line published formula, BaseDetectionModule / factories / registries := minimal shells; every
    other name of the package (export helpers, interfaces, ...) is auto-fabricated as an inert class so the modules import.
  * omegaconf, onnx, onnxsim: inert shells.   torchvision.ops.boxes.nms := oracle.postproc_oracle.nms_torchvision.
  * smplx.lbs.lbs := oracle.flame_oracle.lbs; the licensed generic_model.pkl := the seeded synthetic FLAME model.
Weights are NOT stored: both sides regenerate them from head_detector_amd.arch.random_state_dict(variant, seed) (numpy Generator,
platform independent).  The fixture holds inputs + the reference's outputs only; no reference source text.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from head_detector_amd import arch  # noqa: E402
from oracle import flame_oracle as fo  # noqa: E402
from oracle import net_oracle  # noqa: E402
from oracle import postproc_oracle as po  # noqa: E402


# ------------------------------------------------------------------------------------------------------
# inert shells for the absent third-party packages
# ------------------------------------------------------------------------------------------------------
class _InertMeta(type):
    def __getattr__(cls, name):  # class-level constants of enums etc. (e.g. DetectionOutputFormatMode.BATCH_FORMAT in a default argument)
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


class _Inert(metaclass=_InertMeta):
    """Usable as a base class, an instance, a decorator factory and a decorator."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and callable(a[0]):
            return a[0]  # used as a decorator: hand the class / function back unchanged
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


class _ShellModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = _InertMeta(name, (_Inert,), {})  # a distinct class per name (several may appear in one bases list)
        setattr(self, name, cls)
        return cls


class _ShellFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    PREFIXES = ("super_gradients", "omegaconf", "onnx", "onnxsim")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _ShellModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class BaseDetectionModule(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels


def batch_distance2bbox(points, distance):
    """super_gradients.training.utils.bbox_utils.batch_distance2bbox (published): x1y1 = p - lt, x2y2 = p + rb."""
    lt, rb = torch.split(distance, 2, -1)
    return torch.cat([points - lt, points + rb], -1)


class _Factory:
    """DetectionModulesFactory: {"TypeName": {params}} -> instance of the registered class (only the reference's own head here)."""

    registry = {}

    def insert_module_param(self, conf, name, value):
        (k, v), = conf.items()
        return {k: dict(v, **{name: value})}

    def get(self, conf):
        (k, v), = conf.items()
        return self.registry[k](**v)


def _install_shells():
    sys.meta_path.insert(0, _ShellFinder())
    import super_gradients.common.factories.detection_modules_factory as dmf
    import super_gradients.modules as sgm
    import super_gradients.modules.base_modules as sgb
    import super_gradients.modules.utils as sgu
    import super_gradients.training.utils as sgtu
    import super_gradients.training.utils.bbox_utils as sgbb
    import omegaconf
    import super_gradients.training.models.detection_models.pp_yolo_e.pp_yolo_head as ppy

    # only feeds YoloHeadsRawOutputs (loss-side anchors), which nothing on the inference path reads
    ppy.generate_anchors_for_grid_cell = lambda feats, strides, scale, offset: (None, None, None, None)
    sgm.ConvBNReLU = lambda cin, cout, kernel_size, stride, padding, groups=1, bias=False: net_oracle.ConvBNReLU(cin, cout, kernel_size, stride, padding)
    sgm.QARepVGGBlock = net_oracle.QARepVGGBlock
    sgb.BaseDetectionModule = BaseDetectionModule
    sgu.width_multiplier = net_oracle.width_multiplier
    sgbb.batch_distance2bbox = batch_distance2bbox
    sgtu.torch_version_is_greater_or_equal = lambda major, minor: True
    sgtu.HpmStruct = dict
    omegaconf.DictConfig = dict
    dmf.DetectionModulesFactory = _Factory
    # torchvision / smplx exactly as make_golden.py
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
    # package shell: `from yolo_head.x import y` / `from .x import y` resolve to the reference files without running __init__.py
    pkg = types.ModuleType("yolo_head")
    pkg.__path__ = [os.path.join(REF, "yolo_head_training", "yolo_head")]
    sys.modules["yolo_head"] = pkg


def _head_conf(variant: str, level: int) -> dict:
    """The YoloHeadsDFLHead entries of configs/arch_params/yolo_heads_{m,l}_arch_params.yaml (read here from the product's table,
    which test_host_logic checks against those yaml files)."""
    h = arch.VARIANTS[variant]["head"]
    return {"YoloHeadsDFLHead": dict(
        bbox_inter_channels=h["bbox"][level], flame_inter_channels=h["flame"], flame_regression_blocks=h["blocks"], flame_shape_inter_channels=h["shape_inter"],
        flame_expression_inter_channels=h["expr_inter"], flame_shape_out_channels=h["shape_out"], flame_expression_out_channels=h["expr_out"],
        flame_transformation_inter_channels=h["tr_inter"], shared_stem=False, width_mult=h["width_mult"], first_conv_group_size=0, stride=arch.STRIDES[level], reg_max=16)}


def main():
    _install_shells()
    import yolo_head.flame as rflame
    import yolo_head.yolo_head_dfl_head as dfl
    import yolo_head.yolo_head_ndfl_heads as ndfl
    import yolo_head.yolo_heads as yh
    import yolo_head.yolo_heads_post_prediction_callback as cb

    _Factory.registry["YoloHeadsDFLHead"] = dfl.YoloHeadsDFLHead
    # the licensed pickle is absent: FLAMELayer(FLAME_CONSTS) inside the callback reads the seeded synthetic model instead
    model = fo.synthetic_flame_model(seed=3)
    tmp = tempfile.NamedTemporaryFile(suffix=".pkl", delete=False)
    pickle.dump({**model, "f": np.zeros((4, 3), dtype=np.int64), "kintree_table": np.array([[4294967295, 0, 1, 1, 1], [0, 1, 2, 3, 4]], dtype=np.int64),
                 "weights": model["weights"]}, tmp)
    tmp.close()
    orig_get = rflame.get_flame_model
    rflame.get_flame_model = lambda flame_path=None: orig_get(tmp.name)

    out = {}
    for variant, seed in (("vgg_heads_l", 31), ("vgg_heads_m", 32)):
        tag = variant[-1]
        v = arch.VARIANTS[variant]
        in_ch = (v["neck"][1][0], v["neck"][2][0], v["neck"][3][0])
        heads = ndfl.YoloHeadsNDFLHeads(num_classes=1, in_channels=in_ch, heads_list=[_head_conf(variant, i) for i in range(3)], reg_max=16)
        sd = {k[len("heads."):]: torch.from_numpy(a) for k, a in arch.random_state_dict(variant, seed).items() if k.startswith("heads.")}
        missing, unexpected = heads.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing[:5], unexpected[:5])
        heads.eval()
        B, sizes = 2, ((8, 8), (4, 4), (2, 2))
        g = torch.Generator().manual_seed(seed)
        feats = [torch.randn(B, c, h, w, generator=g).relu() for c, (h, w) in zip(in_ch, sizes)]
        raw = {}
        hooks = []
        for lv in range(3):
            hd = getattr(heads, f"head{lv + 1}")
            for br in ("shape", "expression", "rotation", "jaw", "translation", "scale"):
                hooks.append(getattr(hd, f"flame_{br}_pred").register_forward_hook(lambda m, i, o, key=f"raw{lv}_{br}": raw.__setitem__(key, o.detach().clone())))
        with torch.no_grad():
            per_level = [getattr(heads, f"head{lv + 1}")(feats[lv]) for lv in range(3)]
            decoded, _ = heads(tuple(feats))
            k = 60
            cand = yh.VGGHeadDecodingModule(num_pre_nms_predictions=k)((decoded, None))
            post = cb.YoloHeadsPostPredictionCallback(confidence_threshold=float(decoded.scores.flatten().sort().values[-40]), nms_iou_threshold=0.5,
                                                      pre_nms_max_predictions=30, post_nms_max_predictions=10)((decoded, None))
        for h_ in hooks:
            h_.remove()
        out[f"{tag}_seed"] = np.array(seed)
        for lv in range(3):
            out[f"{tag}_feat{lv}"] = feats[lv].numpy()
            out[f"{tag}_reg{lv}"], out[f"{tag}_cls{lv}"], out[f"{tag}_flame{lv}"] = (t.numpy() for t in per_level[lv])
        out.update({f"{tag}_{k_}": t.numpy() for k_, t in raw.items()})
        out[f"{tag}_boxes"], out[f"{tag}_scores"], out[f"{tag}_flame"] = decoded.boxes_xyxy.numpy(), decoded.scores.numpy(), decoded.flame_params.numpy()
        out[f"{tag}_cand_boxes"], out[f"{tag}_cand_scores"], out[f"{tag}_cand_flame"] = (t.numpy() for t in cand)
        out[f"{tag}_post_conf"] = np.array(float(decoded.scores.flatten().sort().values[-40]), dtype=np.float32)
        out[f"{tag}_post_counts"] = np.array([len(p.scores) for p in post])
        for b, p in enumerate(post):
            out[f"{tag}_post{b}_boxes"], out[f"{tag}_post{b}_scores"], out[f"{tag}_post{b}_params"] = p.bboxes_xyxy.numpy(), p.scores.numpy(), p.mm_params.numpy()
            out[f"{tag}_post{b}_v3d"] = np.asarray(p.predicted_3d_vertices)[:, ::97]  # every 97th vertex keeps the file small
        print(variant, "anchors", decoded.boxes_xyxy.shape[1], "post counts", out[f"{tag}_post_counts"])
    np.savez_compressed(os.path.join(OUT, "head_decode.npz"), **{k: (a.astype(np.float32) if a.dtype == np.float64 else a) for k, a in out.items()})
    os.unlink(tmp.name)
    print("wrote", os.path.join(OUT, "head_decode.npz"), os.path.getsize(os.path.join(OUT, "head_decode.npz")), "bytes")


if __name__ == "__main__":
    main()
