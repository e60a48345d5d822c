"""HeadDetector: the reference's facade (head_detector/detector.py:18-102) on the MI355X engine.

    detector = HeadDetector()                       # model="vgg_heads_l", image_size=640
    predictions = detector(image, confidence_threshold=0.5)
    predictions.heads -> List[HeadMetadata(bbox, score, flame_params, vertices_3d, head_pose)]

Differences that are forced, not chosen:
  * weights: the reference downloads ``okupyn/vgg_heads/<model>.trcd`` from the HF hub (detector.py:25-30).
    There is no network here, so ``weights`` is a path to a user-supplied .trcd / state_dict (default: ``<model>.trcd`` next
    to this package; a missing file raises, naming it), or the explicit opt-in ``"synthetic"`` for seeded random weights of the
    same architecture (throughput / plumbing only);
  * FLAME constants: ``flame_path`` / ``flame_model`` as in FLAMELayer (user-supplied licensed asset);
  * letterbox resize: cv2.INTER_LANCZOS4 + constant border (detector.py:47-50) restated as a HIP kernel (letterbox.py); cv2 is
    not needed at run time (and is absent here: that stage is "parity unpinned").
Preserved quirks: nms() looks at image 0 only; FlameParams.translation is left in padded-640 space while
scale is divided by the letterbox scale (detector.py:78-79); bbox via np.rint -> int; z is divided by scale too.
"""
from __future__ import annotations

import os
import re
import warnings
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .detection_result import PredictionResult
from .engine import VGHeadsEngine
from .flame import FLAMELayer
from .head_info import RPY, Bbox, FlameParams, HeadMetadata
from .utils import calculate_rpy

REPO_ID = "okupyn/vgg_heads"


LAST_LOAD_REPORT: Dict[str, Any] = {}  # diagnostics of the most recent load_weights call (how an .onnx was read, which initializers were not weights)


def _known_keys() -> set:
    """Every parameter name any supported archive form may carry, over all variants: the unfused state dict, `rbr_reparam.*` of fused RepVGG blocks, `conv.bias` of
    Conv + BatchNorm blocks an exporter merged."""
    from . import arch

    keys = set()
    for v in arch.VARIANTS:
        keys |= set(arch.random_state_dict(v, 0))
        for sp in arch.layer_specs(v):
            if sp.kind == "qarep":
                keys |= {f"{sp.name}.rbr_reparam.weight", f"{sp.name}.rbr_reparam.bias"}
            elif sp.kind in ("conv", "cbr"):
                keys.add((sp.name if sp.kind == "conv" else f"{sp.name}.seq") + ".conv.bias")
    return keys


def load_weights(path: str, variant: Optional[str] = None) -> Dict[str, np.ndarray]:
    """state_dict of a released TorchScript archive (.trcd), of a torch checkpoint, or of an ONNX export (.onnx) -> {name: ndarray} with the
    ``model.`` prefix of ConvertableCompletePipelineModel stripped (exportable_mesh_model.py:421-427).  A training checkpoint's
    EMA weights win over the raw ones (they are what the export pipeline serialises).

    ``.onnx`` (README.md:23,199): read by a protobuf wire reader of our own (onnx_wire.py; the `onnx` package is not a dependency), two ways.  A file whose initializers
    still carry parameter names is read BY NAME (initializers that are not parameters -- shape constants, anchors, ``onnx::`` scalars -- are set aside and listed in
    ``LAST_LOAD_REPORT["not_weights"]``, not reported as unexpected keys).  A file that went through the exporter's ``onnxsim.simplify`` (:483-488) has anonymous,
    BatchNorm-merged ``onnx::Conv_NNN`` tensors: it is read BY GRAPH POSITION (onnx_graph.py: Conv nodes bound to the architecture by topology + shape; ``variant``
    narrows the search, default: every known architecture) into the fused naming arch.fold_state_dict accepts.  Both are UNPINNED against a real export (none in this image)."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.lower().endswith(".onnx"):
        from . import onnx_graph, onnx_wire

        graph = onnx_wire.load_graph(path)
        known = _known_keys()
        named, extras = {}, []
        for k, v in graph["tensors"].items():
            kk = k[len("model."):] if k.startswith("model.") else k
            if kk in known and v.dtype.kind in "fiu" and v.dtype != np.bool_:
                named[kk] = np.ascontiguousarray(v, dtype=np.float32)
            else:
                extras.append(k)
        n4 = sum(1 for v in graph["tensors"].values() if getattr(v, "ndim", 0) == 4)
        LAST_LOAD_REPORT.clear()
        if named and sum(1 for v in named.values() if v.ndim == 4) * 2 >= n4:  # the parameter names survived the export
            LAST_LOAD_REPORT.update(path=path, how="by name", not_weights=extras)
            return named
        v_, sd, rep = onnx_graph.load_by_graph(path, variant)
        LAST_LOAD_REPORT.update(path=path, how="by graph position", variant=v_, **rep)
        return sd
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except (RuntimeError, ValueError) as jit_err:  # not a TorchScript archive: a plain checkpoint?
        try:
            obj = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as ckpt_err:
            raise ValueError(f"{path}: neither a TorchScript archive ({str(jit_err).splitlines()[0]}) nor a torch checkpoint ({ckpt_err})") from ckpt_err
        if isinstance(obj, dict):
            sd = obj.get("ema_net", obj.get("net", obj))
        else:
            sd = obj.state_dict()
    out = {}
    for k, v in sd.items():
        k = k[len("model."):] if k.startswith("model.") else k
        out[k] = v.detach().float().numpy()
    return out


def weight_manifest_diff(variant: str, sd: Dict[str, np.ndarray]) -> Dict[str, list]:
    """Keys / shapes of ``sd`` against the architecture reconstructed from the arch yaml: the first contact with a
    real released blob should fail HERE, with the whole difference, not with a KeyError somewhere inside the fold."""
    from . import arch

    want = {k: v.shape for k, v in arch.random_state_dict(variant, 0).items()}
    # u4: a RepVGG block exported AFTER fusion carries `<block>.rbr_reparam.{weight,bias}` (+ post_bn.* when only partially fused) instead
    # of its branch tensors -- arch.fold_state_dict reads either form, so the branch keys of such a block are not "missing"
    for spec in arch.layer_specs(variant):
        if spec.kind == "qarep" and f"{spec.name}.rbr_reparam.weight" in sd and f"{spec.name}.branch_3x3.conv.weight" not in sd:
            for k in [k for k in want if k.startswith(spec.name + ".")]:
                if not (k.startswith(spec.name + ".post_bn.") and k in sd):
                    del want[k]
            want[f"{spec.name}.rbr_reparam.weight"] = (spec.cout, spec.cin, 3, 3)
            want[f"{spec.name}.rbr_reparam.bias"] = (spec.cout,)
    # u8: a Conv + BatchNorm block whose BN the exporter merged (torch.onnx.export folds eval-mode BN into the conv when it keeps the conv's name): `<block>.conv.weight`
    # + `<block>.conv.bias` and no BN tensors -- arch.fold_state_dict takes that form as already folded
    for spec in arch.layer_specs(variant):
        if spec.kind in ("conv", "cbr"):
            pfx = spec.name if spec.kind == "conv" else f"{spec.name}.seq"
            if f"{pfx}.conv.bias" in sd and f"{pfx}.bn.running_var" not in sd:
                for k in [k for k in want if k.startswith(pfx + ".bn.")]:
                    del want[k]
                want[f"{pfx}.conv.bias"] = (spec.cout,)
    ignore = re.compile(r"(num_batches_tracked|anchor_points|stride_tensor|proj_conv|max_batch)")
    have = {k: tuple(np.asarray(v).shape) for k, v in sd.items() if not ignore.search(k)}
    # an UNFUSED block may still carry an unused rbr_reparam conv (SG keeps the attribute around): not an error
    unexpected = sorted(k for k in have if k not in want and not (".rbr_reparam." in k and k.rsplit(".rbr_reparam.", 1)[0] + ".branch_3x3.conv.weight" in have))
    return {"missing": sorted(k for k in want if k not in have), "unexpected": unexpected,
            "shape": sorted(f"{k}: expected {tuple(want[k])}, got {have[k]}" for k in want if k in have and tuple(want[k]) != have[k])}


class HeadDetector:
    def __init__(self, model: str = "vgg_heads_l", image_size: int = 640, *, weights: Optional[str] = None, flame_path: Optional[str] = None,
                 flame_model: Optional[Dict[str, Any]] = None, seed: int = 1, max_batch: int = 1,
                 assets_dir: Optional[str] = None, mesh_assets=None, precision: str = "bf16", calibration_images: Optional[Sequence] = None):
        """``precision``: "bf16" (throughput mode, the default); "fp16" (r05: one fp16 plane per value -- the reference exporter's own FP16 format -- 8 x closer to the fp32 network at
        0.92 x the speed); "fp8" (r05: bf16 with OCP-e4m3 links between 3x3 convs, 1.12 x the speed at 5 - 10 x the bf16 deviation; its activation scales are calibrated on
        ``calibration_images`` -- a list of images like the ones this detector will see (paths, PIL or HWC uint8 arrays), letterboxed here -- and WITHOUT them on two seeded random images, which is
        adequate for plumbing only: a warning says so); "int8" (r05: the same links as int8 codes -- the reference exporter's QuantizationMode.INT8 -- with the folded identity branch of the RepVGG
        convs kept in fp32; 1.07 x the speed at 2 - 3 x the bf16 deviation; calibrated like "fp8"); "fp16x3" -- the matrix-core parity mode whose outputs match the reference's
        fp32 CPU network to IoU >= 0.999 / 1e-4 (csrc/conv_split.hip; ~1/3 of the bf16 throughput); "fp32" = the VALU parity mode."""
        if not torch.cuda.is_available():
            raise _lib.VghError("HeadDetector: no GPU visible. This package is the MI355X HIP path only; it does not fall back to the CPU.")
        self._image_size = image_size
        self._device = torch.device("cuda", torch.cuda.current_device())
        self._max_batch = max_batch
        self._precision = precision
        self._calibration_images = calibration_images
        self._flame = FLAMELayer(flame_path=flame_path, model=flame_model, device=self._device, max_heads=max(1024, 100 * max_batch))
        self.model = self._read_model(model, weights, seed)
        # mesh assets of the reference (head_detector/assets) for PredictionResult.get_pncc(); user-supplied, optional
        self._pncc = None
        if mesh_assets is not None or assets_dir is not None:
            from .pncc import PNCCProcessor

            self._pncc = PNCCProcessor(mesh_assets if mesh_assets is not None else assets_dir)

    def _read_model(self, model: str, weights: Optional[str], seed: int) -> VGHeadsEngine:
        """detector.py:25-30 downloads ``okupyn/vgg_heads/<model>.trcd``; without a network the archive is a user-supplied file:
        ``weights`` = its path (default: ``<model>.trcd`` next to this package).  ``weights="synthetic"`` is the explicit opt-in for
        seeded random weights of the same architecture (plumbing / throughput work) -- never a silent default."""
        if weights is None:
            cand = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{model}.trcd")
            if not os.path.exists(cand):
                raise FileNotFoundError(
                    f"HeadDetector: no weights for {model!r}. The reference fetches {REPO_ID}/{model}.trcd from the Hugging Face hub (head_detector/detector.py:25-30); "
                    f"there is no network here: pass weights=<path to {model}.trcd> (or place it at {cand}), or weights='synthetic' for seeded random weights "
                    "of the same architecture (throughput / plumbing only: the detections are meaningless).")
            weights = cand
        if weights == "synthetic":
            warnings.warn(f"HeadDetector({model!r}): seeded SYNTHETIC weights (seed {seed}) -- detections are meaningless; supply the released {model}.trcd for real use", stacklevel=3)
            sd = None
        else:
            sd = load_weights(weights, model)
            diff = weight_manifest_diff(model, sd)
            if any(diff.values()):
                raise ValueError(f"{weights} does not match the {model} architecture this engine lowers:\n" + "\n".join(f"  {k}: {v[:12]}{' ...' if len(v) > 12 else ''}" for k, v in diff.items() if v))
        calib = None
        if self._precision in ("fp8", "int8"):
            if self._calibration_images is not None and len(self._calibration_images) > 0:
                # the letterboxed u8 canvases the network will see (detector.py:40-52), as one batch
                calib = torch.cat([self._transform_image(self._convert_image(im))[0] for im in self._calibration_images]).contiguous()
            else:
                warnings.warn(f"HeadDetector(precision={self._precision!r}) without calibration_images: the 8-bit links are scaled for two seeded RANDOM images -- pass images like the ones you will detect on",
                              stacklevel=3)
        return VGHeadsEngine(model, state_dict=sd, image_size=self._image_size, max_batch=self._max_batch, seed=seed, precision=self._precision, calib_images=calib)

    # ---- host-side image handling (detector.py:32-56) ----------------------------------------------------
    def _convert_image(self, image) -> np.ndarray:
        if isinstance(image, str):
            try:
                import cv2

                image = cv2.cvtColor(cv2.imread(image), cv2.COLOR_BGR2RGB)
            except ImportError:
                from PIL import Image

                image = np.array(Image.open(image).convert("RGB"))
        elif not isinstance(image, np.ndarray):
            image = np.array(image)  # PIL.Image
        return image

    def _transform_image(self, image: np.ndarray) -> Tuple[torch.Tensor, Tuple[int, int], float]:
        """detector.py:40-52 on the GPU (head_detector_amd.letterbox): LANCZOS4 resize in OpenCV's 8-bit fixed-point arithmetic,
        the (127, 0, 0) constant border of ``copyMakeBorder(..., value=127)``; the u8 -> float /255 conversion (detector.py:51)
        is fused into the stem kernel, so the network input is the u8 NHWC canvas."""
        from .letterbox import letterbox

        canvas, padding, scale = letterbox(image, self._image_size, self._device)
        return canvas.unsqueeze(0), padding, scale

    def _preprocess(self, image: np.ndarray):
        image, padding, scale = self._transform_image(image)
        return image, {"padding": padding, "scale": scale}

    def _process(self, image: torch.Tensor):
        return self.model.model(image)

    # ---- post-processing (detector.py:61-95) -------------------------------------------------------------------
    def _parse_predictions(self, bboxes_xyxy: torch.Tensor, scores: torch.Tensor, flame_params: torch.Tensor, cache: Dict[str, Any]) -> List[HeadMetadata]:
        padding, scale = cache["padding"], cache["scale"]
        n = flame_params.shape[0]
        P = self.model.program
        unpad = torch.tensor([[padding[0], padding[1], scale]], dtype=torch.float32, device=self._device).expand(n, 3).contiguous()
        # decode + (x - pad_x, y - pad_y, z) / scale fused in the kernel epilogue (detector.py:66-69)
        if n:
            _, _, final_3d_pts = self._flame.decode(flame_params, unpad=unpad, shape_live=P.shape_c, expr_live=P.expr_c, want_vertices=False)
            final_3d_pts = final_3d_pts.cpu().numpy()
        else:
            final_3d_pts = np.zeros((0, self._flame.v_template.shape[0], 3), dtype=np.float32)
        bboxes_xyxy = bboxes_xyxy.cpu().numpy().clip(0, self._image_size)
        scores = scores.cpu().numpy()
        bboxes_xyxy[:, [0, 2]] -= padding[0]
        bboxes_xyxy[:, [1, 3]] -= padding[1]
        bboxes_xyxy /= scale
        bboxes_xyxy = np.rint(bboxes_xyxy).astype(int)
        result = []
        flame_params = flame_params.detach().cpu()
        for bbox, score, params, vertices in zip(bboxes_xyxy, scores, flame_params, final_3d_pts):
            params = FlameParams.from_3dmm(params.unsqueeze(0))
            params.scale = params.scale / scale
            box = Bbox(x=bbox[0], y=bbox[1], w=bbox[2] - bbox[0], h=bbox[3] - bbox[1])
            result.append(HeadMetadata(bbox=box, score=score, flame_params=params, vertices_3d=vertices, head_pose=calculate_rpy(params)))
        return result

    def _postprocess(self, predictions, cache: Dict[str, Any], confidence_threshold: float) -> List[HeadMetadata]:
        from .utils import nms

        boxes, scores, flame_params = predictions
        boxes, scores, flame_params = nms(boxes, scores, flame_params, confidence_threshold=confidence_threshold)
        return self._parse_predictions(boxes, scores, flame_params, cache)

    def detect_batch(self, images: Sequence[Union[str, "np.ndarray", Any]], confidence_threshold: float = 0.5) -> List[PredictionResult]:
        """Batched twin of ``__call__`` (what yolo_heads_post_prediction_callback.py:41-99 does for a batch): every image goes
        through ONE fused device call (vgh_detect: net -> top-k -> NMS per image -> FLAME decode + un-pad + head pose of every
        survivor); only the final per-head Python objects are built on the host.  Needs ``max_batch >= len(images)``."""
        if len(images) > self._max_batch:
            raise ValueError(f"detect_batch: {len(images)} images exceed max_batch={self._max_batch} (pass max_batch= to HeadDetector)")
        originals = [self._convert_image(im) for im in images]
        if not originals:
            return []
        tensors, caches = zip(*[self._preprocess(im) for im in originals])
        batch = torch.cat(tensors, 0).contiguous()
        unpad = torch.tensor([[c["padding"][0], c["padding"][1], c["scale"]] for c in caches], dtype=torch.float32, device=self._device)
        det = self.model.detect(batch, confidence_threshold=confidence_threshold, flame=self._flame, unpad=unpad)
        counts = det.counts.cpu().numpy()
        n = det.num_heads
        verts = det.vertices_3d.cpu().numpy()
        rpy = det.head_pose.cpu().numpy().astype(np.float64)
        boxes, scores, params = det.boxes.cpu().numpy(), det.scores.cpu().numpy(), det.flame_params.cpu()
        results, at = [], 0
        S = self._image_size
        for b, (orig, cache) in enumerate(zip(originals, caches)):
            padding, scale = cache["padding"], cache["scale"]
            heads = []
            for i in range(int(counts[b])):
                if at >= n:
                    break  # head capacity exhausted (never with the default capacities)
                bb = boxes[b, i].clip(0, S)
                bb[[0, 2]] -= padding[0]
                bb[[1, 3]] -= padding[1]
                bb = np.rint(bb / scale).astype(int)
                fp = FlameParams.from_3dmm(params[b, i].unsqueeze(0))
                fp.scale = fp.scale / scale
                heads.append(HeadMetadata(bbox=Bbox(x=bb[0], y=bb[1], w=bb[2] - bb[0], h=bb[3] - bb[1]), score=scores[b, i], flame_params=fp, vertices_3d=verts[at],
                                          head_pose=RPY(roll=float(rpy[at, 0]), pitch=float(rpy[at, 1]), yaw=float(rpy[at, 2]))))
                at += 1
            results.append(PredictionResult(original_image=orig, heads=heads, faces=self._flame.faces, pncc_processor=self._pncc))
        return results

    def __call__(self, image: Union[str, "np.ndarray", Any], confidence_threshold: float = 0.5) -> PredictionResult:
        original_image = self._convert_image(image)
        image, cache = self._preprocess(original_image)
        predictions = self._process(image)
        heads = self._postprocess(predictions, cache, confidence_threshold)
        return PredictionResult(original_image=original_image, heads=heads, faces=self._flame.faces, pncc_processor=self._pncc)
