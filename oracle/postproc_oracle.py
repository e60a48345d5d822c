"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the post-network stages of the reference forward path:

  yolo_head_training/yolo_head/yolo_head_ndfl_heads.py:117-175,206-236
        YoloHeadsNDFLHeads.forward / _generate_anchors      -> ndfl_decode, make_anchors
  yolo_head_training/yolo_head/yolo_head_dfl_head.py:162-184
        activations + zero-pad + concat of the six FLAME branches -> assemble_flame_channels
  yolo_head_training/yolo_head/yolo_heads.py:44-86
        VGGHeadDecodingModule.forward (top-k 1000 + gather)  -> decoding_topk
  head_detector/utils.py:159-194     nms() (image 0 only!)   -> nms_reference
  yolo_head_training/yolo_head/yolo_heads_post_prediction_callback.py:41-99
        batched twin                                         -> postprocess_batched
  torchvision ~=0.15.2 ops.boxes.nms (third-party, absent; CPU kernel restated) -> nms_torchvision

Tie policy (the reference leaves it to torch.topk / torch.sort, which are unspecified for equal
keys): equal scores are ordered by ascending original index, in the oracle and in the HIP
kernels alike.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from .flame_oracle import FLAME_CONSTS, join_3dmm, split_3dmm

REG_MAX = 16


def make_anchors(sizes: Sequence[Tuple[int, int]], strides: Sequence[int], offset: float = 0.5, dtype=torch.float32):
    """_generate_anchors (yolo_head_ndfl_heads.py:206-236): (x+0.5, y+0.5) in stride units,
    level-major then row-major (meshgrid indexing='ij', y outer / x inner)."""
    pts, st = [], []
    for (h, w), s in zip(sizes, strides):
        sx = torch.arange(w, dtype=dtype) + offset
        sy = torch.arange(h, dtype=dtype) + offset
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack([xx, yy], dim=-1).reshape(-1, 2))
        st.append(torch.full([h * w, 1], s, dtype=dtype))
    return torch.cat(pts), torch.cat(st)


def assemble_flame_channels(shape, expr, rot, jaw, trans, scale):
    """Tail of YoloHeadsDFLHead.forward (yolo_head_dfl_head.py:162-184). Inputs are the raw
    1x1-conv outputs [B,C,H,W]; returns flame_output [B,413,H,W] in HEAD order
    [shape300, expr100, rot6, jaw3, trans3, scale1]."""
    shape = shape.tanh() * 3
    expr = expr.tanh() * 3
    scale = scale.exp() / 0.05
    shape = torch.nn.functional.pad(shape, (0, 0, 0, 0, 0, FLAME_CONSTS["shape"] - shape.size(1)))
    expr = torch.nn.functional.pad(expr, (0, 0, 0, 0, 0, FLAME_CONSTS["expression"] - expr.size(1)))
    return torch.cat([shape, expr, rot, jaw, trans, scale], dim=1)


def ndfl_decode(levels, strides=(8, 16, 32)):
    """YoloHeadsNDFLHeads.forward in tracing mode (yolo_head_ndfl_heads.py:133-175).

    levels: list of (reg_distri [B,68,H,W], cls_logit [B,1,H,W], flame_output [B,413,H,W]).
    returns boxes [B,A,4] xyxy px, scores [B,A,1], flame [B,A,413] (network OUTPUT layout,
    i.e. after the from_3dmm -> to_3dmm_tensor channel permutation)."""
    cls_l, red_l, fl_l, sizes = [], [], [], []
    for reg, cls, fl in levels:
        b, _, h, w = reg.shape
        hw = h * w
        sizes.append((h, w))
        r = torch.permute(reg.reshape([-1, 4, REG_MAX + 1, hw]), [0, 2, 3, 1])
        proj = torch.linspace(0, REG_MAX, REG_MAX + 1, dtype=reg.dtype).reshape([1, REG_MAX + 1, 1, 1])
        red_l.append(torch.nn.functional.softmax(r, dim=1).mul(proj).sum(1))  # [B,hw,4]
        cls_l.append(cls.reshape([b, -1, hw]))
        fl_l.append(fl.flatten(2))
    cls_all = torch.permute(torch.cat(cls_l, dim=-1), [0, 2, 1])
    red_all = torch.cat(red_l, dim=1)
    anchor_points, stride_tensor = make_anchors(sizes, strides, dtype=red_all.dtype)
    centers = anchor_points * stride_tensor
    scores = cls_all.sigmoid()
    # super_gradients batch_distance2bbox: x1y1 = p - lt ; x2y2 = p + rb
    lt, rb = torch.split(red_all, 2, dim=-1)
    boxes = torch.cat([-lt + anchor_points, rb + anchor_points], dim=-1) * stride_tensor
    fl = torch.cat(fl_l, dim=-1)  # [B,413,A]
    fp = {k: v.clone() for k, v in split_3dmm(fl).items()}
    fp["translation"][:, 0:2] += centers.T[None]
    fp["scale"] *= stride_tensor[None, None, :, 0]
    flame = join_3dmm(fp).permute(0, 2, 1).contiguous()
    return boxes, scores, flame


def stable_topk(scores_1d: torch.Tensor, k: int) -> torch.Tensor:
    """indices of the k largest, descending, ties -> ascending index."""
    order = torch.sort(scores_1d, descending=True, stable=True).indices
    return order[:k]


def decoding_topk(boxes, scores, flame, k: int = 1000):
    """VGGHeadDecodingModule.forward (yolo_heads.py:63-86)."""
    B = scores.shape[0]
    idx = torch.stack([stable_topk(scores[b, :, 0], k) for b in range(B)])
    g = lambda t: torch.stack([t[b, idx[b]] for b in range(B)])  # noqa: E731
    return g(boxes), g(scores), g(flame), idx


def nms_torchvision(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision/csrc/ops/cpu/nms_kernel.cpp semantics, float32 arithmetic throughout:
    areas=(x2-x1)*(y2-y1); visit in descending score; suppress j when
    inter/(area_i+area_j-inter) > thr (strict). Returns kept indices in visit order."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1 :]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), (xx2 - xx1).astype(np.float32))
        h = np.maximum(np.float32(0), (yy2 - yy1).astype(np.float32))
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / ((areas[i] + areas[rest]).astype(np.float32) - inter).astype(np.float32)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def _nms_one(boxes, scores, flame, conf, iou, top_k, keep_top_k):
    s = scores.squeeze(-1)
    mask = s >= conf
    s, b, f = s[mask], boxes[mask], flame[mask]
    if s.size(0) > top_k:
        idx = stable_topk(s, top_k)
        s, b, f = s[idx], b[idx], f[idx]
    keep = torch.from_numpy(nms_torchvision(b.numpy(), s.numpy(), iou))
    return b[keep][:keep_top_k], s[keep][:keep_top_k], f[keep][:keep_top_k]


def nms_reference(boxes, scores, flame, confidence_threshold=0.5, iou_threshold=0.5, top_k=1000, keep_top_k=100):
    """head_detector/utils.py:159-194 -- note the ``return`` inside the loop: IMAGE 0 ONLY."""
    for b, s, f in zip(boxes.detach().float(), scores.detach().float(), flame.detach().float()):
        return _nms_one(b, s, f, confidence_threshold, iou_threshold, top_k, keep_top_k)


def postprocess_batched(boxes, scores, flame, confidence_threshold, iou_threshold, pre_nms_max=1000, post_nms_max=100) -> List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """yolo_heads_post_prediction_callback.py:55-84 -- every image."""
    return [
        _nms_one(b, s, f, confidence_threshold, iou_threshold, pre_nms_max, post_nms_max)
        for b, s, f in zip(boxes.detach().float(), scores.detach().float(), flame.detach().float())
    ]


def synthetic_detections(B: int, A_sizes=((80, 80), (40, 40), (20, 20)), strides=(8, 16, 32), seed: int = 2, mean_heads: float = 3.0, max_heads: int = 100, image_size: int = 640):
    """Synthetic head-output injector of SURVEY.md 8(d) config 3: per image h~Poisson(mean)
    clamped to [1,max_heads] well-separated boxes with scores U(0.6,0.99) plus 5-20 jittered
    duplicates (IoU>0.5) each; everything else low score. Returns dense boxes [B,A,4],
    scores [B,A,1] (no exact ties) for A = sum(h*w)."""
    rng = np.random.default_rng(seed)
    A = sum(h * w for h, w in A_sizes)
    boxes = np.zeros((B, A, 4), dtype=np.float32)
    scores = np.zeros((B, A, 1), dtype=np.float32)
    for b in range(B):
        # background: tiny scores, random small boxes
        cx = rng.uniform(0, image_size, A)
        cy = rng.uniform(0, image_size, A)
        wh = rng.uniform(4, 40, (A, 2))
        boxes[b] = np.stack([cx - wh[:, 0] / 2, cy - wh[:, 1] / 2, cx + wh[:, 0] / 2, cy + wh[:, 1] / 2], 1)
        scores[b, :, 0] = rng.uniform(1e-4, 0.3, A)
        h = int(np.clip(rng.poisson(mean_heads), 1, max_heads))
        grid = int(np.ceil(np.sqrt(h)))
        cell = image_size / grid
        slots = rng.permutation(A)
        cur = 0
        for i in range(h):
            gx, gy = i % grid, i // grid
            size = cell * rng.uniform(0.45, 0.8)
            c = np.array([(gx + 0.5) * cell, (gy + 0.5) * cell])
            base = np.array([c[0] - size / 2, c[1] - size / 2, c[0] + size / 2, c[1] + size / 2])
            nd = int(rng.integers(5, 21))
            for d in range(nd + 1):
                a = slots[cur]
                cur += 1
                jit = rng.uniform(-0.06, 0.06, 4) * size if d else 0.0
                boxes[b, a] = base + jit
                scores[b, a, 0] = rng.uniform(0.6, 0.99) if d == 0 else rng.uniform(0.5, 0.6)
    # break exact ties deterministically
    scores += (np.arange(A, dtype=np.float32)[None, :, None] % 997) * np.float32(1e-7)
    return torch.from_numpy(boxes), torch.from_numpy(scores)
