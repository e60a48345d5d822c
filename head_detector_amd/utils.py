"""Host-side helpers with the reference's names (head_detector/utils.py): nms, rot_mat_from_6dof,
calculate_rpy, limit_angle.  ``nms`` runs on the GPU through libvgh (vgh_topk + vgh_nms + vgh_compact);
there is no torchvision / CPU path behind it."""
from __future__ import annotations

from typing import Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .head_info import RPY

IMAGE_SIZE = 640


def rot_mat_from_6dof(v: torch.Tensor) -> torch.Tensor:
    """6D rotation -> [N,3,3], columns (b1, b2, b3) (head_detector/utils.py:120-128). Plain torch; used by
    host-side consumers (calculate_rpy). The decode kernel has its own copy of this arithmetic."""
    assert v.shape[-1] == 6
    v = v.reshape(-1, 6)
    vx, vy = v[..., :3].clone(), v[..., 3:].clone()
    b1 = F.normalize(vx, dim=-1)
    b3 = F.normalize(torch.cross(b1, vy, dim=-1), dim=-1)
    b2 = -torch.cross(b1, b3, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def limit_angle(angle: Union[int, float], pi: Union[int, float] = 180.0) -> Union[int, float]:
    """Wrap an angle in degrees to [-180, 180] (head_detector/utils.py:131-143)."""
    if angle < -pi:
        angle = angle + (-2 * (int(angle / pi) // 2)) * pi
    if angle > pi:
        angle = angle - (2 * ((int(angle / pi) + 1) // 2)) * pi
    return angle


def rotation_mat_from_flame_params(flame_params) -> np.ndarray:
    return rot_mat_from_6dof(flame_params.rotation.detach().float().cpu()).numpy()[0]


def calculate_rpy(flame_params) -> RPY:
    """head_detector/utils.py:146-151: euler 'xyz' (degrees) of R^T; roll = a[2], pitch = a[0]-180, yaw = a[1]."""
    from scipy.spatial.transform import Rotation

    angle = Rotation.from_matrix(np.transpose(rotation_mat_from_flame_params(flame_params))).as_euler("xyz", degrees=True)
    roll, pitch, yaw = (limit_angle(a) for a in (angle[2], angle[0] - 180, angle[1]))
    return RPY(roll=roll, pitch=pitch, yaw=yaw)


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def nms_batched(boxes_xyxy: torch.Tensor, scores: torch.Tensor, flame_params: torch.Tensor, confidence_threshold: float = 0.5, iou_threshold: float = 0.5,
                top_k: int = 1000, keep_top_k: int = 100) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Every image (the semantics of yolo_heads_post_prediction_callback.py:55-84): fixed-capacity slabs
    boxes [B,keep,4], scores [B,keep], params [B,keep,413] and counts [B] (int32), all on the GPU."""
    lib = _lib.load()
    if not boxes_xyxy.is_cuda:
        raise _lib.VghError("nms: tensors must live on the GPU (no CPU path in this package)")
    B, n, _ = boxes_xyxy.shape
    dev = boxes_xyxy.device
    boxes = boxes_xyxy.detach().float().contiguous()
    sc = scores.detach().float().reshape(B, n).contiguous()
    fl = flame_params.detach().float().contiguous()
    if fl.shape[2] != _lib.NUM_FLAME_PARAMS:
        raise ValueError(f"Invalid number of parameters. Expected: {_lib.NUM_FLAME_PARAMS}. Got: {fl.shape[2]}.")
    if min(top_k, n) > 1024:
        raise _lib.VghError("nms: top_k > 1024 is not supported by the HIP kernel")
    # one library call (vgh_topk_nms): stable top-k, conf filter, greedy NMS, keep-k, rows gathered from the original tensors
    ws = torch.empty(int(lib.vgh_topk_nms_workspace_bytes(B, n, top_k, keep_top_k)), dtype=torch.uint8, device=dev)
    ob = torch.empty(B, keep_top_k, 4, dtype=torch.float32, device=dev)
    os_ = torch.empty(B, keep_top_k, dtype=torch.float32, device=dev)
    of = torch.empty(B, keep_top_k, fl.shape[2], dtype=torch.float32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(lib.vgh_topk_nms(_lib.ptr(boxes), _lib.ptr(sc), _lib.ptr(fl), fl.shape[2], B, n, float(confidence_threshold), float(iou_threshold), int(top_k), int(keep_top_k),
                                _lib.ptr(ws), _lib.ptr(ob), _lib.ptr(os_), _lib.ptr(of), _lib.ptr(counts), _stream_ptr()))
    return ob, os_, of, counts


def nms(boxes_xyxy, scores, flame_params, confidence_threshold: float = 0.5, iou_threshold: float = 0.5, top_k: int = 1000, keep_top_k: int = 100):
    """Drop-in for head_detector/utils.py:159-194, including its quirk: the reference returns from inside
    the batch loop, i.e. the result is for IMAGE 0 ONLY -> (boxes [n,4], scores [n], params [n,413])."""
    ob, os_, of, counts = nms_batched(boxes_xyxy[:1], scores[:1], flame_params[:1], confidence_threshold, iou_threshold, top_k, keep_top_k)
    n = int(counts[0].item())
    return ob[0, :n], os_[0, :n], of[0, :n]
