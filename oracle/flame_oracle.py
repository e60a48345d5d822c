"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the FLAME decode path of the reference:

  head_detector/head_info.py:44-89   FlameParams.from_3dmm   -> split_3dmm
  head_detector/utils.py:120-128     rot_mat_from_6dof       -> rot_mat_from_6dof
  head_detector/flame.py:122-169     FLAMELayer.forward      -> flame_forward
  head_detector/flame.py:179-208     reproject_spatial_vertices -> reproject
  head_detector/detector.py:61-90    _parse_predictions (vertex/bbox un-pad, un-scale)
  head_detector/utils.py:131-156     calculate_rpy / limit_angle -> calculate_rpy

``lbs`` restates smplx==0.1.26 ``smplx.lbs.lbs`` (third-party, not under /root/reference;
call site head_detector/flame.py:152-161): blend_shapes, vertices2joints,
batch_rodrigues, batch_rigid_transform, skinning.

Everything is dtype-generic torch: float64 = arbiter, float32 = "reference CPU path".
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# head_detector/head_info.py:12-21
FLAME_CONSTS = {
    "shape": 300,
    "expression": 100,
    "rotation": 6,
    "jaw": 3,
    "eyeballs": 0,
    "neck": 0,
    "translation": 3,
    "scale": 1,
}
NUM_PARAMS = sum(FLAME_CONSTS.values())  # 413
MESH_OFFSET_Z = 0.05  # head_detector/flame.py:34


# --------------------------------------------------------------------------------------
# smplx.lbs restatement
# --------------------------------------------------------------------------------------
def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """smplx.lbs.batch_rodrigues: [N,3] axis-angle -> [N,3,3].

    angle = ||r + 1e-8||, dir = r / angle, R = I + sin*K + (1-cos)*K@K.
    """
    n = rot_vecs.shape[0]
    dtype, device = rot_vecs.dtype, rot_vecs.device
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=dtype, device=device)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((n, 3, 3))
    ident = torch.eye(3, dtype=dtype, device=device).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def _transform_mat(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """smplx.lbs.transform_mat: [N,3,3],[N,3,1] -> [N,4,4]."""
    return torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rigid_transform(rot_mats, joints, parents):
    """smplx.lbs.batch_rigid_transform.

    rot_mats [B,J,3,3], joints [B,J,3], parents [J] (parents[0] == -1)
    returns posed_joints [B,J,3], rel_transforms A [B,J,4,4].
    """
    joints = torch.unsqueeze(joints, dim=-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] -= joints[:, parents[1:]]
    transforms_mat = _transform_mat(rot_mats.reshape(-1, 3, 3), rel_joints.reshape(-1, 3, 1)).reshape(
        -1, joints.shape[1], 4, 4
    )
    transform_chain = [transforms_mat[:, 0]]
    for i in range(1, parents.shape[0]):
        transform_chain.append(torch.matmul(transform_chain[int(parents[i])], transforms_mat[:, i]))
    transforms = torch.stack(transform_chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_homogen), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot: bool = True):
    """smplx.lbs.lbs (call site head_detector/flame.py:152-161).

    betas [B,NB]; pose [B,3J]; v_template [B,V,3]; shapedirs [V,3,NB]; posedirs [P,3V];
    J_regressor [J,V]; parents [J]; lbs_weights [V,J]  ->  verts [B,V,3], joints [B,J,3]
    """
    bs = max(betas.shape[0], pose.shape[0])
    dtype, device = betas.dtype, betas.device
    # 1. shape blend
    v_shaped = v_template + torch.einsum("bl,mkl->bmk", betas, shapedirs)
    # 2. joints
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    # 3. pose blend shapes
    ident = torch.eye(3, dtype=dtype, device=device)
    if pose2rot:
        rot_mats = batch_rodrigues(pose.view(-1, 3)).view(bs, -1, 3, 3)
        pose_feature = (rot_mats[:, 1:, :, :] - ident).view(bs, -1)
        pose_offsets = torch.matmul(pose_feature, posedirs).view(bs, -1, 3)
    else:
        pose_feature = pose[:, 1:].view(bs, -1, 3, 3) - ident
        rot_mats = pose.view(bs, -1, 3, 3)
        pose_offsets = torch.matmul(pose_feature.view(bs, -1), posedirs).view(bs, -1, 3)
    v_posed = pose_offsets + v_shaped
    # 4. global joint transforms
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    # 5. skinning
    W = lbs_weights.unsqueeze(dim=0).expand(bs, -1, -1)
    num_joints = J_regressor.shape[0]
    T = torch.matmul(W, A.view(bs, num_joints, 16)).view(bs, -1, 4, 4)
    homogen = torch.ones([bs, v_posed.shape[1], 1], dtype=dtype, device=device)
    v_posed_homo = torch.cat([v_posed, homogen], dim=2)
    v_homo = torch.matmul(T, torch.unsqueeze(v_posed_homo, dim=-1))
    return v_homo[:, :, :3, 0], J_transformed


# --------------------------------------------------------------------------------------
# reference glue restatement
# --------------------------------------------------------------------------------------
def split_3dmm(p: torch.Tensor, consts: Optional[Dict[str, int]] = None) -> Dict[str, torch.Tensor]:
    """FlameParams.from_3dmm (head_detector/head_info.py:44-89): READ order is
    [shape, expression, jaw, rotation, eyeballs, neck, translation, scale]."""
    consts = consts or FLAME_CONSTS
    if p.size(1) != sum(consts.values()):
        raise ValueError(f"Invalid number of parameters. Expected: {sum(consts.values())}. Got: {p.size(1)}.")
    out, cur = {}, 0
    for name in ("shape", "expression", "jaw", "rotation", "eyeballs", "neck", "translation", "scale"):
        out[name] = p[:, cur : cur + consts[name]]
        cur += consts[name]
    return out


def join_3dmm(d: Dict[str, torch.Tensor]) -> torch.Tensor:
    """FlameParams.to_3dmm_tensor (head_info.py:91-109): WRITE order is
    [shape, expression, rotation, jaw, eyeballs, neck, translation, scale]."""
    return torch.cat([d[k] for k in ("shape", "expression", "rotation", "jaw", "eyeballs", "neck", "translation", "scale")], dim=1)


def rot_mat_from_6dof(v: torch.Tensor) -> torch.Tensor:
    """head_detector/utils.py:120-128."""
    assert v.shape[-1] == 6
    v = v.reshape(-1, 6)
    vx, vy = v[..., :3].clone(), v[..., 3:].clone()
    b1 = F.normalize(vx, dim=-1)
    b3 = F.normalize(torch.cross(b1, vy, dim=-1), dim=-1)
    b2 = -torch.cross(b1, b3, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


class FlameConstants:
    """The buffers FLAMELayer.__init__ registers (head_detector/flame.py:75-95)."""

    def __init__(self, model: Dict[str, np.ndarray], dtype=torch.float32):
        self.dtype = dtype
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).to(dtype)  # noqa: E731
        self.v_template = t(model["v_template"])  # [V,3]
        self.shapedirs = t(model["shapedirs"])  # [V,3,NB]
        nb = model["posedirs"].shape[-1]
        self.posedirs = t(np.reshape(model["posedirs"], [-1, nb]).T)  # [P,3V]   flame.py:86-88
        jr = model["J_regressor"]
        jr = jr.todense() if hasattr(jr, "todense") else jr
        self.J_regressor = t(jr)  # [J,V]
        parents = torch.as_tensor(np.asarray(model["kintree_table"][0]).astype(np.int64)).clone()
        parents[0] = -1  # flame.py:91-93
        self.parents = parents
        self.lbs_weights = t(model["weights"])  # [V,J]
        self.faces = np.asarray(model["f"]).astype(np.int64)

    def to(self, dtype):
        o = FlameConstants.__new__(FlameConstants)
        o.dtype = dtype
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
            setattr(o, k, getattr(self, k).to(dtype))
        o.parents, o.faces = self.parents, self.faces
        return o


def flame_forward(c: FlameConstants, fp: Dict[str, torch.Tensor], zero_rot: bool = False, zero_jaw: bool = False):
    """FLAMELayer.forward (head_detector/flame.py:122-169) with FLAME_CONSTS widths."""
    bs = fp["shape"].shape[0]
    dt = c.dtype
    betas = torch.cat(
        [
            fp["shape"].to(dt),
            torch.zeros(bs, 300 - fp["shape"].shape[1], dtype=dt),
            fp["expression"].to(dt),
            torch.zeros(bs, 100 - fp["expression"].shape[1], dtype=dt),
        ],
        dim=1,
    )
    neck = fp["neck"].to(dt) if fp["neck"].shape[1] else torch.zeros(bs, 3, dtype=dt)
    eyes = fp["eyeballs"].to(dt) if fp["eyeballs"].shape[1] else torch.zeros(bs, 6, dtype=dt)
    jaw = fp["jaw"].to(dt) if fp["jaw"].shape[1] else torch.zeros(bs, 3, dtype=dt)
    rotation = torch.zeros(bs, 3, dtype=dt)
    if zero_jaw:
        jaw = torch.zeros_like(jaw)
    full_pose = torch.cat([rotation, neck, jaw, eyes], dim=1)
    template = c.v_template.unsqueeze(0).repeat(bs, 1, 1)
    vertices, _ = lbs(betas, full_pose, template, c.shapedirs, c.posedirs, c.J_regressor, c.parents, c.lbs_weights)
    vertices = vertices.clone()
    vertices[:, :, 2] += MESH_OFFSET_Z
    if not zero_rot:
        R = rot_mat_from_6dof(fp["rotation"].to(dt))
        vertices = torch.matmul(R.unsqueeze(1), vertices.unsqueeze(-1))[..., 0]
    return vertices


def reproject(c: FlameConstants, params: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """reproject_spatial_vertices(flame, params, to_2d=False) (head_detector/flame.py:179-208).

    params [n,413] -> (vertices [n,V,3] un-rotated, R [n,3,3], projected [n,V,3])."""
    dt = c.dtype
    V = c.v_template.shape[0]
    if params.shape[0] == 0:
        return torch.zeros(0, V, 3, dtype=dt), torch.eye(3, dtype=dt).unsqueeze(0).expand(0, 3, 3), torch.zeros(0, V, 3, dtype=dt)
    fp = split_3dmm(params.to(dt))
    vertices = flame_forward(c, fp, zero_rot=True)
    R = rot_mat_from_6dof(fp["rotation"])
    rot_vertices = torch.matmul(R.unsqueeze(1), vertices.unsqueeze(-1))[..., 0]
    scale = torch.clamp(fp["scale"][:, None], 1e-8)
    projected = rot_vertices * scale + fp["translation"][:, None]
    return vertices, R, projected


def parse_predictions(c: FlameConstants, boxes: np.ndarray, params: torch.Tensor, padding, scale: float, image_size: int = 640):
    """The array math of HeadDetector._parse_predictions (head_detector/detector.py:61-90).

    returns (bbox_xywh int [n,4], vertices_3d [n,V,3] in original-image pixels,
             params_out [n,413] with only scale/=scale_factor (translation NOT un-padded))."""
    _, _, final = reproject(c, params)
    final = final.clone()
    final[:, :, 0] -= padding[0]
    final[:, :, 1] -= padding[1]
    final = (final / scale).numpy()
    b = np.asarray(boxes, dtype=np.float32).clip(0, image_size)
    b[:, [0, 2]] -= padding[0]
    b[:, [1, 3]] -= padding[1]
    b /= scale
    b = np.rint(b).astype(int)
    xywh = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1) if len(b) else np.zeros((0, 4), dtype=int)
    p = params.clone()
    p[:, 412] = p[:, 412] / scale
    return xywh, final, p


def limit_angle(angle: float, pi: float = 180.0) -> float:
    """head_detector/utils.py:131-143."""
    if angle < -pi:
        k = -2 * (int(angle / pi) // 2)
        angle = angle + k * pi
    if angle > pi:
        k = 2 * ((int(angle / pi) + 1) // 2)
        angle = angle - k * pi
    return angle


def calculate_rpy(rot6: torch.Tensor) -> Tuple[float, float, float]:
    """head_detector/utils.py:146-151 (+154-156): scipy Rotation.from_matrix(R^T).as_euler('xyz', deg)."""
    from scipy.spatial.transform import Rotation

    R = rot_mat_from_6dof(rot6.reshape(1, 6).float()).numpy()[0]
    angle = Rotation.from_matrix(np.transpose(R)).as_euler("xyz", degrees=True)
    roll, pitch, yaw = (limit_angle(a) for a in (angle[2], angle[0] - 180, angle[1]))
    return roll, pitch, yaw


# --------------------------------------------------------------------------------------
# synthetic FLAME constants (the licensed generic_model.pkl is a user-supplied asset)
# --------------------------------------------------------------------------------------
def synthetic_flame_model(seed: int = 3, V: int = 5023, NB: int = 400, NJ: int = 5, v_template: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """FLAME-shaped constants with the same shapes / sparsity pattern as generic_model.pkl
    (SURVEY.md 8(d) config 3): shapedirs,posedirs ~ N(0,1e-3); J_regressor rows sparse,
    non-negative, summing to 1; skinning weights = softmax rows; kintree [-1,0,1,1,1]."""
    rng = np.random.default_rng(seed)
    if v_template is None:
        # ellipsoidal head-sized point cloud (metres), deterministic
        u = rng.normal(size=(V, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        v_template = u * np.array([0.10, 0.16, 0.11]) + np.array([0.0, -0.03, -0.04])
    shapedirs = rng.normal(0, 1e-3, size=(V, 3, NB))
    # decaying spectrum like a PCA basis
    shapedirs *= (1.0 / np.sqrt(1.0 + np.arange(NB) / 10.0))[None, None, :]
    posedirs = rng.normal(0, 1e-3, size=(V, 3, (NJ - 1) * 9))
    J_regressor = np.zeros((NJ, V))
    for j in range(NJ):
        idx = rng.choice(V, size=64, replace=False)
        w = rng.random(64)
        J_regressor[j, idx] = w / w.sum()
    logits = rng.normal(0, 2.0, size=(V, NJ))
    weights = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    kintree = np.array([[4294967295, 0, 1, 1, 1][:NJ], list(range(NJ))], dtype=np.int64)
    faces = rng.integers(0, V, size=(9976, 3))
    return {
        "v_template": np.asarray(v_template, dtype=np.float64),
        "shapedirs": shapedirs,
        "posedirs": posedirs,
        "J_regressor": J_regressor,
        "kintree_table": kintree,
        "weights": weights,
        "f": faces,
    }


def synthetic_params(n: int, seed: int = 2, live_shape: int = 128, live_expr: int = 64, dtype=torch.float32) -> torch.Tensor:
    """Detector-like 413-vectors: shape/expr = 3*tanh(N(0,1)) on the
    live channels (rest exactly zero), jaw ~ N(0,0.1), rot6 ~ N(0,1), t ~ U(0,640), s ~ U(20,200).
    Layout is the detector OUTPUT layout read by from_3dmm: [shape300|expr100|jaw3|rot6|t3|s1]."""
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(n, NUM_PARAMS, dtype=torch.float64)
    p[:, :live_shape] = 3 * torch.tanh(torch.randn(n, live_shape, generator=g, dtype=torch.float64))
    p[:, 300 : 300 + live_expr] = 3 * torch.tanh(torch.randn(n, live_expr, generator=g, dtype=torch.float64))
    p[:, 400:403] = 0.1 * torch.randn(n, 3, generator=g, dtype=torch.float64)
    p[:, 403:409] = torch.randn(n, 6, generator=g, dtype=torch.float64)
    p[:, 409:412] = 640 * torch.rand(n, 3, generator=g, dtype=torch.float64)
    p[:, 412] = 20 + 180 * torch.rand(n, generator=g, dtype=torch.float64)
    return p.to(dtype)
