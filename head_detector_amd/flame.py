"""FLAME layer on libvgh: same names / call signatures as head_detector/flame.py
(get_flame_model, FLAMELayer, reproject_spatial_vertices), arithmetic in csrc/flame.hip.

The licensed FLAME ``generic_model.pkl`` is a user-supplied asset exactly as in the reference
(``flame_path`` argument, head_detector/flame.py:18-24,43); nothing here ships or substitutes it.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .head_info import FLAME_CONSTS, FlameParams
from .utils import rot_mat_from_6dof

MAX_SHAPE = 300
MAX_EXPRESSION = 100
ROT_COEFFS = 3
JAW_COEFFS = 3
EYE_COEFFS = 6
NECK_COEFFS = 3
MESH_OFFSET_Z = 0.05


class _ChArray:
    """Stand-in for chumpy.Ch objects inside the official FLAME pickle (only their numeric payload is needed)."""

    def __init__(self, *a, **k):
        self.x = None

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})
        if self.x is None:
            for v in self.__dict__.values():
                if isinstance(v, np.ndarray):
                    self.x = v
                    break

    @property
    def r(self):
        return np.asarray(self.x)


class _FlameUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChArray
        return super().find_class(module, name)


def _to_np(a, dtype=np.float64) -> np.ndarray:
    if hasattr(a, "todense"):  # scipy sparse J_regressor
        a = a.todense()
    if hasattr(a, "r"):
        a = a.r
    return np.asarray(a, dtype=dtype)


def get_flame_model(flame_path: Optional[str] = None) -> Dict[str, Any]:
    """head_detector/flame.py:18-24: default path = generic_model.pkl next to this file; latin1 pickle."""
    if flame_path is None:
        flame_path = os.path.join(os.path.dirname(__file__), "generic_model.pkl")
    with open(flame_path, "rb") as f:  # FileNotFoundError if the user has not supplied the asset
        return dict(_FlameUnpickler(f, encoding="latin1").load())


class FLAMELayer:
    """Drop-in for head_detector/flame.py:37-169. Buffers are registered as in flame.py:75-95 (as attributes:
    v_template [V,3], shapedirs [V,3,NB], posedirs [P,3V], J_regressor [J,V], parents [J], lbs_weights [V,J],
    faces_tensor [F,3]); the constants also live packed on the GPU inside a vgh_flame handle."""

    def __init__(self, consts: Dict[str, Any] = None, batch_size: int = 1, flame_path: Optional[str] = None, *, model: Optional[Dict[str, Any]] = None,
                 device: Optional[torch.device] = None, max_heads: int = 1024) -> None:
        self.flame_constants = FLAME_CONSTS if consts is None else consts
        self.batch_size = batch_size
        self.dtype = torch.float32
        self.flame_model = model if model is not None else get_flame_model(flame_path)
        m = self.flame_model
        self.faces = _to_np(m["f"], np.int64)
        self.faces_tensor = torch.from_numpy(self.faces.astype(np.int64))
        self.v_template = torch.from_numpy(_to_np(m["v_template"]).astype(np.float32))
        self.shapedirs = torch.from_numpy(_to_np(m["shapedirs"]).astype(np.float32))
        pd = _to_np(m["posedirs"])
        self.posedirs = torch.from_numpy(np.reshape(pd, [-1, pd.shape[-1]]).T.astype(np.float32).copy())
        self.J_regressor = torch.from_numpy(_to_np(m["J_regressor"]).astype(np.float32))
        parents = _to_np(m["kintree_table"], np.int64)[0].copy()
        parents[0] = -1
        self.parents = torch.from_numpy(parents)
        self.lbs_weights = torch.from_numpy(_to_np(m["weights"]).astype(np.float32))
        self._handle = None
        self._device = None
        self._max_heads = max_heads
        if device is not None:
            self.to(device)

    # -- device management -------------------------------------------------------------------------
    def to(self, device) -> "FLAMELayer":
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.VghError("FLAMELayer runs on the GPU only (libvgh); there is no CPU implementation in this package")
        if self._handle is not None and self._device == device:
            return self
        self._free()
        lib = _lib.load()
        V, NB = self.v_template.shape[0], self.shapedirs.shape[2]
        NJ = self.J_regressor.shape[0]
        h = C.c_void_p()
        arrs = [np.ascontiguousarray(t.numpy()) for t in (self.v_template, self.shapedirs, self.posedirs, self.J_regressor, self.lbs_weights)]
        par = np.ascontiguousarray(self.parents.numpy().astype(np.int32))
        idx = device.index if device.index is not None else torch.cuda.current_device()
        _lib.check(lib.vgh_flame_create(idx, V, NB, NJ, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]), _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), _lib.ptr(par), _lib.ptr(arrs[4]),
                                        self._max_heads, C.byref(h)))
        self._handle, self._device = h, torch.device("cuda", idx)
        return self

    def _free(self):
        if self._handle is not None:
            _lib.load().vgh_flame_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    @property
    def max_heads(self) -> int:
        return self._max_heads

    @property
    def num_vertices(self) -> int:
        return int(self.v_template.shape[0])

    def _need_handle(self):
        if self._handle is None:
            if not torch.cuda.is_available():
                raise _lib.VghError("FLAMELayer: no GPU available and no CPU fallback exists")
            self.to(torch.device("cuda", torch.cuda.current_device()))
        return self._handle

    # -- the fused path used by the detector ------------------------------------------------------------
    def decode(self, params: Tensor, unpad: Optional[Tensor] = None, shape_live: int = 300, expr_live: int = 100, want_vertices: bool = True,
               want_projected: bool = True) -> Tuple[Optional[Tensor], Tensor, Optional[Tensor]]:
        """params [n,413] (GPU) -> (vertices [n,V,3] | None, R [n,3,3], projected [n,V,3] | None); see vgh_flame_decode."""
        h = self._need_handle()
        if params.dim() != 2 or params.size(1) != _lib.NUM_FLAME_PARAMS:
            raise ValueError(f"Invalid number of parameters. Expected: {_lib.NUM_FLAME_PARAMS}. Got: {params.size(-1)}.")
        p = params.detach().to(self._device, torch.float32).contiguous()
        n, V = p.shape[0], self.v_template.shape[0]
        verts = torch.empty(n, V, 3, dtype=torch.float32, device=self._device) if want_vertices else None
        proj = torch.empty(n, V, 3, dtype=torch.float32, device=self._device) if want_projected else None
        rot = torch.empty(n, 3, 3, dtype=torch.float32, device=self._device)
        up = unpad.detach().to(self._device, torch.float32).contiguous() if unpad is not None else None
        _lib.check(_lib.load().vgh_flame_decode(h, _lib.ptr(p), n, shape_live, expr_live, _lib.ptr(up), _lib.ptr(verts), _lib.ptr(rot), _lib.ptr(proj),
                                                 torch.cuda.current_stream(self._device).cuda_stream))
        return verts, rot, proj

    # -- reference-compatible general forward ---------------------------------------------------------------
    def forward(self, flame_params: FlameParams, zero_rot: bool = False, zero_jaw: bool = False) -> Tensor:
        """head_detector/flame.py:122-169 for any ``consts`` widths: betas / full_pose assembly (host glue, torch),
        lbs on the GPU (vgh_flame_lbs), z += 0.05, optional global rotation."""
        h = self._need_handle()
        dev = self._device
        c = self.flame_constants
        bs = flame_params.shape.shape[0]
        f32 = lambda t: t.detach().to(dev, torch.float32)  # noqa: E731
        z = lambda w: torch.zeros(bs, w, dtype=torch.float32, device=dev)  # noqa: E731
        betas = torch.cat([f32(flame_params.shape), z(MAX_SHAPE - c["shape"]), f32(flame_params.expression), z(MAX_EXPRESSION - c["expression"])], dim=1)
        neck = f32(flame_params.neck) if 0 not in flame_params.neck.shape else z(NECK_COEFFS)
        eyes = f32(flame_params.eyeballs) if 0 not in flame_params.eyeballs.shape else z(EYE_COEFFS)
        jaw = f32(flame_params.jaw) if 0 not in flame_params.jaw.shape else z(JAW_COEFFS)
        if zero_jaw:
            jaw = torch.zeros_like(jaw)
        full_pose = torch.cat([z(ROT_COEFFS), neck, jaw, eyes], dim=1).contiguous()
        betas = betas.contiguous()
        NB, NJ = self.shapedirs.shape[2], self.J_regressor.shape[0]
        if betas.shape[1] != NB or full_pose.shape[1] != 3 * NJ:
            raise ValueError(f"FLAME model expects {NB} betas / {3 * NJ} pose coefficients, got {betas.shape[1]} / {full_pose.shape[1]}")
        verts = torch.empty(bs, self.v_template.shape[0], 3, dtype=torch.float32, device=dev)
        if bs:
            _lib.check(_lib.load().vgh_flame_lbs(h, _lib.ptr(betas), _lib.ptr(full_pose), bs, _lib.ptr(verts), None, torch.cuda.current_stream(dev).cuda_stream))
        verts[:, :, 2] += MESH_OFFSET_Z
        if not zero_rot:
            R = rot_mat_from_6dof(f32(flame_params.rotation)).type(verts.dtype)
            verts = torch.matmul(R.unsqueeze(1), verts.unsqueeze(-1))[..., 0]
        return verts

    __call__ = forward


def reproject_spatial_vertices(flame: FLAMELayer, flame_params: Tensor, to_2d: bool = True, subset_indexes=None) -> Tuple[Tensor, Tensor, Tensor]:
    """Drop-in for head_detector/flame.py:179-208: flame_params [..., 413] -> (vertices, rotation_mat, projected)."""
    shape = flame_params.size()
    V = flame.v_template.size(0)
    if flame_params.size(0) == 0:  # flame.py:186-189
        dev = flame_params.device
        projected = torch.zeros((0, V, 2 if to_2d else 3), device=dev)
        vertices = torch.zeros((0, V, 3), device=dev)
        rotation_mat = torch.eye(3, device=dev).unsqueeze(0).expand(0, 3, 3)
    else:
        if flame_params.size(1) != sum(FLAME_CONSTS.values()):
            raise ValueError(f"Invalid number of parameters. Expected: {sum(FLAME_CONSTS.values())}. Got: {flame_params.size(1)}.")
        vertices, rotation_mat, projected = flame.decode(flame_params)
    if subset_indexes is not None:
        projected = projected[:, subset_indexes]
    if to_2d:
        projected = projected[..., :2]
    projected = projected.reshape(*shape[:-1], *projected.size()[-2:]).contiguous()
    return vertices, rotation_mat, projected
