"""Result-side consumers on the GPU, with the reference's names:

  rasterize(vertices, triangles, colors, bg=..., reverse=False)   <- head_detector/Sim3DR/Sim3DR.py:17-38
  compute_ncc_color_codes(template, subset)                       <- head_detector/pncc_processor.py:40-55
  PNCCProcessor(...)(image, heads)                                <- head_detector/pncc_processor.py:58-73
  refined_head_bbox(vertices)                                     <- head_detector/utils.py:26-35

The reference ships three mesh assets next to its sources (assets/full_faces.npy, assets/v_template.npy,
assets/flame_indices/{head_w_ears,head_indices}.npy).  They are data the user supplies here (``assets_dir`` = the
reference's ``head_detector/assets`` directory, or the arrays themselves); nothing is bundled.
All arithmetic runs in libvgh (csrc/raster.hip); there is no CPU path."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .head_info import Bbox, HeadMetadata


def _dev() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.VghError("head_detector_amd.pncc needs a GPU: the HIP rasteriser is the only implementation")
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def rasterize(vertices: np.ndarray, triangles: np.ndarray, colors: np.ndarray, bg: Optional[np.ndarray] = None, height: Optional[int] = None,
              width: Optional[int] = None, channel: Optional[int] = None, reverse: bool = False) -> np.ndarray:
    """Drop-in for Sim3DR.rasterize: z-buffer render of one mesh onto ``bg`` (uint8 [H,W,C]); like the reference, ``bg`` itself is
    painted and returned."""
    lib, dev = _lib.load(), _dev()
    if bg is not None:
        height, width, channel = bg.shape
    else:
        assert height is not None and width is not None and channel is not None
        bg = np.zeros((height, width, channel), dtype=np.uint8)
    if bg.dtype != np.uint8:
        raise ValueError("bg must be uint8 [H,W,C]")
    v = torch.from_numpy(np.ascontiguousarray(vertices, dtype=np.float32)).to(dev)
    t = torch.from_numpy(np.ascontiguousarray(triangles, dtype=np.int32)).to(dev)
    c = torch.from_numpy(np.ascontiguousarray(colors, dtype=np.float32)).to(dev)
    if c.shape != (v.shape[0], channel):
        raise ValueError(f"colors must be [{v.shape[0]},{channel}], got {tuple(c.shape)}")
    if t.numel() and (int(t.min()) < 0 or int(t.max()) >= v.shape[0]):
        raise ValueError("triangle index out of range")
    img = torch.from_numpy(np.ascontiguousarray(bg)).to(dev)
    zbuf = torch.empty(height * width, dtype=torch.int64, device=dev)
    _lib.check(lib.vgh_rasterize(v.data_ptr(), t.data_ptr(), t.shape[0], c.data_ptr(), channel, img.data_ptr(), height, width, int(bool(reverse)), zbuf.data_ptr(), _stream()))
    out = img.cpu().numpy()
    if bg.flags.c_contiguous and bg.flags.writeable:
        bg[...] = out
        return bg
    return out


def compute_ncc_color_codes(template_face: np.ndarray, subset_indexes: Optional[np.ndarray] = None) -> np.ndarray:
    if not isinstance(template_face, np.ndarray):
        raise ValueError(f"Argument template_face must be a numpy array, got type {type(template_face)}")
    if len(template_face.shape) != 2 or template_face.shape[1] != 3:
        raise ValueError(f"Argument template_face must have shape [N,3], got shape {template_face.shape}")
    if subset_indexes is not None and not isinstance(subset_indexes, np.ndarray):
        raise ValueError(f"Argument subset_indexes must be a numpy array, got type {type(subset_indexes)}")
    sub = template_face[subset_indexes] if subset_indexes is not None else template_face
    u_min = sub.min(axis=0, keepdims=True, initial=0)
    u_max = sub.max(axis=0, keepdims=True, initial=0)
    return (template_face - u_min) / (u_max - u_min)


class MeshAssets:
    """The reference's mesh assets (head_detector/assets).  ``MeshAssets.load(dir)`` reads the four .npy files."""

    def __init__(self, full_faces: np.ndarray, v_template: np.ndarray, head_w_ears: np.ndarray, head_indices: Optional[np.ndarray] = None):
        self.full_faces = np.asarray(full_faces)
        self.v_template = np.asarray(v_template)
        self.head_w_ears = np.asarray(head_w_ears)
        self.head_indices = None if head_indices is None else np.asarray(head_indices)

    @classmethod
    def load(cls, assets_dir: str) -> "MeshAssets":
        def need(rel):
            p = os.path.join(assets_dir, rel)
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p} not found: pass assets_dir=<reference checkout>/head_detector/assets (the mesh assets are not bundled)")
            return p

        hi = os.path.join(assets_dir, "flame_indices", "head_indices.npy")
        return cls(np.load(need("full_faces.npy")), np.load(need("v_template.npy")), np.load(need(os.path.join("flame_indices", "head_w_ears.npy"))),
                   np.load(hi, allow_pickle=True)[()] if os.path.exists(hi) else None)


class PNCCProcessor:
    """pncc_processor.py:58-73.  ``__call__(image, heads)`` -> uint8 [H,W,3] PNCC image of all heads."""

    def __init__(self, assets: Union[MeshAssets, str]):
        if isinstance(assets, str):
            assets = MeshAssets.load(assets)
        self.indices = assets.head_w_ears
        keep = np.isin(assets.full_faces, self.indices).all(axis=1)  # pncc_processor.py:62
        self.triangles = np.ascontiguousarray(assets.full_faces[keep]).astype(np.int32)
        self.colors = compute_ncc_color_codes(assets.v_template, self.indices)
        self._dev_cache = None

    def _device_arrays(self, dev):
        if self._dev_cache is None or self._dev_cache[0] != dev:
            self._dev_cache = (dev, torch.from_numpy(self.triangles).to(dev), torch.from_numpy(self.colors.astype(np.float32)).to(dev))
        return self._dev_cache[1:]

    def render(self, image_shape: Sequence[int], vertices: torch.Tensor) -> torch.Tensor:
        """vertices [n,V,3] float32 on the GPU (NOT modified; z is negated inside the kernel) -> uint8 [H,W,3] on the GPU."""
        lib, dev = _lib.load(), vertices.device
        H, W = int(image_shape[0]), int(image_shape[1])
        tri, col = self._device_arrays(dev)
        v = vertices.detach().to(torch.float32).contiguous()
        n, V = (v.shape[0], v.shape[1]) if v.dim() == 3 else (0, col.shape[0])
        if n and V != col.shape[0]:
            raise ValueError(f"vertices have {V} points, the colour table {col.shape[0]}")
        img = torch.empty(H, W, 3, dtype=torch.uint8, device=dev)
        zbuf = torch.empty(H * W, dtype=torch.int64, device=dev)
        _lib.check(lib.vgh_pncc_render(v.data_ptr() if n else None, n, V, tri.data_ptr(), tri.shape[0], col.data_ptr(), img.data_ptr(), H, W, zbuf.data_ptr(), _stream()))
        return img

    def __call__(self, image: np.ndarray, heads: List[HeadMetadata]) -> np.ndarray:
        dev = _dev()
        if image.ndim != 3 or image.shape[2] != 3:
            raise ValueError("image must be [H,W,3]")
        if not heads:
            return np.zeros_like(image)
        verts = np.stack([np.asarray(h.vertices_3d, dtype=np.float32) for h in heads])
        out = self.render(image.shape, torch.from_numpy(verts).to(dev)).cpu().numpy()
        for h in heads:  # the reference's side effect: `vertices[:, 2] *= -1` on the array each head owns (pncc_processor.py:69-70)
            h.vertices_3d[:, 2] *= -1
        return out.astype(image.dtype, copy=False)


def refined_head_bbox(vertices: Union[np.ndarray, torch.Tensor], head_indices: np.ndarray) -> Union[Bbox, List[Bbox]]:
    """utils.py:26-35.  ``vertices`` [V,3] -> Bbox, or [n,V,3] -> list of Bbox (one kernel for all heads)."""
    lib, dev = _lib.load(), _dev()
    v = vertices if isinstance(vertices, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(vertices, dtype=np.float32))
    v = v.to(dev, torch.float32).contiguous()
    single = v.dim() == 2
    if single:
        v = v.unsqueeze(0)
    idx = torch.from_numpy(np.ascontiguousarray(np.array(head_indices), dtype=np.int32)).to(dev)
    if idx.numel() == 0 or int(idx.min()) < 0 or int(idx.max()) >= v.shape[1]:
        raise ValueError("head_indices empty or out of range")
    out = torch.empty(v.shape[0], 4, dtype=torch.int32, device=dev)
    _lib.check(lib.vgh_refined_head_bbox(v.data_ptr(), v.shape[0], v.shape[1], idx.data_ptr(), idx.numel(), out.data_ptr(), _stream()))
    boxes = [Bbox(x=int(r[0]), y=int(r[1]), w=int(r[2]), h=int(r[3])) for r in out.cpu().numpy()]
    return boxes[0] if single else boxes
