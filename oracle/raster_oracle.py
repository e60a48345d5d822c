"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's result-side consumers:

  * ``rasterize``                = Sim3DR ``_rasterize``  (head_detector/Sim3DR/lib/rasterize_kernel.cpp:219-293,
                                   barycentric weights ``get_point_weight`` :53-79) behind ``Sim3DR.rasterize``
                                   (head_detector/Sim3DR/Sim3DR.py:17-38: depth buffer = -1e8, alpha = 1)
  * ``compute_ncc_color_codes``  = head_detector/pncc_processor.py:40-55
  * ``pncc_triangles``           = the triangle filter of ``PNCCProcessor.__init__`` (pncc_processor.py:58-64)
  * ``pncc_image``               = ``PNCCProcessor.__call__`` (pncc_processor.py:66-73) incl. its in-place ``z *= -1``
  * ``refined_head_bbox``        = head_detector/utils.py:26-35

PINNED: ``rasterize`` is checked bit-for-bit against the reference's own C++ (``oracle/_ref/libsim3dr_ref.so``, built by
``oracle/build_ref.py`` from the sources under /root/reference) in tests/test_oracle_golden.py, and the committed fixture
tests/golden/raster_ref.npz holds outputs of that library.  All arithmetic is float32 in the reference's operation order
(no fused multiply-add: an x86-64 baseline build has none)."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32


def _weights(px: np.ndarray, py: np.ndarray, p0, p1, p2) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """get_point_weight (rasterize_kernel.cpp:53-79) for arrays of pixel centres; float32, reference op order."""
    v0x, v0y = f32(p2[0] - p0[0]), f32(p2[1] - p0[1])
    v1x, v1y = f32(p1[0] - p0[0]), f32(p1[1] - p0[1])
    v2x, v2y = (px - p0[0]).astype(f32), (py - p0[1]).astype(f32)
    dot00 = f32(f32(v0x * v0x) + f32(v0y * v0y))
    dot01 = f32(f32(v0x * v1x) + f32(v0y * v1y))
    dot02 = (v0x * v2x).astype(f32) + (v0y * v2y).astype(f32)
    dot11 = f32(f32(v1x * v1x) + f32(v1y * v1y))
    dot12 = (v1x * v2x).astype(f32) + (v1y * v2y).astype(f32)
    den = f32(f32(dot00 * dot11) - f32(dot01 * dot01))
    inv = f32(0.0) if den == 0 else f32(f32(1.0) / den)
    u = ((dot11 * dot02).astype(f32) - (dot01 * dot12).astype(f32)).astype(f32) * inv
    v = ((dot00 * dot12).astype(f32) - (dot01 * dot02).astype(f32)).astype(f32) * inv
    u, v = u.astype(f32), v.astype(f32)
    w0 = ((f32(1.0) - u).astype(f32) - v).astype(f32)
    return w0, v, u


def rasterize(vertices: np.ndarray, triangles: np.ndarray, colors: np.ndarray, bg: np.ndarray, reverse: bool = False,
              depth: Optional[np.ndarray] = None) -> np.ndarray:
    """Sim3DR.rasterize(vertices, triangles, colors, bg=bg) (Sim3DR.py:17-38) -> the painted copy of ``bg`` (uint8 [H,W,C]).
    Triangles are processed in order; a pixel is overwritten when its interpolated depth is strictly greater."""
    img = np.ascontiguousarray(bg).copy()
    h, w, c = img.shape
    ver = np.ascontiguousarray(vertices, dtype=f32)
    col = np.ascontiguousarray(colors, dtype=f32)
    tri = np.ascontiguousarray(triangles, dtype=np.int32)
    zb = np.full((h, w), f32(-1e8), dtype=f32) if depth is None else depth
    with np.errstate(all="ignore"):
        for t in range(tri.shape[0]):
            i0, i1, i2 = (int(k) for k in tri[t])
            p0, p1, p2 = ver[i0], ver[i1], ver[i2]
            xs, ys = (p0[0], p1[0], p2[0]), (p0[1], p1[1], p2[1])
            if not all(math.isfinite(float(q)) for q in xs + ys):
                continue  # (int)ceil(nan) is undefined in C; such triangles never cover a pixel in practice
            x_min = max(int(math.ceil(min(xs))), 0)
            x_max = min(int(math.floor(max(xs))), w - 1)
            y_min = max(int(math.ceil(min(ys))), 0)
            y_max = min(int(math.floor(max(ys))), h - 1)
            if x_max < x_min or y_max < y_min:
                continue
            py, px = np.meshgrid(np.arange(y_min, y_max + 1, dtype=f32), np.arange(x_min, x_max + 1, dtype=f32), indexing="ij")
            w0, w1, w2 = _weights(px, py, p0, p1, p2)
            inside = (w2 > 0) & (w1 > 0) & (w0 > 0)
            pd = ((w0 * p0[2]).astype(f32) + (w1 * p1[2]).astype(f32)).astype(f32) + (w2 * p2[2]).astype(f32)
            pd = pd.astype(f32)
            sub = zb[y_min : y_max + 1, x_min : x_max + 1]
            win = inside & (pd > sub)
            if not win.any():
                continue
            rows = slice(h - 1 - y_max, h - y_min) if reverse else slice(y_min, y_max + 1)
            view = img[rows, x_min : x_max + 1]
            if reverse:
                view = view[::-1]
            for k in range(c):
                pc = ((w0 * col[i0, k]).astype(f32) + (w1 * col[i1, k]).astype(f32)).astype(f32) + (w2 * col[i2, k]).astype(f32)
                val = (f32(0.0) * view[..., k].astype(f32)).astype(f32) + (f32(255.0) * pc.astype(f32)).astype(f32)  # alpha = 1
                # (unsigned char)float: truncation toward zero, then the low 8 bits (what x86 cvttss2si + mov does)
                q = (val.astype(f32).astype(np.int64) & 0xFF).astype(np.uint8)
                view[..., k] = np.where(win, q, view[..., k])
            sub[win] = pd[win]
    return img


def compute_ncc_color_codes(template: np.ndarray, subset: Optional[np.ndarray] = None) -> np.ndarray:
    """pncc_processor.py:40-55 (min/max with ``initial=0`` over the subset, applied to ALL vertices)."""
    sub = template[subset] if subset is not None else template
    u_min = sub.min(axis=0, keepdims=True, initial=0)
    u_max = sub.max(axis=0, keepdims=True, initial=0)
    return (template - u_min) / (u_max - u_min)


def pncc_triangles(full_faces: np.ndarray, indices: np.ndarray) -> np.ndarray:
    """pncc_processor.py:62: keep the triangles whose three vertices are all in ``indices`` (order preserved)."""
    keep = np.isin(full_faces, indices).all(axis=1)
    return np.ascontiguousarray(full_faces[keep]).astype(np.int32)


def pncc_image(image_shape: Sequence[int], heads_vertices: List[np.ndarray], triangles: np.ndarray, colors: np.ndarray) -> np.ndarray:
    """PNCCProcessor.__call__ (pncc_processor.py:66-73).  NOTE the reference negates z IN PLACE on each head's vertices
    (``vertices[:, 2] *= -1`` on the array the head owns): the arrays passed here are mutated the same way."""
    out = np.zeros(tuple(image_shape), dtype=np.uint8)
    col = colors.astype(np.float32)
    for vertices in heads_vertices:
        vertices[:, 2] *= -1
        cur = rasterize(vertices, triangles, col, bg=out)
        m = cur.sum(2) != 0
        out[m] = cur[m]
    return out


def refined_head_bbox(vertices: np.ndarray, head_indices: np.ndarray) -> Tuple[int, int, int, int]:
    """utils.py:26-35 -> (x, y, w, h): int() truncation of the min / max over the HEAD_INDICES subset."""
    pts = np.take(vertices, np.array(head_indices), axis=0)
    x, y, x1, y1 = (int(v) for v in (pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()))
    return x, y, x1 - x, y1 - y


def random_mesh(seed: int, n_side: int = 12, size: float = 90.0, centre=(64.0, 64.0), depth_scale: float = 40.0):
    """A bumpy, partially self-occluding grid mesh (vertices [V,3], triangles [T,3] int32, colors [V,3] in [0,1])."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-0.5, 0.5, n_side, dtype=np.float64)
    yy, xx = np.meshgrid(g, g, indexing="ij")
    ang = rng.uniform(0, 2 * np.pi)
    xr = np.cos(ang) * xx - np.sin(ang) * yy
    yr = np.sin(ang) * xx + np.cos(ang) * yy
    z = np.sin(4 * xx + rng.uniform(0, 3)) * np.cos(3 * yy + rng.uniform(0, 3))
    fold = 0.35 * np.sin(6 * yy)  # folds the sheet over itself so the z-test matters
    ver = np.stack([centre[0] + size * (xr + fold) + rng.normal(0, 0.3, xx.shape), centre[1] + size * yr + rng.normal(0, 0.3, xx.shape), depth_scale * z], -1)
    ver = ver.reshape(-1, 3).astype(np.float32)
    idx = np.arange(n_side * n_side).reshape(n_side, n_side)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tri = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int32)
    tri = tri[rng.permutation(tri.shape[0])]
    col = rng.uniform(0, 1, (ver.shape[0], 3)).astype(np.float32)
    return ver, tri, col
