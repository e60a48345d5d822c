"""GPU letterbox: HeadDetector._transform_image (head_detector/detector.py:40-52) on the device.

The host only builds the two small fixed-point tap tables of ``cv::resize(..., INTER_LANCZOS4)`` for 8-bit images (8 short weights
+ one source coordinate per destination row / column, modules/imgproc/src/resize.cpp); the resampling, the constant border and the
u8 NHWC canvas the stem kernel consumes are produced by ``vgh_letterbox`` (csrc/letterbox.hip).  OpenCV itself is not present in
this image, so equality with cv2 is "parity unpinned" (see oracle/letterbox_oracle.py); the border colour is what
``cv2.copyMakeBorder(..., value=127)`` produces for a 3-channel image: a bare Python number becomes ``cv::Scalar(127, 0, 0, 0)``,
i.e. (127, 0, 0)."""
from __future__ import annotations

import functools
import math
from typing import Tuple

import numpy as np
import torch

from . import _lib

PAD_VALUE = (127, 0, 0)  # cv2.copyMakeBorder(..., cv2.BORDER_CONSTANT, value=127) on an RGB image (detector.py:50)
_COEF_SCALE = 2048  # INTER_RESIZE_COEF_SCALE = 1 << 11
_S45 = 0.70710678118654752440084436210485
_CS = ((1, 0), (-_S45, -_S45), (0, 1), (_S45, -_S45), (-1, 0), (_S45, _S45), (0, -1), (-_S45, _S45))


def _lanczos4(x: float) -> np.ndarray:
    """interpolateLanczos4: float weights of the 8 taps for the fractional position x."""
    c = np.zeros(8, dtype=np.float32)
    if x < np.finfo(np.float32).eps:
        c[3] = 1.0
        return c
    y0 = -(x + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    total = np.float32(0.0)
    for i in range(8):
        y = -(x + 3 - i) * math.pi * 0.25
        c[i] = np.float32((_CS[i][0] * s0 + _CS[i][1] * c0) / (y * y))
        total = np.float32(total + c[i])
    return (c * np.float32(np.float32(1.0) / total)).astype(np.float32)


@functools.lru_cache(maxsize=64)
def axis_tables(src: int, dst: int) -> Tuple[np.ndarray, np.ndarray]:
    """(ofs [dst] int32, coef [dst,8] int16) for one axis, as resize.cpp computes xofs/ialpha (yofs/ibeta)."""
    scale = 1.0 / (dst / src)
    ofs = np.zeros(dst, dtype=np.int32)
    coef = np.zeros((dst, 8), dtype=np.int16)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = math.floor(float(f))
        ofs[d] = s
        w = _lanczos4(float(np.float32(f - np.float32(s))))
        coef[d] = np.clip(np.rint((w * np.float32(_COEF_SCALE)).astype(np.float64)), -32768, 32767).astype(np.int16)
    return ofs, coef


def geometry(h: int, w: int, S: int) -> Tuple[int, int, int, int, float]:
    """detector.py:41-46,48-49 -> (new_h, new_w, pad_x, pad_y, scale)."""
    if h > w:
        new_h, new_w = S, int(w * S / h)
    else:
        new_h, new_w = int(h * S / w), S
    return new_h, new_w, (S - new_w) // 2, (S - new_h) // 2, S / max(h, w)


def letterbox(image, S: int, device: torch.device, out: torch.Tensor = None) -> Tuple[torch.Tensor, Tuple[int, int], float]:
    """uint8 [H,W,3+] (numpy, or a torch uint8 tensor already on the GPU) -> (uint8 [S,S,3] on the GPU, (pad_x, pad_y), scale)."""
    if not torch.cuda.is_available():
        raise _lib.VghError("letterbox: no GPU available; the HIP kernel is the only implementation")
    lib = _lib.load()
    src = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] < 3:
        raise ValueError(f"letterbox expects a uint8 image [H,W,3]; got {src.dtype} {tuple(src.shape)}")
    src = src.to(device, non_blocking=True).contiguous()
    h, w, cn = src.shape
    new_h, new_w, px, py, scale = geometry(h, w, S)
    if new_h < 1 or new_w < 1:
        raise ValueError(f"image {h}x{w} is too elongated for a {S}x{S} letterbox")
    xo, al = axis_tables(w, new_w)
    yo, be = axis_tables(h, new_h)
    tabs = [torch.from_numpy(t).to(device, non_blocking=True) for t in (xo, al, yo, be)]
    dst = out if out is not None else torch.empty(S, S, 3, dtype=torch.uint8, device=device)
    pad = (_lib.C.c_uint8 * 3)(*PAD_VALUE)
    _lib.check(lib.vgh_letterbox(src.data_ptr(), h, w, cn, w * cn, tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), tabs[3].data_ptr(), new_w, new_h, px, py,
                                 pad, dst.data_ptr(), S, torch.cuda.current_stream(device).cuda_stream))
    return dst, (px, py), scale
