"""TEST INFRASTRUCTURE ONLY -- CPU restatement of HeadDetector._transform_image (head_detector/detector.py:40-52):
``cv2.resize(image, (new_w, new_h), interpolation=cv2.INTER_LANCZOS4)`` + ``cv2.copyMakeBorder(..., BORDER_CONSTANT, value=127)``.

PARITY UNPINNED: OpenCV (opencv-python, requirements.txt:4, unpinned version) is absent from this image and from
/root/reference, and no fixture of the reference holds a resized image.  What follows restates the published algorithm of
``cv::resize`` for 8-bit images (modules/imgproc/src/resize.cpp: ``resizeGeneric_`` with ``HResizeLanczos4<uchar,int,short>`` /
``VResizeLanczos4<uchar,int,short, FixedPtCast<int,uchar,INTER_RESIZE_COEF_BITS*2>>`` and ``interpolateLanczos4``):

  * destination pixel dx samples the source at fx = float((dx + 0.5) * scale_x - 0.5), scale_x = src_w / dst_w (double),
    sx = floor(fx); the 8 taps are source columns sx-3 .. sx+4, out-of-range taps replicate the edge pixel
  * tap weights: interpolateLanczos4(fx - sx) in float (sin/cos in double), normalised to sum 1 in float, then converted to
    fixed point short = saturate_cast<short>(w * 2048)  (round-half-to-even), no sum correction
  * horizontal pass accumulates int32, vertical pass accumulates int32, result = saturate_u8((v + 2^21) >> 22)

and of ``copyMakeBorder`` with a Python scalar ``value=127``: the binding converts a bare number to ``cv::Scalar(127, 0, 0, 0)``,
so the border of a 3-channel image is (127, 0, 0), not grey (the well-known "blue border" behaviour of ``value=255`` on BGR
images).  Both statements come from the OpenCV sources as published, not from a run of cv2: first contact with the real
package should re-run tests/test_letterbox."""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS
PAD_VALUE = (127, 0, 0)


def lanczos4_coeffs(x: float) -> np.ndarray:
    """interpolateLanczos4 (imgproc/src/precomp / resize helpers): 8 float weights for the fractional offset x in [0,1)."""
    s45 = 0.70710678118654752440084436210485
    cs = ((1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45))
    c = np.zeros(8, dtype=np.float32)
    if x < np.finfo(np.float32).eps:
        c[3] = 1.0
        return c
    y0 = -(float(x) + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    total = np.float32(0.0)
    for i in range(8):
        y = -(float(x) + 3 - i) * math.pi * 0.25
        c[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        total = np.float32(total + c[i])
    inv = np.float32(np.float32(1.0) / total)
    return (c * inv).astype(np.float32)


def resize_tables(src: int, dst: int) -> Tuple[np.ndarray, np.ndarray]:
    """(ofs [dst] int32 = floor source coordinate, coef [dst,8] int16 fixed-point weights) along one axis."""
    scale = 1.0 / (dst / src)  # inv_scale = (double)dst/src; scale = 1./inv_scale  (resize.cpp)
    ofs = np.zeros(dst, dtype=np.int32)
    coef = np.zeros((dst, 8), dtype=np.int16)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        fr = np.float32(f - np.float32(s))
        ofs[d] = s
        w = lanczos4_coeffs(float(fr))
        q = np.rint((w * np.float32(COEF_SCALE)).astype(np.float32).astype(np.float64))  # cvRound: round half to even
        coef[d] = np.clip(q, -32768, 32767).astype(np.int16)
    return ofs, coef


def resize_lanczos4(image: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """cv2.resize(image, (new_w, new_h), interpolation=cv2.INTER_LANCZOS4) for uint8 [H,W,C]."""
    h, w = image.shape[:2]
    if (h, w) == (new_h, new_w):
        return image.copy()
    xofs, alpha = resize_tables(w, new_w)
    yofs, beta = resize_tables(h, new_h)
    src = image.astype(np.int64)
    # horizontal pass: int32 rows (wrap-around like C int arithmetic; never reached by real coefficient tables)
    cols = np.clip(xofs[:, None] - 3 + np.arange(8)[None, :], 0, w - 1)  # [new_w,8]
    hbuf = np.einsum("hxkc,xk->hxc", src[:, cols], alpha.astype(np.int64))  # [H,new_w,C]
    hbuf = ((hbuf + 2**31) % 2**32) - 2**31
    rows = np.clip(yofs[:, None] - 3 + np.arange(8)[None, :], 0, h - 1)  # [new_h,8]
    v = np.einsum("ykxc,yk->yxc", hbuf[rows], beta.astype(np.int64))
    v = ((v + 2**31) % 2**32) - 2**31
    out = (v + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(h: int, w: int, S: int) -> Tuple[int, int, int, int, float]:
    """detector.py:41-46,48-49 -> (new_h, new_w, pad_x, pad_y, scale)."""
    if h > w:
        new_h, new_w = S, int(w * S / h)
    else:
        new_h, new_w = int(h * S / w), S
    return new_h, new_w, (S - new_w) // 2, (S - new_h) // 2, S / max(h, w)


def transform_image(image: np.ndarray, S: int):
    """_transform_image up to (not including) the float conversion: uint8 [S,S,3] canvas, (pad_x, pad_y), scale."""
    h, w = image.shape[:2]
    new_h, new_w, px, py, scale = letterbox_geometry(h, w, S)
    r = resize_lanczos4(image[..., :3], new_w, new_h)
    canvas = np.empty((S, S, 3), dtype=np.uint8)
    canvas[...] = np.array(PAD_VALUE, dtype=np.uint8)
    canvas[py : py + new_h, px : px + new_w] = r
    return canvas, (px, py), scale
