"""Python host of the MI355X VGGHeads engine: owns one vgh_net (network program + arena on one GPU), the
post-network device buffers, and a HIP stream.  torch is used for device memory and stream plumbing only;
every arithmetic step is a libvgh kernel.

``VGHeadsEngine.model(image)`` honours the contract of the TorchScript blob the reference calls at
head_detector/detector.py:58-59:  f32[B,3,S,S] -> (boxes f32[B,1000,4], scores f32[B,1000,1], flame f32[B,1000,413]).
``detect`` is the batched twin of HeadDetector._postprocess (yolo_heads_post_prediction_callback.py:41-99).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib, arch
from .flame import FLAMELayer

TUNING_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning")


@dataclass
class Detections:
    """Fixed-capacity slabs on the GPU (+ per-head vertices for the valid heads, image-major order)."""

    boxes: torch.Tensor  # [B, keep, 4] xyxy in network (padded-square) pixels
    scores: torch.Tensor  # [B, keep]
    flame_params: torch.Tensor  # [B, keep, 413]
    counts: torch.Tensor  # [B] int32
    head_image: Optional[torch.Tensor] = None  # [n] image index of every valid head
    vertices_3d: Optional[torch.Tensor] = None  # [n, V, 3] projected vertices (reproject_spatial_vertices(..., to_2d=False)[2])


class VGHeadsEngine:
    def __init__(self, variant: str = "vgg_heads_l", state_dict: Optional[Dict[str, np.ndarray]] = None, image_size: int = 640, max_batch: int = 1,
                 device: Optional[int] = None, seed: int = 1, pre_nms_top_k: int = 1000, keep_top_k: int = 100, use_tuning: bool = True,
                 arena_batch: Optional[int] = None, precision: str = "bf16"):
        if variant not in arch.VARIANTS:
            raise ValueError(f"unknown model variant {variant!r}; known: {sorted(arch.VARIANTS)}")
        if not torch.cuda.is_available():
            raise _lib.VghError("VGHeadsEngine needs a GPU: the HIP path is the only implementation (no CPU fallback)")
        self.lib = _lib.load()
        self.variant, self.image_size, self.max_batch = variant, image_size, max_batch
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        if state_dict is None:
            state_dict = arch.random_state_dict(variant, seed)  # synthetic weights of the exact architecture
        self.precision = precision
        self.program = arch.build_program(variant, state_dict, image_size, precision)
        P = self.program
        # the conv loader addresses an input tensor with 32-bit byte offsets: keep every arena tensor below 2 GiB by running
        # large batches through the network in chunks (post-network stages always see the whole batch)
        per_image = max(bf["h"] * bf["w"] * bf["pitch"] * (4 if bf["is_f32"] else 2) for bf in P.bufs)
        self.arena_batch = max(1, min(max_batch, ((1 << 31) - 1) // per_image, arena_batch or max_batch))
        w, b = P.arrays()
        bufs = (_lib.BufDesc * len(P.bufs))(*[_lib.BufDesc(bf["h"], bf["w"], bf["pitch"], bf["is_f32"]) for bf in P.bufs])
        fields = [f for f, _ in _lib.OpDesc._fields_]
        ops = (_lib.OpDesc * len(P.ops))(*[_lib.OpDesc(**{f: (op.get(f, 0) if f != "in_buf" else max(op[f], 0)) for f in fields}) for op in P.ops])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vgh_net_create(self.device_index, image_size, self.arena_batch, bufs, len(P.bufs), ops, len(P.ops), _lib.ptr(w), w.size, _lib.ptr(b), b.size, C.byref(h)))
        self._net = h
        self.stream = torch.cuda.Stream(device=self.device)
        self.A = sum(lv["h"] * lv["w"] for lv in P.levels)
        self.pre_k, self.keep_k = min(pre_nms_top_k, self.A), keep_top_k
        B, A, k, kk = max_batch, self.A, self.pre_k, keep_top_k
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.boxes_all = torch.empty(B, A, 4, **f32)
        self.scores_all = torch.empty(B, A, **f32)
        self.idx = torch.empty(B, k, **i32)
        self.cand_scores = torch.empty(B, k, **f32)
        self.cand_boxes = torch.empty(B, k, 4, **f32)
        self.cand_flame = torch.empty(B, k, _lib.NUM_FLAME_PARAMS, **f32)
        self.keep_idx = torch.empty(B, kk, **i32)
        self.counts = torch.empty(B, **i32)
        self.out_boxes = torch.empty(B, kk, 4, **f32)
        self.out_scores = torch.empty(B, kk, **f32)
        self.out_flame = torch.empty(B, kk, _lib.NUM_FLAME_PARAMS, **f32)
        self._levels = None
        self._graph_key = None
        if use_tuning and precision == "bf16":
            self.load_tuning()

    # ---------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_net", None) is not None:
            self.lib.vgh_net_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def flops_per_image(self) -> float:
        return self.program.flops

    def _sp(self) -> int:
        return self.stream.cuda_stream

    def _levels_arr(self, B: int):
        P = self.program
        arr = (_lib.HeadLevel * len(P.levels))()
        for i, lv in enumerate(P.levels):
            arr[i] = _lib.HeadLevel(self.lib.vgh_net_buffer(self._net, lv["buf"]), lv["h"], lv["w"], lv["pitch"], lv["stride"])
        return arr

    def buffer(self, name_or_id, B: int) -> torch.Tensor:
        """Copy of an activation buffer as a torch tensor [B,h,w,pitch] (tests / debugging)."""
        P = self.program
        bid = name_or_id if isinstance(name_or_id, int) else next(i for i, bf in enumerate(P.bufs) if bf["name"] == name_or_id)
        bf = P.bufs[bid]
        n = B * bf["h"] * bf["w"] * bf["pitch"]
        self.stream.synchronize()

        class _Ext:  # zero-copy alias of arena memory through the CUDA array interface
            pass

        e = _Ext()
        e.__cuda_array_interface__ = dict(shape=(n,), typestr="<f4" if bf["is_f32"] else "<i2", data=(int(self.lib.vgh_net_buffer(self._net, bid)), False), version=2)
        t = torch.as_tensor(e, device=self.device)
        if not bf["is_f32"]:
            t = t.view(torch.bfloat16)
        return t.clone().view(B, bf["h"], bf["w"], bf["pitch"])

    # ---------------------------------------------------------------------------------------------------
    def _check_images(self, images: torch.Tensor) -> Tuple[int, int]:
        S = self.image_size
        if images.dtype == torch.float32 and images.dim() == 4 and images.shape[1:] == (3, S, S):
            fmt = _lib.VGH_IMG_F32_NCHW
        elif images.dtype == torch.uint8 and images.dim() == 4 and images.shape[1:] == (S, S, 3):
            fmt = _lib.VGH_IMG_U8_NHWC
        else:
            raise ValueError(f"images must be f32 [B,3,{S},{S}] or u8 [B,{S},{S},3]; got {images.dtype} {tuple(images.shape)}")
        if not images.is_cuda or not images.is_contiguous():
            raise ValueError("images must be a contiguous GPU tensor")
        B = images.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        return B, fmt

    def forward_net(self, images: torch.Tensor, use_graph: bool = False) -> int:
        """Backbone + neck + heads for one arena-sized batch: leaves the fp32 prediction buffers inside the arena. Returns B."""
        B, fmt = self._check_images(images)
        if B > self.arena_batch:
            raise ValueError(f"forward_net handles at most arena_batch={self.arena_batch} images; use forward_candidates() for larger batches")
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        if use_graph:
            key = (images.data_ptr(), B, fmt)
            if self._graph_key != key:
                _lib.check(self.lib.vgh_net_forward(self._net, images.data_ptr(), fmt, B, self._sp()))  # warm: lazy attributes before capture
                self.stream.synchronize()
                _lib.check(self.lib.vgh_net_capture(self._net, images.data_ptr(), fmt, B, self._sp()))
                self._graph_key = key
            _lib.check(self.lib.vgh_net_forward_graph(self._net, self._sp()))
        else:
            _lib.check(self.lib.vgh_net_forward(self._net, images.data_ptr(), fmt, B, self._sp()))
        return B

    def candidates(self, B: int, at: int = 0):
        """K6 + K7 + K6b on the engine stream for the B images currently in the arena: boxes/scores for all anchors, top-k,
        gather + FLAME fix-up; results land in rows [at, at+B) of the batch-level candidate tensors."""
        lv = self._levels_arr(B)
        P = self.program
        sp = self._sp()
        ba, sa, ix = self.boxes_all[at:], self.scores_all[at:], self.idx[at:]
        cs, cb, cf = self.cand_scores[at:], self.cand_boxes[at:], self.cand_flame[at:]
        _lib.check(self.lib.vgh_head_decode(lv, len(P.levels), B, _lib.ptr(ba), _lib.ptr(sa), sp))
        _lib.check(self.lib.vgh_topk(_lib.ptr(sa), B, self.A, self.pre_k, _lib.ptr(ix), _lib.ptr(cs), sp))
        _lib.check(self.lib.vgh_gather_candidates(lv, len(P.levels), B, self.A, P.shape_c, P.expr_c, _lib.ptr(ba), _lib.ptr(ix), self.pre_k, _lib.ptr(cb), _lib.ptr(cf), sp))

    def forward_candidates(self, images: torch.Tensor, use_graph: bool = False) -> int:
        """Network + candidate stages for a batch of any size <= max_batch, in arena-sized chunks."""
        B = images.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        for i in range(0, B, self.arena_batch):
            n = self.forward_net(images[i : i + self.arena_batch], use_graph and B <= self.arena_batch)
            self.candidates(n, at=i)
        return B

    def model(self, images: torch.Tensor, use_graph: bool = False):
        """Drop-in for ``self.model(image)`` (detector.py:58-59)."""
        B = self.forward_candidates(images, use_graph)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return self.cand_boxes[:B], self.cand_scores[:B].unsqueeze(-1), self.cand_flame[:B]

    def detect(self, images: torch.Tensor, confidence_threshold: float = 0.5, iou_threshold: float = 0.5, flame: Optional[FLAMELayer] = None,
               unpad: Optional[torch.Tensor] = None, use_graph: bool = False) -> Detections:
        """net -> top-k -> NMS (every image) -> optional FLAME decode of every surviving head."""
        B = self.forward_candidates(images, use_graph)
        sp = self._sp()
        _lib.check(self.lib.vgh_nms(_lib.ptr(self.cand_boxes), _lib.ptr(self.cand_scores), B, self.pre_k, float(confidence_threshold), float(iou_threshold), self.keep_k,
                                    _lib.ptr(self.keep_idx), _lib.ptr(self.counts), sp))
        _lib.check(self.lib.vgh_compact(_lib.ptr(self.cand_boxes), _lib.ptr(self.cand_scores), _lib.ptr(self.cand_flame), B, self.pre_k, _lib.ptr(self.keep_idx), self.keep_k,
                                        _lib.ptr(self.out_boxes), _lib.ptr(self.out_scores), _lib.ptr(self.out_flame), sp))
        det = Detections(self.out_boxes[:B], self.out_scores[:B], self.out_flame[:B], self.counts[:B])
        if flame is not None:
            with torch.cuda.stream(self.stream):
                valid = torch.arange(self.keep_k, device=self.device)[None, :] < det.counts[:, None]
                det.head_image = valid.nonzero()[:, 0]
                params = det.flame_params[valid]  # [n,413], image-major (host sync: n is data dependent)
                up = unpad[det.head_image] if unpad is not None else None
                P = self.program
                _, _, det.vertices_3d = flame.decode(params, unpad=up, shape_live=P.shape_c, expr_live=P.expr_c, want_vertices=False)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return det

    # ---------------------------------------------------------------------------------------------------
    def profile_ops(self, images: torch.Tensor) -> List[dict]:
        """Per-op device time (HIP events on the engine stream) with the algorithmic FLOPs of each op."""
        B, fmt = self._check_images(images)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        ms = (C.c_float * len(self.program.ops))()
        _lib.check(self.lib.vgh_net_profile(self._net, images.data_ptr(), fmt, B, self._sp(), ms))
        out = []
        for op, t in zip(self.program.ops, ms):
            fl = 2.0 * op["macs"] * B
            out.append(dict(name=op["name"], kind=op["kind"], ms=float(t), gflop=fl / 1e9, tflops=(fl / (t * 1e-3) / 1e12) if t > 0 else 0.0, gemm=op["gemm"]))
        return out

    def set_cfg(self, op_index: int, cfg: int):
        _lib.check(self.lib.vgh_net_set_cfg(self._net, op_index, cfg))
        self._graph_key = None

    def cfg_names(self) -> List[str]:
        return [self.lib.vgh_conv_cfg_name(i).decode() for i in range(self.lib.vgh_conv_num_cfgs())]

    def load_tuning(self, path: Optional[str] = None) -> int:
        """Apply a measured per-layer tile table: {gemm-shape key: cfg name}. Missing file -> heuristic choice."""
        path = path or os.path.join(TUNING_DIR, "conv_cfg.json")
        if not os.path.exists(path):
            return 0
        table = json.load(open(path))
        names = {n: i for i, n in enumerate(self.cfg_names())}
        applied = 0
        for i, op in enumerate(self.program.ops):
            if op["kind"] != 1:
                continue
            key = tuning_key(op, self.max_batch)
            name = table.get(key)
            if name in names:
                self.set_cfg(i, names[name])
                applied += 1
        return applied


def tuning_key(op: dict, batch: int) -> str:
    m, n, k = op["gemm"]
    bucket = 1 if batch <= 2 else (8 if batch <= 16 else 32)
    return f"b{bucket}_m{m}_n{n}_k{k}_ks{op['ksize']}_s{op['stride']}"
