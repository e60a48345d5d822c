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
    """Fixed-capacity slabs on the GPU + per-head outputs (valid heads only, image-major order).  Nothing here forces a
    host synchronisation until ``num_heads`` / ``vertices_3d`` / ``head_image`` / ``head_pose`` are read: the per-head
    tensors are capacity-sized on the device and the live count is a device scalar.

    Lifetime: what ``VGHeadsEngine.detect`` / ``model`` return by default are FRESH tensors the caller owns (like the reference's
    TorchScript module).  ``select``, ``detect(reuse_outputs=True)`` and ``model(copy=False)`` -- the benchmark's zero-allocation
    variants -- return views of engine-owned buffers (or of the ``slot`` passed in): valid until the next call that writes the
    same buffers, and in overlap mode only after ``join()``."""

    boxes: torch.Tensor  # [B, keep, 4] xyxy in network (padded-square) pixels
    scores: torch.Tensor  # [B, keep]
    flame_params: torch.Tensor  # [B, keep, 413]
    counts: torch.Tensor  # [B] int32
    n_heads: Optional[torch.Tensor] = None  # [1] int32 on the device: number of valid rows in the per-head tensors
    head_image_cap: Optional[torch.Tensor] = None  # [capacity] int32
    vertices_cap: Optional[torch.Tensor] = None  # [capacity, V, 3] projected (and un-padded) vertices
    rpy_cap: Optional[torch.Tensor] = None  # [capacity, 3] roll, pitch, yaw (degrees)
    _n: Optional[int] = None

    @property
    def num_heads(self) -> int:
        if self._n is None:
            self._n = int(self.n_heads.item()) if self.n_heads is not None else 0  # the one host sync
        return self._n

    @property
    def head_image(self) -> Optional[torch.Tensor]:
        """[n] image index of every valid head."""
        return None if self.head_image_cap is None else self.head_image_cap[: self.num_heads].long()

    @property
    def vertices_3d(self) -> Optional[torch.Tensor]:
        """[n, V, 3] = reproject_spatial_vertices(..., to_2d=False)[2] (un-padded when ``unpad`` was given)."""
        return None if self.vertices_cap is None else self.vertices_cap[: self.num_heads]

    @property
    def head_pose(self) -> Optional[torch.Tensor]:
        """[n, 3] (roll, pitch, yaw) degrees = calculate_rpy of every valid head."""
        return None if self.rpy_cap is None else self.rpy_cap[: self.num_heads]


def _alias(ptr: int, shape, typestr: str, device: torch.device) -> torch.Tensor:
    """Zero-copy torch view of library-owned device memory (CUDA array interface)."""

    class _Ext:
        pass

    e = _Ext()
    e.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)
    return torch.as_tensor(e, device=device)


class VGHeadsEngine:
    def __init__(self, variant: str = "vgg_heads_l", state_dict: Optional[Dict[str, np.ndarray]] = None, image_size: int = 640, max_batch: int = 1,
                 device: Optional[int] = None, seed: int = 1, pre_nms_top_k: int = 1000, keep_top_k: int = 100, use_tuning: bool = True,
                 arena_batch: Optional[int] = None, precision: str = "bf16", fp8_scales: Optional[Dict[str, float]] = None, calib_images: Optional[torch.Tensor] = None,
                 fp8_min_px: int = 40, latency_lanes: bool = True):
        """``precision="fp8"`` (r05): the bf16 network with OCP-e4m3 links between 3x3 / stride-1 convs (arch.build_program).  Every link needs the largest
        activation it will carry: ``fp8_scales`` {link name: max|x|} from an earlier ``calibrate_fp8``, or ``calib_images`` (u8 NHWC / f32 NCHW GPU batch of
        representative inputs) to run that calibration now; with neither, two seeded random images are used -- adequate for the synthetic benchmark, NOT for
        real weights and real photographs.  ``precision="int8"``: the same links as signed bytes (VGH_FMT_I8, the reference exporter's QuantizationMode.INT8), same
        calibration, same arguments."""
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
        self.fp8_scales = None
        if precision in arch.Q8_PRECISIONS:
            if fp8_scales is None:
                if calib_images is None:
                    # not silent: with real weights and real photographs, activations above the random-image maximum times the headroom are CLAMPED in
                    # the epilogue -- the caller has to know that these scales are for plumbing and the synthetic benchmark only
                    import warnings

                    warnings.warn(f"VGHeadsEngine(precision={precision!r}) without fp8_scales / calib_images: the 8-bit link scales are calibrated on two seeded RANDOM "
                                  "images -- adequate for the synthetic benchmark, not for real weights and photographs (pass calib_images or fp8_scales)", stacklevel=2)
                    calib_images = torch.randint(0, 256, (2, image_size, image_size, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(12345)).to(torch.device("cuda", self.device_index))
                fp8_scales = calibrate_fp8(variant, state_dict, image_size, calib_images, self.device_index, fp8_min_px)
            self.fp8_scales = dict(fp8_scales)
        self.program = arch.build_program(variant, state_dict, image_size, precision, fp8_scales=self.fp8_scales, fp8_min_px=fp8_min_px)
        if max_batch <= LATENCY_MAX_BATCH and latency_lanes:
            # single-image engines (HeadDetector's default): the three heads on lane streams with exact dependencies, each right behind its pyramid level (r06)
            arch.schedule_latency(self.program)
        P = self.program
        # the conv loader addresses an input tensor with 32-bit byte offsets: keep every arena tensor below 2 GiB by running
        # large batches through the network in chunks (post-network stages always see the whole batch)
        per_image = max(bf["h"] * bf["w"] * bf["pitch"] * arch.FMT_BYTES[bf["is_f32"]] for bf in P.bufs)
        self.arena_batch = max(1, min(max_batch, ((1 << 31) - 1) // per_image, arena_batch or max_batch))
        w, b = P.arrays()
        bufs = (_lib.BufDesc * len(P.bufs))(*[_lib.BufDesc(bf["h"], bf["w"], bf["pitch"], bf["is_f32"], float(bf.get("scale", 0.0))) for bf in P.bufs])
        fields = [f for f, _ in _lib.OpDesc._fields_]
        ops = (_lib.OpDesc * len(P.ops))(*[_lib.OpDesc(**{f: (op.get(f, 0) if f != "in_buf" else max(op[f], 0)) for f in fields}) for op in P.ops])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vgh_net_create(self.device_index, image_size, self.arena_batch, bufs, len(P.bufs), ops, len(P.ops), _lib.ptr(w), w.size, _lib.ptr(b), b.size, C.byref(h)))
        self._net = h
        if int(self.lib.vgh_net_b2b_pairs(h)) != len(arch.b2b_pairs(P)):  # (the PMC tools and the algorithmic-bytes accounting use the Python predicate)
            raise _lib.VghError(f"back-to-back pairs: the library found {int(self.lib.vgh_net_b2b_pairs(h))}, arch.b2b_pairs {len(arch.b2b_pairs(P))}")
        self.stream = torch.cuda.Stream(device=self.device)
        self.A = sum(lv["h"] * lv["w"] for lv in P.levels)
        self.pre_k, self.keep_k = min(pre_nms_top_k, self.A), keep_top_k
        B, A, k, kk = max_batch, self.A, self.pre_k, keep_top_k
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        # the fused detector (csrc/detect.hip) owns the post-network scratch; torch sees it through zero-copy aliases
        cfg = _lib.DetectCfg()
        cfg.n_levels = len(P.levels)
        for i, lv in enumerate(P.levels):
            cfg.level_buf[i], cfg.level_h[i], cfg.level_w[i], cfg.level_pitch[i], cfg.level_stride[i] = lv["buf"], lv["h"], lv["w"], lv["pitch"], lv["stride"]
        cfg.shape_live, cfg.expr_live, cfg.pre_k, cfg.keep_k, cfg.max_batch = P.shape_c, P.expr_c, k, kk, B
        d = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vgh_detector_create(self._net, None, C.byref(cfg), C.byref(d)))
        self._det = d
        self._flame_ref = None
        pb, ps, pf = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.vgh_detector_candidate_buffers(self._det, C.byref(pb), C.byref(ps), C.byref(pf)))
        self.cand_boxes = _alias(pb.value, (B, k, 4), "<f4", self.device)
        self.cand_scores = _alias(ps.value, (B, k), "<f4", self.device)
        self.cand_flame = _alias(pf.value, (B, k, _lib.NUM_FLAME_PARAMS), "<f4", self.device)
        sc = lambda which: self.lib.vgh_detector_scratch(self._det, which)
        self.boxes_all = _alias(sc(_lib.SCRATCH_BOXES_ALL), (B, A, 4), "<f4", self.device)
        self.scores_all = _alias(sc(_lib.SCRATCH_SCORES_ALL), (B, A), "<f4", self.device)
        self.idx = _alias(sc(_lib.SCRATCH_TOPK_IDX), (B, k), "<i4", self.device)
        self.keep_idx = _alias(sc(_lib.SCRATCH_KEEP_IDX), (B, kk), "<i4", self.device)
        self.counts = torch.empty(B, **i32)
        self.out_boxes = torch.empty(B, kk, 4, **f32)
        self.out_scores = torch.empty(B, kk, **f32)
        self.out_flame = torch.empty(B, kk, _lib.NUM_FLAME_PARAMS, **f32)
        self.n_heads = torch.zeros(1, **i32)
        self._head_out = None  # (capacity, head_image, proj, rpy) allocated on first FLAME use
        self._levels = None
        self._graph_key = None
        self._use_tuning = bool(use_tuning and precision in ("bf16", "fp8", "int8", "fp16", "fp16x3", "bf16x3"))  # the split modes have their own keys (precision prefix) and tile set
        self.nsplit = 1
        if self._use_tuning:
            self.load_tuning()

    # ---------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_det", None) is not None:
            self.lib.vgh_detector_destroy(self._det)
            self._det = None
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
        """Copy of an activation buffer as a torch tensor [B,h,w,pitch] (tests / debugging): bf16 / fp32 as stored; the two-plane split
        formats joined to fp32 (hi + lo, fp16 planes: hi + lo / 2048 -- exact in fp32)."""
        P = self.program
        bid = name_or_id if isinstance(name_or_id, int) else next(i for i, bf in enumerate(P.bufs) if bf["name"] == name_or_id)
        bf = P.bufs[bid]
        fmt = bf["is_f32"]
        n = B * bf["h"] * bf["w"] * bf["pitch"]
        self.stream.synchronize()
        if fmt == arch.FMT_FP8:  # e4m3 link: decoded to fp32 (stored * scale)
            t = _alias(self.lib.vgh_net_buffer(self._net, bid), (n,), "|u1", self.device).clone().view(torch.float8_e4m3fn).float() * bf["scale"]
            return t.view(B, bf["h"], bf["w"], bf["pitch"])
        if fmt == arch.FMT_I8:  # int8 link
            t = _alias(self.lib.vgh_net_buffer(self._net, bid), (n,), "|i1", self.device).clone().float() * bf["scale"]
            return t.view(B, bf["h"], bf["w"], bf["pitch"])
        if fmt in (arch.FMT_BF16X2, arch.FMT_F16X2):
            t = _alias(self.lib.vgh_net_buffer(self._net, bid), (2 * n,), "<i2", self.device).clone().view(B, bf["h"], bf["w"], 2, bf["pitch"])
            t = t.view(torch.bfloat16 if fmt == arch.FMT_BF16X2 else torch.float16).float()
            return t[..., 0, :] + t[..., 1, :] * (1.0 if fmt == arch.FMT_BF16X2 else 1.0 / 2048.0)
        t = _alias(self.lib.vgh_net_buffer(self._net, bid), (n,), "<f4" if fmt == arch.FMT_F32 else "<i2", self.device)
        if fmt == arch.FMT_BF16:
            t = t.view(torch.bfloat16)
        elif fmt == arch.FMT_F16:
            t = t.view(torch.float16)
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
            if getattr(self, "_overlap", False):
                raise ValueError("hipGraph replay cannot honour the prediction-buffer guard of overlap mode: use one or the other")
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

    def set_split(self, nsplit: int = 1):
        """vgh_net_set_split: run each batch as ``nsplit`` independent sub-batches on the net's lane streams (throughput mode)."""
        _lib.check(self.lib.vgh_net_set_split(self._net, int(nsplit)))
        self.nsplit = int(nsplit)
        if self._use_tuning:
            self.load_tuning()  # the table may hold tile choices measured in this split mode

    def set_b2b(self, enable=True):
        """vgh_net_set_b2b (r06): a stage's downsample and the conv1|conv2 behind it as ONE back-to-back-GEMM launch (default) or as their two launches -- the same
        output bits; unfused, the tensor between them exists in the arena (per-op inspection).  ``b2b_pairs``: how many such pairs the program has.
        Default (1): the stage-1 pair runs on its persistent "t" tile (csrc/ds_b2b.hip) and, for u8 images, the STEM conv runs inside that launch as a bf16 x 3 split GEMM
        (the stem tensor is never written either; exact products, another fp32 summation order: a flipped bf16 ulp in ~4e-5 of the stem values).  ``enable=3``: the t tile
        fed by the stem launch -- bit-identical to ``enable=2`` (every pair on the implicit-GEMM b2b tile) and, for untuned engines, to ``enable=0``."""
        _lib.check(self.lib.vgh_net_set_b2b(self._net, int(enable) if enable in (2, 3) and enable is not True else int(bool(enable))))
        self._graph_key = None

    @property
    def stem_fused(self) -> bool:
        """u8 forwards run the stem conv inside the stage-1 pair's launch (vgh_net_stem_fused): the stem tensor is neither written nor read."""
        return bool(self.lib.vgh_net_stem_fused(self._net))

    @property
    def b2b_pairs(self) -> int:
        return int(self.lib.vgh_net_b2b_pairs(self._net))

    def set_fuse_stem(self, enable: bool = True):
        """EXPERIMENTS build only (VGH_LIB_PATH=libvgh_exp.so; r06): stem + stage-1 downsample as one kernel (csrc/stem_ds.hip: less HBM traffic, same time --
        measured r03, not part of the product library) or as the two launches.  Results are bit-identical."""
        if not hasattr(self.lib, "vgh_net_set_fuse_stem"):
            raise _lib.VghError("set_fuse_stem: the fused stem + downsample kernel exists in the -DVGH_EXPERIMENTS build only (python -m head_detector_amd.build --experiments)")
        _lib.check(self.lib.vgh_net_set_fuse_stem(self._net, int(bool(enable))))
        self._graph_key = None

    def set_overlap(self, enable: bool = True):
        """Throughput mode (vgh_detector_set_overlap): ``select`` / the select half of ``detect`` run on a detector-owned side
        stream underneath the next batch's network.  Call ``join()`` before reading a batch's outputs."""
        _lib.check(self.lib.vgh_detector_set_overlap(self._det, int(bool(enable))))
        self._overlap = bool(enable)

    def tune_overlap(self, images: torch.Tensor, flame=None, forwards: int = 8, confidence_threshold: float = 0.5) -> Dict[str, float]:
        """Keep the post-stage overlap only if it pays on THIS engine: times ``forwards`` pipelined forward + detect steps with the post stages on the side stream and with
        them serial, leaves the faster setting on, returns both times (ms per forward).  Normally the overlap wins by ~0.1 ms; an engine whose side stream is starved loses
        2.5 - 5 ms per forward with it (profiles/r05_engine_sequence.txt) -- bench.py makes the same comparison before its timed steps."""
        import time

        B = int(images.shape[0])
        unpad = torch.tensor([[0.0, 0.0, 1.0]], device=self.device).expand(B, 3).contiguous()

        def run(n):
            for _ in range(n):
                self.forward_net(images)
                self.candidates(B)
                self.select(B, confidence_threshold=confidence_threshold, iou_threshold=0.5, flame=flame, unpad=unpad)
            self.join()
            torch.cuda.synchronize(self.device)

        out = {}
        for on in (True, False):
            self.join()
            self.set_overlap(on)
            run(2)
            t = time.perf_counter()
            run(forwards)
            out["overlapped" if on else "serial"] = (time.perf_counter() - t) / forwards * 1e3
        self.set_overlap(out["overlapped"] <= out["serial"])
        return out

    def join(self):
        """Make the engine stream (and the caller's current stream) wait for the last queued select."""
        _lib.check(self.lib.vgh_detector_join(self._det, self._sp()))
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def candidates(self, B: int, at: int = 0, lazy_flame: bool = False):
        """K6 + K7 + K6b for the B images currently in the arena (vgh_detector_decode_candidates): boxes/scores for all anchors,
        top-k, gather + FLAME fix-up; results land in rows [at, at+B) of the batch-level candidate tensors.  In overlap mode
        they are queued on the detector's side stream (``join()`` before reading them).

        ``lazy_flame`` (r06, vgh_detector_set_lazy_flame): gather the candidates' BOXES only; the 413-vectors of the survivors are then built by the next ``select`` straight
        from the prediction buffers (the candidate FLAME tensor -- 1.65 MB per image for typically a handful of survivors -- is neither written nor read; ``cand_flame``
        is stale).  Contract: that ``select`` is queued before the next forward touches the arena.  Same bits in the detections."""
        self._set_lazy_flame(lazy_flame)
        _lib.check(self.lib.vgh_detector_decode_candidates(self._det, B, at, self._sp()))

    def _set_lazy_flame(self, enable: bool):
        if getattr(self, "_lazy_flame", False) != bool(enable):
            _lib.check(self.lib.vgh_detector_set_lazy_flame(self._det, int(bool(enable))))
            self._lazy_flame = bool(enable)

    def forward_candidates(self, images: torch.Tensor, use_graph: bool = False, lazy_flame: bool = False) -> int:
        """Network + candidate stages for a batch of any size <= max_batch (arena-sized chunks): one vgh_detector_candidates call.  ``lazy_flame``: see ``candidates``
        (a batch that runs in several arena chunks gathers eagerly whatever the flag says)."""
        B, fmt = self._check_images(images)
        if use_graph and B <= self.arena_batch:
            self.forward_net(images, True)
            self.candidates(B, lazy_flame=lazy_flame)
            return B
        self._set_lazy_flame(lazy_flame)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        _lib.check(self.lib.vgh_detector_candidates(self._det, images.data_ptr(), fmt, B, self._sp()))
        return B

    def _join_if_overlap(self):
        if getattr(self, "_overlap", False):
            _lib.check(self.lib.vgh_detector_join(self._det, self._sp()))

    def model(self, images: torch.Tensor, use_graph: bool = False, copy: bool = True):
        """Drop-in for ``self.model(image)`` (detector.py:58-59): fresh tensors; ``copy=False`` returns views of the engine's candidate
        buffers instead (overwritten by the next forward)."""
        B = self.forward_candidates(images, use_graph)
        self._join_if_overlap()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        out = self.cand_boxes[:B], self.cand_scores[:B].unsqueeze(-1), self.cand_flame[:B]
        return tuple(t.clone() for t in out) if copy else out

    def new_output_slot(self, flame: Optional[FLAMELayer] = None, batch: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """A private set of result buffers for ``select(..., slot=...)``: with two of them the results of batch s stay intact (for a
        consumer on another stream, e.g. dist.DetectionGatherer) while batch s+1 is being written."""
        B, kk = batch or self.max_batch, self.keep_k
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        # ONE zero-filled block, five views (r06: five torch.zeros were five fill kernels queued ahead of the network -- 75 us of a 2.0-ms single-image call,
        # profiles/r06_latency_trace_l1.txt); every view starts 16-byte aligned
        nb, ns, nf = B * kk * 4, (B * kk + 3) // 4 * 4, B * kk * _lib.NUM_FLAME_PARAMS
        nf4, nc = (nf + 3) // 4 * 4, (B + 3) // 4 * 4
        flat = torch.zeros(nb + ns + nf4 + nc + 4, **f32)
        ints = flat.view(torch.int32)
        slot = dict(boxes=flat[:nb].view(B, kk, 4), scores=flat[nb:nb + B * kk].view(B, kk), flame=flat[nb + ns:nb + ns + nf].view(B, kk, _lib.NUM_FLAME_PARAMS),
                    counts=ints[nb + ns + nf4:nb + ns + nf4 + B], n_heads=ints[nb + ns + nf4 + nc:nb + ns + nf4 + nc + 1])
        if flame is not None:
            cap = min(B * self.keep_k, flame.max_heads)  # per-head rows beyond the live count are never written: no need to clear them
            slot.update(cap=cap, head_image=torch.empty(cap, **i32), proj=torch.empty(cap, flame.num_vertices, 3, **f32), rpy=torch.empty(cap, 3, **f32))
        return slot

    def _detect_out(self, B: int, flame: Optional[FLAMELayer], unpad: Optional[torch.Tensor], n_heads_out: Optional[torch.Tensor] = None,
                    slot: Optional[Dict[str, torch.Tensor]] = None) -> Tuple["_lib.DetectOut", Detections]:
        o = _lib.DetectOut()
        ob, os_, of, oc = (slot["boxes"], slot["scores"], slot["flame"], slot["counts"]) if slot is not None else (self.out_boxes, self.out_scores, self.out_flame, self.counts)
        o.boxes_dev, o.scores_dev, o.flame_dev, o.counts_dev = ob.data_ptr(), os_.data_ptr(), of.data_ptr(), oc.data_ptr()
        det = Detections(ob[:B], os_[:B], of[:B], oc[:B])
        if flame is not None:
            handle = flame._need_handle()
            if self._flame_ref is not flame:
                _lib.check(self.lib.vgh_detector_set_flame(self._det, handle))
                self._flame_ref = flame
            if slot is not None:
                if "proj" not in slot or slot["proj"].shape[1] != flame.num_vertices:
                    raise ValueError("output slot was created without (or for another) FLAME layer: new_output_slot(flame)")
                cap, himg, proj, rpy = slot["cap"], slot["head_image"], slot["proj"], slot["rpy"]
            else:
                cap = min(self.max_batch * self.keep_k, flame.max_heads)
                if self._head_out is None or self._head_out[0] != cap or self._head_out[2].shape[1] != flame.num_vertices:
                    self._head_out = (cap, torch.empty(cap, dtype=torch.int32, device=self.device), torch.empty(cap, flame.num_vertices, 3, dtype=torch.float32, device=self.device),
                                      torch.empty(cap, 3, dtype=torch.float32, device=self.device))
                cap, himg, proj, rpy = self._head_out
            if unpad is not None:
                if unpad.shape != (B, 3) or unpad.dtype != torch.float32 or not unpad.is_cuda or not unpad.is_contiguous():
                    raise ValueError("unpad must be a contiguous float32 GPU tensor [B,3] = (pad_x, pad_y, scale) per image")
                o.unpad_dev = unpad.data_ptr()
            nh = n_heads_out if n_heads_out is not None else (slot["n_heads"] if slot is not None else self.n_heads)
            o.n_heads_dev, o.head_image_dev, o.head_capacity = nh.data_ptr(), himg.data_ptr(), cap
            o.proj_dev, o.rpy_dev = proj.data_ptr(), rpy.data_ptr()
            det.n_heads, det.head_image_cap, det.vertices_cap, det.rpy_cap = nh, himg, proj, rpy
        return o, det

    def detect(self, images: torch.Tensor, confidence_threshold: float = 0.5, iou_threshold: float = 0.5, flame: Optional[FLAMELayer] = None,
               unpad: Optional[torch.Tensor] = None, use_graph: bool = False, reuse_outputs: bool = False) -> Detections:
        """net -> top-k -> NMS (every image) -> optional FLAME decode + head pose of every surviving head: ONE asynchronous
        library call (vgh_detect); the data-dependent head count stays on the device (see ``Detections``).
        ``unpad`` [B,3] = (pad_x, pad_y, scale) per image fuses detector.py:67-69.  The result is written into fresh tensors the
        caller owns; ``reuse_outputs=True`` writes into the engine's own output buffers (no allocation, see ``Detections``)."""
        B, fmt = self._check_images(images)
        cur = torch.cuda.current_stream(self.device)
        # The result tensors are allocated (and zero-filled) on the engine stream, whose pool the caching allocator returns them to when the
        # caller drops them.  Two orderings make that safe: the engine stream first waits for the caller's stream (a block freed by the caller
        # may still be read by kernels the caller queued earlier), and the tensors are recorded on the caller's stream before they are handed
        # out (the allocator then keeps a dropped block until the caller's reads of it have finished).
        self.stream.wait_stream(cur)
        # r06: the network goes to the GPU FIRST; the result tensors are allocated, zero-filled and marshalled while it runs (vgh_detect IS vgh_detector_candidates +
        # vgh_detector_select, csrc/detect.hip).  The other order left the GPU idle for the ~75 us of host work in front of a single-image call's first kernel
        # (profiles/r06_latency_trace_l1.txt: 76 us between the fill kernel and the stem).
        # (Overlap mode keeps the old order: its select runs on the side stream, ordered behind the NETWORK's event only -- a fill queued behind the network on the
        # engine stream would race with it.)
        net_first = not getattr(self, "_overlap", False)
        slot = None
        if not net_first and not reuse_outputs:
            with torch.cuda.stream(self.stream):
                slot = self.new_output_slot(flame, B)
        # (r06) lazy FLAME gather: detect() queues the select right behind the candidates, so the survivors' vectors come straight from the prediction buffers
        if use_graph and B <= self.arena_batch:
            self.forward_candidates(images, True, lazy_flame=True)
        else:
            self._set_lazy_flame(True)
            _lib.check(self.lib.vgh_detector_candidates(self._det, images.data_ptr(), fmt, B, self._sp()))
        if net_first and not reuse_outputs:
            with torch.cuda.stream(self.stream):
                slot = self.new_output_slot(flame, B)
        o, det = self._detect_out(B, flame, unpad, None, slot)
        _lib.check(self.lib.vgh_detector_select(self._det, B, float(confidence_threshold), float(iou_threshold), C.byref(o), self._sp()))
        if getattr(self, "_overlap", False):
            _lib.check(self.lib.vgh_detector_join(self._det, self._sp()))  # detect() keeps stream-ordered semantics
        cur.wait_stream(self.stream)
        if slot is not None:
            for t in slot.values():
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        return det

    def select(self, B: int, confidence_threshold: float = 0.5, iou_threshold: float = 0.5, flame: Optional[FLAMELayer] = None,
               unpad: Optional[torch.Tensor] = None, n_heads_out: Optional[torch.Tensor] = None, slot: Optional[Dict[str, torch.Tensor]] = None) -> Detections:
        """The post-candidate half of ``detect`` for the B images whose candidates are already in place
        (after ``forward_candidates`` / ``forward_net`` + ``candidates``).  In overlap mode it is queued on the detector's side
        stream: ``join()`` before reading the result.  ``n_heads_out`` [1] int32: where to write the head count."""
        o, det = self._detect_out(B, flame, unpad, n_heads_out, slot)
        _lib.check(self.lib.vgh_detector_select(self._det, B, float(confidence_threshold), float(iou_threshold), C.byref(o), self._sp()))
        return det

    def streams_in_use(self) -> List["torch.cuda.Stream"]:
        """The streams this engine queues work on in its current mode: its own, the net lanes of the batch split, the overlap-mode side
        stream (library-owned ones as ExternalStream views)."""
        out = (C.c_void_p * 4)()
        _lib.check(self.lib.vgh_detector_streams(self._det, self._sp(), out))
        ptrs = [out[i] for i in range(max(self.nsplit - 1, 0))] + ([out[3]] if out[3] else [])
        return [self.stream] + [torch.cuda.ExternalStream(p, device=self.device) for p in ptrs]

    def acquire_stream(self) -> "torch.cuda.Stream":
        """A stream measured to run side by side with the streams this engine works on (its own, the net lanes in use, the overlap-mode
        side stream): HIP maps streams onto 4 hardware queues and two streams on one queue serialise (include/vgh.h,
        vgh_stream_acquire).  For consumers that must not stall the pipeline, e.g. the communication stream of dist.DetectionGatherer."""
        out = (C.c_void_p * 4)()
        _lib.check(self.lib.vgh_detector_streams(self._det, self._sp(), out))
        lanes, side = [out[i] for i in range(3)], out[3]
        order = [self._sp()] + lanes[: max(self.nsplit - 1, 0)] + ([side] if side else []) + lanes[max(self.nsplit - 1, 0):]
        avoid = (C.c_void_p * len(order))(*order)
        got = C.c_void_p()
        _lib.check(self.lib.vgh_stream_acquire(self.device.index or 0, avoid, len(order), C.byref(got)))
        return torch.cuda.ExternalStream(got.value, device=self.device)

    def make_event(self) -> "torch.cuda.Event":
        return torch.cuda.Event()

    def record_select_done(self, event: "torch.cuda.Event"):
        """Record ``event`` behind the post-network stages queued so far (vgh_detector_record: on the side stream in overlap mode), with no stream waiting on it:
        the host can synchronise on an earlier batch's event and then queue that batch's consumers without a device-side wait."""
        if not event.cuda_event:  # torch creates the HIP event lazily on its first record (the library re-records it where it belongs)
            event.record(self.stream)
        _lib.check(self.lib.vgh_detector_record(self._det, event.cuda_event, self._sp()))

    def join_into(self, stream: "torch.cuda.Stream"):
        """Make ``stream`` (not the engine stream) wait for the last queued select: a consumer on its own stream -- e.g. the
        communication stream of dist.DetectionGatherer -- picks the results up without stalling the next batch's network."""
        _lib.check(self.lib.vgh_detector_join(self._det, stream.cuda_stream))

    # ---------------------------------------------------------------------------------------------------
    def profile_ops(self, images: torch.Tensor, repeats: int = 1) -> List[dict]:
        """Per-op device time (HIP events on the engine stream) with the algorithmic FLOPs of each op; repeats > 1: the per-op MEDIAN over that many profiled forwards
        (one pass is one sample per op: a box's clock state moves every op of a pass by the same 5 - 10 %)."""
        B, fmt = self._check_images(images)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        passes = []
        for _ in range(max(1, repeats)):
            ms = (C.c_float * len(self.program.ops))()
            _lib.check(self.lib.vgh_net_profile(self._net, images.data_ptr(), fmt, B, self._sp(), ms))
            passes.append(list(ms))
        med = [sorted(col)[len(col) // 2] for col in zip(*passes)]
        out = []
        for op, t in zip(self.program.ops, med):
            fl = 2.0 * op["macs"] * B
            by = arch.op_algorithmic_bytes(self.program, op, B)
            out.append(dict(name=op["name"], kind=op["kind"], ms=float(t), gflop=fl / 1e9, tflops=(fl / (t * 1e-3) / 1e12) if t > 0 else 0.0, gemm=op["gemm"],
                            read_mb=by["read"] / 1e6, write_mb=by["write"] / 1e6, gbps=((by["read"] + by["write"]) / (t * 1e-3) / 1e9) if t > 0 else 0.0))
        return out

    def set_cfg(self, op_index: int, cfg: int):
        _lib.check(self.lib.vgh_net_set_cfg(self._net, op_index, cfg))
        self._graph_key = None

    def cfg_names(self) -> List[str]:
        """Tile names ``set_cfg`` indexes: the bf16 table, or the split-precision table for the fp16x3 / bf16x3 modes."""
        if self.precision in ("fp16x3", "bf16x3", "fp16"):
            return [self.lib.vgh_conv_split_cfg_name(i).decode() for i in range(self.lib.vgh_conv_split_num_cfgs())]
        return [self.lib.vgh_conv_cfg_name(i).decode() for i in range(self.lib.vgh_conv_num_cfgs())]

    def cfg_ok(self, cfg: int, op: dict) -> bool:
        """Can tile ``cfg`` of this engine's table run ``op`` (tuning tools)?"""
        ob = self.program.bufs[op["out_buf"]]
        al = all(op[k] % 8 == 0 for k in ("out_coff", "out_coff2", "out_split", "cout_store", "res_coff")) and ob["pitch"] % 8 == 0
        fast = int(ob["is_f32"] != arch.FMT_F32 and al)
        if self.precision in ("fp16x3", "bf16x3", "fp16"):
            if self.precision != "fp16" and self.cfg_names()[cfg][0] == "g":
                return False  # the fp16 ping-pong pseudo-entries of the split table run single-plane fp16 nets only (vgh_conv_split_cfg_ok answers for those); a two-plane
                # net would fall back to another tile at launch and a tuner would time the fallback under the g name
            return bool(self.lib.vgh_conv_split_cfg_ok(cfg, op["ksize"], op["stride"], op["cout_pad"], fast, op["shuffle"], op.get("grp_cout", 0)))
        if not self.lib.vgh_conv_cfg_ok(cfg, op["ksize"], op["stride"], op["cout_pad"], fast, op["shuffle"]):
            return False
        if self.cfg_names()[cfg].startswith("t") and (op.get("res_buf", -1) >= 0 or op.get("grp_cout") or op.get("act") == 2):
            return False  # streaming 1x1 tiles: plain bf16 -> bf16 convs only (the executor would fall back to another tile)
        if self.cfg_names()[cfg][0] in "ghs" and (op.get("grp_cout") or op.get("act") == 2):
            return False  # ping-pong 3x3 tiles: dense convs, ReLU / none
        return not op.get("grp_cout") or op["grp_cout"] % self.lib.vgh_conv_cfg_cout_tile(cfg) == 0

    def load_tuning(self, path: Optional[str] = None) -> int:
        """Apply a measured per-layer tile table: {gemm-shape key: cfg name}. Missing file -> heuristic choice."""
        if path is not None:
            self._tuning_path = path  # remembered: set_split() re-applies the same table for its lane-count keys
        path = getattr(self, "_tuning_path", None) or os.path.join(TUNING_DIR, "conv_cfg.json")
        if not os.path.exists(path):
            return 0
        table = json.load(open(path))
        names = {n: i for i, n in enumerate(self.cfg_names())}
        applied = 0
        for i, op in enumerate(self.program.ops):
            if op["kind"] != 1 or arch.op_touches_fp8(self.program, op):  # e4m3 links run on the g tile the library picks
                continue
            pre = "" if self.precision in ("bf16", "fp8", "int8") else self.precision + ":"
            name = tuning_lookup(table, op, self.max_batch, getattr(self, "nsplit", 1), pre)
            if name in names and self.cfg_ok(names[name], op):  # a stale entry (a tile that cannot run this op) is skipped here, not replaced -- and logged -- by the library
                self.set_cfg(i, names[name])
                applied += 1
            else:
                self.set_cfg(i, -1)  # back to the library's own choice: set_split(2) then set_split(1) must not leave the two-lane table's tile on a shape the one-lane table lacks
        return applied


def calibrate_fp8(variant: str, state_dict: Optional[Dict[str, np.ndarray]], image_size: int, images: torch.Tensor, device: Optional[int] = None, fp8_min_px: int = 40) -> Dict[str, float]:
    """{e4m3 link name: max|activation|} of the "fp8" program, measured on ``images`` with the bf16 engine (every tensor of a forward stays in the arena: single
    assignment), for ``VGHeadsEngine(precision="fp8", fp8_scales=...)`` / ``arch.build_program(..., fp8_scales=...)``.  Pass images like the ones the detector will see."""
    # chunks of at most CALIB_CHUNK images through one temporary bf16 engine, a running maximum per link (one batch of every calibration image sized the arena by
    # their count -- a few hundred photographs ran out of memory or past arena_batch)
    n = int(images.shape[0])
    if n == 0:
        raise ValueError("calibrate_fp8: no calibration image")
    chunk = min(n, CALIB_CHUNK)
    eng = VGHeadsEngine(variant, state_dict, image_size, max_batch=chunk, device=device, precision="bf16", use_tuning=False)
    try:
        links = arch.fp8_link_names(variant, image_size, fp8_min_px)
        out = {link: 0.0 for link in links}
        for at in range(0, n, chunk):
            x = images[at:at + chunk].contiguous()
            for a0 in range(0, x.shape[0], eng.arena_batch):  # (an engine whose tensors pass 2 GiB runs its batch in arena chunks: the buffers hold the last one)
                xa = x[a0:a0 + eng.arena_batch].contiguous()
                eng.forward_net(xa)
                eng.stream.synchronize()
                for link, (src, live) in links.items():
                    out[link] = max(out[link], float(eng.buffer(src, xa.shape[0])[..., :live].float().abs().max()))
        return out
    finally:
        eng.close()


CALIB_CHUNK = 8  # images per calibration forward
LATENCY_MAX_BATCH = 2  # engines built for at most this many images schedule their heads on lane streams (arch.schedule_latency)


def tuning_key(op: dict, batch: int, nsplit: int = 1, bucket: Optional[int] = None, res: Optional[bool] = None) -> str:
    m, n, k = op["gemm"]
    if bucket is None:  # b64 (r03): the 20^2 / 40^2 maps of a 32-image batch fill the chip differently from those of 64 images
        bucket = 1 if batch <= 2 else (8 if batch <= 16 else (32 if batch <= 32 else 64))
    lanes = f"x{nsplit}" if nsplit > 1 else ""  # tile choices measured with the batch split over `nsplit` lane streams
    grp = f"_g{op['grp_cout']}" if op.get("grp_cout") else ""
    # r04: a conv with a residual input has another epilogue (16 more 1-KiB loads per wave through the per-CU path): its own entry, the plain key as fallback
    rs = "_res" if (op.get("res_buf", -1) >= 0 if res is None else res) else ""
    return f"b{bucket}{lanes}_m{m}_n{n}_k{k}_ks{op['ksize']}_s{op['stride']}{grp}{rs}"


def tuning_lookup(table: Dict[str, str], op: dict, batch: int, nsplit: int = 1, prefix: str = "") -> Optional[str]:
    """Tile name of the measured table for this op: the key of this batch bucket and lane count first (for batches above 32 then the b32 entry of that
    lane count: b32 covered every large batch before the b64 bucket existed), then the same without lanes."""
    b32 = batch > 32
    keys = []
    for r in ((None, False) if op.get("res_buf", -1) >= 0 else (None,)):  # the residual entry first, then the plain one
        keys += [tuning_key(op, batch, nsplit, res=r)] + ([tuning_key(op, batch, nsplit, bucket=32, res=r)] if b32 else [])  # entries measured with this lane count first
        keys += [tuning_key(op, batch, res=r)] + ([tuning_key(op, batch, bucket=32, res=r)] if b32 else [])
    for k in keys:
        if prefix + k in table:
            return table[prefix + k]
    return None
