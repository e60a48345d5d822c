"""Data-parallel inference across the GPUs of one node: one process per GPU, contiguous batch shards, weights and
FLAME constants replicated, no collective on the data path except the final gather of detections to the caller's
rank.  The reference has no multi-GPU inference at all (its only collective is a 4-byte training
all-reduce, yolo_head_loss.py:463-465); this is the MI355X-native addition north_star asks for.

xGMI is a point-to-point full mesh, so the right shape is a direct gather (7 peers -> root on 7 independent links),
not a ring.  Two implementations of the same exchange:
  * ``gather_detections``  -- one-shot, synchronous convenience (sizes the vertex payload from the counts: one host sync);
  * ``DetectionGatherer``  -- the throughput path: every buffer pre-allocated once, fixed-capacity messages only (no size ever
    visits the host), collectives queued asynchronously on a communication stream, two slots so that the gather of batch s
    runs underneath the network of batch s+1 ("must be overlapped with the next batch")."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, single_rank_group: bool = False) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default process group if needed
    (``single_rank_group``: also for WORLD_SIZE = 1, so that the exchange path can be exercised on one GPU)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def collective_shares_queue_with(stream: "torch.cuda.Stream", group=None, spin_us: int = 400) -> bool:
    """Does the process group's internal stream (RCCL runs its kernels on a stream of its own, taken from torch's pool) share a
    hardware queue with ``stream``?  HIP maps streams onto 4 hardware queues and a queue is in-order: a collective that waits for its
    input would then hold up every kernel queued behind it on ``stream``.  Measured, like vgh_streams_overlap: a tiny all_reduce issued
    while ``stream`` spins for ``spin_us`` finishes at once (different queues) or only after the spin (same queue)."""
    from . import _lib

    import ctypes as C

    lib = _lib.load()
    dev = stream.device
    # the observer must neither sit on the spinning stream's queue (its event would wait behind the spin) nor come from torch's pool
    # (a draw here would shift the pool index the group's first collective is about to take)
    got = C.c_void_p()
    avoid = (C.c_void_p * 1)(stream.cuda_stream)
    _lib.check(lib.vgh_stream_acquire(dev.index or 0, avoid, 1, C.byref(got)))
    probe = torch.cuda.ExternalStream(got.value, device=dev)
    t = torch.ones(1, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    worst = 0.0
    for _ in range(2):  # the first pass also pays the communicator's lazy set-up
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(probe):
            e0.record()
            _lib.check(lib.vgh_stream_spin(stream.cuda_stream, spin_us))
            w = dist.all_reduce(t, group=group, async_op=True)
            w.wait()
            e1.record()
        torch.cuda.synchronize(dev)
        worst = e0.elapsed_time(e1) * 1e3
    lib.vgh_stream_release(dev.index or 0, got)
    return worst > 0.6 * spin_us


def steer_collective_stream(avoid: List["torch.cuda.Stream"], group=None, max_draws: int = 24) -> bool:
    """Best effort, BEFORE the group's first collective on this device: torch hands RCCL the next stream of its round-robin pool, and
    pool streams land on the hardware queues in creation order.  Draw pool streams until the one RCCL is about to get is measured to
    overlap with every stream in ``avoid`` -- by finding a good one and skipping one full period of the queue count -- then verify
    with ``collective_shares_queue_with``.  Returns True when the verification finds no shared queue."""
    from . import _lib

    lib = _lib.load()
    dev = avoid[0].device
    held = []  # keep the drawn streams alive: pool streams are never destroyed anyway

    def good(s):
        return all(lib.vgh_streams_overlap(a.cuda_stream, s.cuda_stream) == 1 for a in avoid)

    period = None
    first_good = None
    for i in range(max_draws):
        s = torch.cuda.Stream(device=dev)
        held.append(s)
        if first_good is None:
            if good(s):
                first_good = i
            continue
        # the period of the pool -> queue map: the next draw that does NOT overlap with the first good one sits on its queue
        if lib.vgh_streams_overlap(held[first_good].cuda_stream, s.cuda_stream) == 0 and good(s):
            period = i - first_good
            break
    if period:
        for _ in range(period - 1):  # RCCL's draw is the one after these: first_good + 2 * period
            held.append(torch.cuda.Stream(device=dev))
    shared = [collective_shares_queue_with(a, group) for a in avoid]
    if os.environ.get("HEAD_DETECTOR_AMD_DEBUG_STREAMS"):
        import sys

        print(f"[steer] first_good={first_good} period={period} draws={len(held)} shared_with={shared}", file=sys.stderr)
    return not any(shared)


def shard_batch(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, stop) of a global batch; the first (total % world) ranks take one extra image."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


@dataclass
class GatheredDetections:
    boxes: torch.Tensor  # [B_total, keep, 4]
    scores: torch.Tensor  # [B_total, keep]
    flame_params: torch.Tensor  # [B_total, keep, 413]
    counts: torch.Tensor  # [B_total]
    vertices_3d: Optional[torch.Tensor] = None  # [n_total, V, 3], image-major
    head_image: Optional[torch.Tensor] = None  # [n_total] global image index
    # DetectionGatherer.result only (capacity-shaped, nothing trimmed on the host):
    n_heads_per_rank: Optional[torch.Tensor] = None  # [world] int32
    vertex_slabs: Optional[torch.Tensor] = None  # [world, vertex_rows, V, 3]: rank r's first min(n_heads[r], vertex_rows) rows are live
    images_per_rank: Optional[torch.Tensor] = None  # [world] int32: rows [r*B_local, r*B_local + images[r]) of the slabs are real images
    dropped_rows_per_rank: Optional[torch.Tensor] = None  # [world] int32 (compact exchange): survivors of rank r that did not fit its compact_rows cap (0 = nothing was cut)
    compact_slabs: Optional[torch.Tensor] = None  # [world, compact_rows, 418] (DetectionGatherer(compact_rows=...)): survivors packed image-major per rank


def gather_detections(boxes: torch.Tensor, scores: torch.Tensor, flame_params: torch.Tensor, counts: torch.Tensor, vertices_3d: Optional[torch.Tensor] = None,
                      dst: int = 0, group=None) -> Optional[GatheredDetections]:
    """Every rank passes its local slabs (equal local batch on every rank); rank `dst` receives the concatenation in
    rank order, others return None.  Works on any backend (RCCL on the GPUs, gloo in the CPU tests)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        n = int(counts.sum()) if vertices_3d is not None else 0
        hi = torch.repeat_interleave(torch.arange(counts.numel(), device=counts.device), counts.long()) if vertices_3d is not None else None
        return GatheredDetections(boxes, scores, flame_params, counts, vertices_3d[:n] if vertices_3d is not None else None, hi)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B, keep = scores.shape
    # 1) counts: tiny all_gather (every rank learns the payload sizes -> no second handshake for the vertices)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts.contiguous(), group=group)
    # 2) fixed-capacity slab: [B, keep, 4 + 1 + 413] in one message per peer
    slab = torch.cat([boxes, scores.unsqueeze(-1), flame_params], dim=-1).contiguous()
    recv = [torch.empty_like(slab) for _ in range(world)] if rank == dst else None
    dist.gather(slab, recv, dst=dst, group=group)
    verts_all = None
    if vertices_3d is not None:
        # 3) variable-length payload, padded to the global max so that one gather moves it
        totals = [int(c.sum()) for c in all_counts]
        cap = max(max(totals), 1)
        V = vertices_3d.shape[1]
        pad = torch.zeros(cap, V, 3, dtype=vertices_3d.dtype, device=vertices_3d.device)
        pad[: totals[rank]] = vertices_3d[: totals[rank]]
        vrecv = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, vrecv, dst=dst, group=group)
        if rank == dst:
            verts_all = torch.cat([v[:t] for v, t in zip(vrecv, totals)], dim=0)
    if rank != dst:
        return None
    full = torch.cat(recv, dim=0)
    cnt = torch.cat(all_counts, dim=0)
    hi = torch.repeat_interleave(torch.arange(cnt.numel(), device=cnt.device), cnt.long()) if verts_all is not None else None
    return GatheredDetections(full[..., :4], full[..., 4], full[..., 5:], cnt, verts_all, hi)


class DetectionGatherer:
    """Sync-free gather of per-rank detection slabs to ``dst``, double-buffered.

    Per step and rank the message is ONE packed slab ``[B_local, keep, 418]`` (boxes 4 | score 1 | FLAME 413) + ``counts [B_local]``
    (+ ``n_heads [1]``), and -- optionally -- the first ``vertex_rows`` rows of the image-major vertex list ``[vertex_rows, V, 3]``
    (rows beyond the rank's head count are don't-care; ``n_heads`` > ``vertex_rows`` means the tail was cut and is visible to the
    consumer).  All sizes are fixed at construction, so nothing here reads a device value on the host:

        g = DetectionGatherer(B_local, keep, num_vertices, vertex_rows, device)
        for s in ...:
            slot = s & 1
            g.wait_slot_free(slot, stream)            # the engine may overwrite the slot's producer buffers again
            ... engine writes its outputs ...         # (any stream; record `ready` on it)
            g.submit(slot, boxes, scores, flame, counts, n_heads, vertices, ready_event)   # returns at once
        out = g.result(slot)                           # on dst: tensors valid once the current stream has waited (done inside)

    ``submit`` packs on the communication stream (after ``ready_event``), then queues the collectives there; ``result`` makes the
    caller's stream wait for them.  Ranks may own different numbers of images (uneven shards): pass ``B_local`` = the largest shard
    and ``local_images`` = this rank's count; rows beyond it carry count 0."""

    def __init__(self, B_local: int, keep: int, num_vertices: int = 0, vertex_rows: int = 0, device=None, dst: int = 0, group=None, slots: int = 2, stream=None, always_collective: bool = False,
                 compact_rows: int = 0):
        """``compact_rows`` > 0 (SURVEY 8(e) option (ii)): instead of the capacity slab ``[B_local, keep, 418]`` (10.7 MB per rank and step at B = 64, ~3 % of
        it live) each rank sends its survivors packed image-major into ``[compact_rows, 418]`` -- a fixed cap, so still no size on the host: row r is
        detection r - first[image] of image = searchsorted(cumsum(counts), r); rows beyond the rank's total are exact zeros (``masked_fill``, not a multiply:
        a NaN in a dead NMS tail row stays out).  A total above the cap is cut at the cap AND the per-image counts that travel are clamped to the rows
        actually shipped, so ``counts`` never marks a row live that was not sent; the number of cut rows travels too (``dropped_rows_per_rank``; ``overflowed()``
        is the check a caller asserts on).  A rank that owns no image (``local_images=0``) still posts every collective with an all-zero slab.  ``result``
        then carries ``compact_slabs [world, compact_rows, 418]`` and ``compact()`` rebuilds the per-image layout."""
        self.group, self.dst = group, dst
        self.crows = int(compact_rows)
        # always_collective: go through the process group even when it has a single rank (exercises the RCCL calls on a 1-GPU box)
        self.collective = dist.is_initialized() and (dist.get_world_size(group) > 1 or always_collective)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.B, self.keep, self.V, self.vrows = B_local, keep, num_vertices, vertex_rows
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        W = self.world if self.rank == dst else 0
        self.slots = []
        for _ in range(slots):
            sl = dict(
                send=torch.zeros(B_local, keep, 418, **f32) if not self.crows else torch.zeros(self.crows, 418, **f32),
                send_counts=torch.zeros(B_local + 3, **i32),  # [counts | n_heads | images owned | compact rows cut at the cap]
                # compact exchange: index / gather scratch allocated once (the steady-state loop allocates nothing for the payload)
                pack=(dict(flat=torch.zeros(self.crows, dtype=torch.int64, device=self.device), dead=torch.zeros(self.crows, dtype=torch.bool, device=self.device),
                           b=torch.zeros(self.crows, 4, **f32), s=torch.zeros(self.crows, **f32), f=torch.zeros(self.crows, 413, **f32)) if self.crows else None),
                send_verts=torch.zeros(vertex_rows, num_vertices, 3, **f32) if vertex_rows else None,
                recv=(torch.zeros(max(W, 1), B_local, keep, 418, **f32) if not self.crows else torch.zeros(max(W, 1), self.crows, 418, **f32)) if self.rank == dst else None,
                recv_counts=torch.zeros(self.world, B_local + 3, **i32),  # all_gather target: every rank has it
                recv_verts=torch.zeros(max(W, 1), vertex_rows, num_vertices, 3, **f32) if (self.rank == dst and vertex_rows) else None,
                work=[], done=torch.cuda.Event() if self.cuda else None, busy=False, reader=None)
            self.slots.append(sl)
        # `stream`: the communication stream; pass VGHeadsEngine.acquire_stream() so that it does not share a hardware queue with the engine
        self.stream = (stream if stream is not None else torch.cuda.Stream(device=self.device)) if self.cuda else None

    # -------------------------------------------------------------------------------------------------------------------
    def wait_slot_free(self, slot: int, stream=None):
        """Order ``stream`` (default: current) after the slot's previous exchange: its producer buffers may be reused afterwards."""
        sl = self.slots[slot]
        if not sl["busy"]:
            return
        if self.cuda:
            (stream or torch.cuda.current_stream(self.device)).wait_event(sl["done"])
        else:
            for w in sl["work"]:
                w.wait()
            sl["work"] = []

    def submit(self, slot: int, boxes: torch.Tensor, scores: torch.Tensor, flame_params: torch.Tensor, counts: torch.Tensor, n_heads: Optional[torch.Tensor] = None,
               vertices: Optional[torch.Tensor] = None, ready_event=None, local_images: Optional[int] = None):
        sl = self.slots[slot]
        nb = boxes.shape[0] if local_images is None else local_images
        ctx = torch.cuda.stream(self.stream) if self.cuda else _NullCtx()
        if self.cuda and ready_event is not None:
            self.stream.wait_event(ready_event)
        if self.cuda and sl["reader"] is not None:  # whoever took result(slot) reads the receive buffers on that stream: let it finish
            self.stream.wait_stream(sl["reader"])
            sl["reader"] = None
        with ctx:
            # pack into the pre-allocated send buffers (device-side copies; the compact form gathers fixed-shape index tensors: no size visits the host)
            if not self.crows:
                sl["send"][:nb, :, 0:4].copy_(boxes[:nb], non_blocking=True)
                sl["send"][:nb, :, 4].copy_(scores[:nb], non_blocking=True)
                sl["send"][:nb, :, 5:].copy_(flame_params[:nb], non_blocking=True)
            elif nb == 0:  # a rank without images (uneven shards): nothing to pack, but every collective below is still posted
                sl["send"].zero_()
                sl["send_counts"][: self.B].zero_()
                sl["send_counts"][self.B + 2] = 0
            else:
                pk = sl["pack"]
                c = counts[:nb].long().clamp(min=0, max=self.keep)
                ends = torch.cumsum(c, 0)
                first = ends - c
                r = torch.arange(self.crows, device=ends.device)
                img = torch.searchsorted(ends, r, right=True).clamp(min=0, max=nb - 1)  # image of compact row r
                j = (r - first[img]).clamp(min=0, max=self.keep - 1)  # its rank inside the image
                torch.add(img * int(boxes.shape[1]), j, out=pk["flat"])
                torch.ge(r, ends[nb - 1 : nb], out=pk["dead"])  # rows beyond the rank's total
                torch.index_select(boxes[:nb].reshape(-1, 4), 0, pk["flat"], out=pk["b"])
                torch.index_select(scores[:nb].reshape(-1), 0, pk["flat"], out=pk["s"])
                torch.index_select(flame_params[:nb].reshape(-1, flame_params.shape[-1]), 0, pk["flat"], out=pk["f"])
                sl["send"][:, 0:4] = pk["b"].masked_fill_(pk["dead"].unsqueeze(-1), 0.0)
                sl["send"][:, 4] = pk["s"].masked_fill_(pk["dead"], 0.0)
                sl["send"][:, 5:] = pk["f"].masked_fill_(pk["dead"].unsqueeze(-1), 0.0)
                # the counts that travel describe the rows that travel: an image whose detections straddle the cap keeps the part that fitted
                shipped = (ends.clamp(max=self.crows) - first.clamp(max=self.crows)).to(torch.int32)
                sl["send_counts"][:nb].copy_(shipped, non_blocking=True)
                sl["send_counts"][self.B + 2] = (ends[nb - 1] - self.crows).clamp(min=0).to(torch.int32)
            if not self.crows:
                sl["send_counts"][:nb].copy_(counts[:nb], non_blocking=True)
            if nb < self.B:
                sl["send_counts"][nb : self.B].zero_()
            if n_heads is not None:
                sl["send_counts"][self.B : self.B + 1].copy_(n_heads.reshape(1), non_blocking=True)
            else:
                sl["send_counts"][self.B] = sl["send_counts"][: self.B].sum()
            sl["send_counts"][self.B + 1] = nb
            if self.vrows and vertices is not None:
                r = min(self.vrows, vertices.shape[0])
                sl["send_verts"][:r].copy_(vertices[:r], non_blocking=True)
            work = []
            if self.collective:
                work.append(dist.all_gather(list(sl["recv_counts"].unbind(0)), sl["send_counts"], group=self.group, async_op=True))
                gl = list(sl["recv"].unbind(0)) if self.rank == self.dst else None
                work.append(dist.gather(sl["send"], gl, dst=self.dst, group=self.group, async_op=True))
                if self.vrows:
                    gv = list(sl["recv_verts"].unbind(0)) if self.rank == self.dst else None
                    work.append(dist.gather(sl["send_verts"], gv, dst=self.dst, group=self.group, async_op=True))
            else:
                sl["recv_counts"][0].copy_(sl["send_counts"])
                sl["recv"][0].copy_(sl["send"])
                if self.vrows:
                    sl["recv_verts"][0].copy_(sl["send_verts"])
            if self.cuda:
                for w in work:
                    w.wait()  # NCCL: orders the communication stream after the collective, does not block the host
                sl["done"].record(self.stream)
                sl["work"] = []
            else:
                sl["work"] = work
        sl["busy"] = True

    def result(self, slot: int) -> Optional[GatheredDetections]:
        """On ``dst``: the gathered batch of the slot (views of the receive buffers; the next submit of the slot is ordered after the
        work the calling stream has queued by then, so read -- or copy -- them on the stream that called ``result``).
        Per-head tensors stay capacity-shaped -- ``counts`` / ``n_heads_per_rank`` say which rows are live -- so no size is read on the
        host here either; ``compact()`` does the host-side trimming when a caller wants the one-shot layout."""
        sl = self.slots[slot]
        self.wait_slot_free(slot)
        if self.rank != self.dst:
            return None
        if self.cuda:
            sl["reader"] = torch.cuda.current_stream(self.device)
        if self.crows:
            out = GatheredDetections(None, None, None, sl["recv_counts"][:, : self.B].reshape(-1))
            out.compact_slabs = sl["recv"]  # [world, compact_rows, 418]: rank r's survivors image-major, rows beyond its total are zero
        else:
            full = sl["recv"].reshape(self.world * self.B, self.keep, 418)
            out = GatheredDetections(full[..., :4], full[..., 4], full[..., 5:], sl["recv_counts"][:, : self.B].reshape(-1))
        out.n_heads_per_rank = sl["recv_counts"][:, self.B]
        out.images_per_rank = sl["recv_counts"][:, self.B + 1]
        out.dropped_rows_per_rank = sl["recv_counts"][:, self.B + 2]
        out.vertex_slabs = sl["recv_verts"]  # [world, vertex_rows, V, 3] or None
        return out

    @staticmethod
    def overflowed(out: GatheredDetections) -> bool:
        """Did any rank cut survivors at its ``compact_rows`` cap in this exchange (reads one small device tensor: a host sync)?"""
        return out.dropped_rows_per_rank is not None and bool((out.dropped_rows_per_rank > 0).any())

    def compact(self, out: GatheredDetections) -> GatheredDetections:
        """Host-side trimming of ``result`` into the layout of ``gather_detections`` (reads the counts: one sync): the padding rows
        of short shards are dropped, so row i is global image i of a ``shard_batch`` split; vertices are image-major."""
        imgs = out.images_per_rank.tolist()
        rows = torch.cat([torch.arange(r * self.B, r * self.B + imgs[r]) for r in range(self.world)]).to(out.counts.device)
        cnt = out.counts[rows]
        if out.compact_slabs is not None:  # rebuild [image, keep, .] from the packed survivor rows (zeros where an image has fewer than keep)
            full = torch.zeros(self.world * self.B, self.keep, 418, dtype=out.compact_slabs.dtype, device=out.compact_slabs.device)
            for r in range(self.world):
                c = out.counts[r * self.B : (r + 1) * self.B].long().clamp(min=0, max=self.keep)
                ends = torch.cumsum(c, 0)
                n = min(int(ends[-1]), self.crows)
                k = torch.arange(n, device=c.device)
                img = torch.searchsorted(ends, k, right=True)
                full[r * self.B + img, k - (ends - c)[img]] = out.compact_slabs[r, :n]
            out = GatheredDetections(full[..., :4], full[..., 4], full[..., 5:], out.counts, n_heads_per_rank=out.n_heads_per_rank, vertex_slabs=out.vertex_slabs,
                                     images_per_rank=out.images_per_rank, dropped_rows_per_rank=out.dropped_rows_per_rank)
        verts = hi = None
        if out.vertex_slabs is not None:
            nh = [min(int(n), self.vrows) for n in out.n_heads_per_rank.tolist()]
            verts = torch.cat([out.vertex_slabs[r, : nh[r]] for r in range(self.world)], dim=0)
            hi = torch.repeat_interleave(torch.arange(cnt.numel(), device=cnt.device), cnt.long())[: verts.shape[0]] if sum(nh) == int(cnt.sum()) else None
        return GatheredDetections(out.boxes[rows], out.scores[rows], out.flame_params[rows], cnt, verts, hi)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
