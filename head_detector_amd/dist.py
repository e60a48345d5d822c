"""Data-parallel inference across the GPUs of one node: one process per GPU, contiguous batch shards, weights and
FLAME constants replicated, no collective on the data path except the final gather of detections to the caller's
rank.  The reference has no multi-GPU inference at all (its only collective is a 4-byte training
all-reduce, yolo_head_loss.py:463-465); this is the MI355X-native addition north_star asks for.

xGMI is a point-to-point full mesh, so the right shape is a direct gather (7 peers -> root on 7 independent links),
not a ring: counts first (tiny all_gather), then fixed-capacity slabs with dist.gather (RCCL ncclSend/ncclRecv group),
then -- only if requested -- the variable-length vertex payload padded to the global max head count."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default process group if needed."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


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
