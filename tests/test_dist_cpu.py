"""CPU: the N>1 path (batch sharding + gather of detections) with world_size 2 on the gloo backend."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from head_detector_amd.dist import gather_detections, shard_batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(rank, B=3, keep=5, V=11):
    g = torch.Generator().manual_seed(100 + rank)
    counts = torch.tensor([1 + rank, 0, 3], dtype=torch.int32)
    boxes = torch.rand(B, keep, 4, generator=g)
    scores = torch.rand(B, keep, generator=g)
    flame = torch.rand(B, keep, 413, generator=g)
    verts = torch.rand(int(counts.sum()), V, 3, generator=g)
    return boxes, scores, flame, counts, verts


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = gather_detections(*_make(rank), dst=0)
    if rank == 0:
        # by value (numpy): a torch tensor travels through the queue as a file descriptor served by THIS process, which may have
        # exited by the time the parent asks for it
        q.put({k: getattr(out, k).numpy() for k in ("boxes", "scores", "flame_params", "counts", "vertices_3d", "head_image")})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_detections_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {k: torch.from_numpy(v) for k, v in q.get(timeout=120).items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = [_make(r) for r in range(2)]
    assert torch.equal(got["boxes"], torch.cat([p[0] for p in parts]))
    assert torch.equal(got["scores"], torch.cat([p[1] for p in parts]))
    assert torch.equal(got["flame_params"], torch.cat([p[2] for p in parts]))
    assert torch.equal(got["counts"], torch.cat([p[3] for p in parts]))
    assert torch.equal(got["vertices_3d"], torch.cat([p[4] for p in parts]))
    assert got["head_image"].tolist() == [0, 2, 2, 2, 3, 3, 5, 5, 5]


def test_single_process_passthrough_and_sharding():
    b, s, f, c, v = _make(0)
    out = gather_detections(b, s, f, c, v)
    assert torch.equal(out.boxes, b) and out.head_image.tolist() == [0, 2, 2, 2]
    for total, world in ((512, 8), (10, 4), (3, 8), (64, 1)):
        spans = [shard_batch(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b_ - a_ for a_, b_ in spans]
        assert max(sizes) - min(sizes) <= 1
