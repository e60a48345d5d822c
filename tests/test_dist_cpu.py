"""CPU: the N>1 path (batch sharding + gather of detections) with world_size 2 on the gloo backend."""
import json
import os
import socket
import sys

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from head_detector_amd.dist import DetectionGatherer, gather_detections, shard_batch


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


# ---------------------------------------------------------------------------------------------------------------------
# the throughput path: pre-allocated slabs, asynchronous collectives, two slots (what bench.py's N>1 step drives)
def _shard_outputs(step, rank, world, total, keep=5, V=7):
    """What the engine would leave in its output slot for this rank's shard of global batch ``step``: a function of the GLOBAL image
    index only, so the expected gathered batch does not depend on how it was sharded."""
    lo, hi = shard_batch(total, rank, world)
    g = [torch.Generator().manual_seed(1000 * step + i) for i in range(lo, hi)]
    counts = torch.tensor([int(torch.randint(0, keep + 1, (1,), generator=gi)) for gi in g], dtype=torch.int32)
    boxes = torch.stack([torch.rand(keep, 4, generator=gi) for gi in g])
    scores = torch.stack([torch.rand(keep, generator=gi) for gi in g])
    flame = torch.stack([torch.rand(keep, 413, generator=gi) for gi in g])
    verts = torch.cat([torch.rand(int(c), V, 3, generator=gi) for c, gi in zip(counts, g)] + [torch.zeros(0, V, 3)])
    return boxes, scores, flame, counts, verts


def _gatherer_worker(rank, world, port, q, total, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keep, V = 5, 7
    B_max = shard_batch(total, 0, world)[1]  # rank 0 owns the largest shard
    g = DetectionGatherer(B_max, keep, V, vertex_rows=B_max * keep, device="cpu", dst=0)
    got = []

    def collect(slot):
        out = g.result(slot)
        if rank == 0:
            c = g.compact(out)
            got.append({k: getattr(c, k).clone().numpy() for k in ("boxes", "scores", "flame_params", "counts", "vertices_3d", "head_image")})
        else:
            assert out is None

    for s in range(steps):  # the loop of bench.py's N>1 step: slot s&1 is reused every second batch, the previous batch is read late
        slot = s & 1
        g.wait_slot_free(slot)
        b, sc, f, c, v = _shard_outputs(s, rank, world, total, keep, V)
        nb = b.shape[0]
        pad = lambda t: torch.cat([t, torch.full((B_max - nb, *t.shape[1:]), 7, dtype=t.dtype)])  # rows beyond the shard hold junk
        g.submit(slot, pad(b), pad(sc), pad(f), pad(c), None, v, None, local_images=nb)
        if s >= 1:
            collect((s - 1) & 1)
    collect((steps - 1) & 1)
    if rank == 0:
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


def _run_gatherer(world, total, steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, port, q, total, steps)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == steps
    for s, out in enumerate(got):
        exp = _shard_outputs(s, 0, 1, total)  # the unsharded batch
        for k, e in zip(("boxes", "scores", "flame_params", "counts", "vertices_3d"), exp):
            assert torch.equal(torch.from_numpy(out[k]), e), (s, k)
        assert out["head_image"].tolist() == torch.repeat_interleave(torch.arange(total), exp[3].long()).tolist()


def test_gatherer_double_buffered_world2_gloo():
    _run_gatherer(world=2, total=6, steps=5)


def test_gatherer_uneven_shards_10_images_4_ranks():
    _run_gatherer(world=4, total=10, steps=3)


def test_gatherer_single_process_and_vertex_cut():
    keep, V, B = 5, 7, 3
    g = DetectionGatherer(B, keep, V, vertex_rows=2, device="cpu")
    b, sc, f, c, v = _shard_outputs(0, 0, 1, B, keep, V)
    c[:] = torch.tensor([2, 0, 1])
    v = torch.rand(3, V, 3)
    g.submit(0, b, sc, f, c, None, v)
    out = g.result(0)
    assert int(out.n_heads_per_rank[0]) == 3 and out.vertex_slabs.shape == (1, 2, V, 3)  # the cut is visible: 3 heads, 2 rows sent
    assert torch.equal(out.vertex_slabs[0], v[:2]) and torch.equal(out.boxes, b) and torch.equal(out.counts, c)
    assert g.compact(out).head_image is None  # head list and vertex rows no longer line up


def test_compact_exchange_clamps_counts_at_the_cap_and_keeps_dead_rows_zero():
    """with compact_rows set, (1) survivors beyond the cap are cut AND the travelling counts are clamped to the rows shipped (no all-zero
    row is ever reported as a detection) with the number of cut rows visible; (2) NaN / Inf in the dead tail rows of the engine's slabs never reach the
    message (masked_fill, not 0 * NaN); (3) a rank that owns no image still goes through submit."""
    keep, B = 5, 3
    g = DetectionGatherer(B, keep, device="cpu", compact_rows=5)
    gen = torch.Generator().manual_seed(5)
    b, sc, f = torch.rand(B, keep, 4, generator=gen), torch.rand(B, keep, generator=gen), torch.rand(B, keep, 413, generator=gen)
    c = torch.tensor([3, 3, 3], dtype=torch.int32)
    b[:, 3:], sc[:, 3:], f[:, 3:] = float("nan"), float("inf"), float("nan")  # what NMS leaves beyond an image's count is undefined
    g.submit(0, b, sc, f, c)
    out = g.result(0)
    assert out.counts.tolist() == [3, 2, 0] and int(out.dropped_rows_per_rank[0]) == 4 and g.overflowed(out)
    full = g.compact(out)
    assert torch.equal(full.boxes[0, :3], b[0, :3]) and torch.equal(full.boxes[1, :2], b[1, :2]) and torch.equal(full.flame_params[1, :2], f[1, :2])
    assert torch.isfinite(out.compact_slabs).all() and float(full.boxes[1, 2:].abs().sum()) == 0.0 and float(full.boxes[2].abs().sum()) == 0.0
    # below the cap: nothing cut, rows beyond the total are exact zeros although the gathered tail slots hold NaN
    c2 = torch.tensor([1, 0, 2], dtype=torch.int32)
    g.wait_slot_free(1)
    g.submit(1, b, sc, f, c2)
    out2 = g.result(1)
    assert out2.counts.tolist() == [1, 0, 2] and int(out2.dropped_rows_per_rank[0]) == 0 and not g.overflowed(out2)
    assert torch.isfinite(out2.compact_slabs).all() and float(out2.compact_slabs[0, 3:].abs().sum()) == 0.0
    assert torch.equal(out2.compact_slabs[0, 1, :4], b[2, 0]) and torch.equal(out2.compact_slabs[0, 2, 5:], f[2, 1])
    # a rank without images
    g.wait_slot_free(0)
    g.submit(0, b, sc, f, c, local_images=0)
    out3 = g.result(0)
    assert out3.counts.tolist() == [0, 0, 0] and int(out3.images_per_rank[0]) == 0 and float(out3.compact_slabs.abs().sum()) == 0.0 and not g.overflowed(out3)


def _empty_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keep, total = 5, 2  # 2 images over 3 ranks: rank 2 owns none (shard_batch) and must still post its collectives
    lo, hi = shard_batch(total, rank, world)
    g = DetectionGatherer(1, keep, device="cpu", dst=0, compact_rows=4)
    if hi > lo:
        b, sc, f, c, _ = _shard_outputs(0, rank, world, total, keep)
    else:
        b, sc, f, c = torch.zeros(0, keep, 4), torch.zeros(0, keep), torch.zeros(0, keep, 413), torch.zeros(0, dtype=torch.int32)
    pad = lambda t: torch.cat([t, torch.full((1 - (hi - lo), *t.shape[1:]), 7, dtype=t.dtype)])
    g.submit(0, pad(b), pad(sc), pad(f), pad(c), None, None, None, local_images=hi - lo)
    out = g.result(0)
    if rank == 0:
        full = g.compact(out)
        q.put({"images": out.images_per_rank.tolist(), "counts": full.counts.numpy(), "boxes": full.boxes.numpy()})
    dist.barrier()
    dist.destroy_process_group()


def test_compact_exchange_with_a_rank_that_owns_no_image_world3_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = _shard_outputs(0, 0, 1, 2)
    cnt = torch.minimum(exp[3], torch.tensor(4, dtype=torch.int32))  # one image per rank, cap 4 rows
    assert got["images"] == [1, 1, 0] and got["counts"].tolist() == cnt.tolist()
    for i in range(2):
        assert torch.equal(torch.from_numpy(got["boxes"])[i, : int(cnt[i])], exp[0][i, : int(cnt[i])])


# ---------------------------------------------------------------------------------------------------------------------
# bench.py's own step closure (make_step) on gloo: the N>1 control flow of the benchmark -- two output slots, wait_slot_free before a
# slot is rewritten, join_into + submit after every select, the late read of the previous batch -- executed with the stand-in engine of
# tools/dryrun_dist.py (the same code `python bench.py --gpus 2 --dry-run-cpu` runs from a shell: the first-contact kit for an 8-GPU lease)
def _dryrun():
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)  # spawn'ed workers inherit sys.path and import `dryrun_dist` by name
    import dryrun_dist

    return dryrun_dist


@pytest.mark.parametrize("compact", [False, True], ids=["capacity_slab", "compact_rows"])
def test_bench_step_closure_world2_gloo(compact):
    _dryrun().run(world=2, total=6, steps=5, compact=compact)


def test_dry_run_cli_from_bench():
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "3"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ranks"] == 2 and line["gathered_batches_equal_unsharded_expectation"] is True


def test_bench_refuses_a_rank_count_that_differs_from_gpus(tmp_path):
    """`--gpus N` must never print an n_gpus line for another world size: under a launcher with WORLD_SIZE != N the script exits non-zero
    with the launch recipe (without a launcher it spawns the N ranks itself: bench._respawn_under_torchrun)."""
    import subprocess
    import sys

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and "n_gpus" not in r.stdout
