#!/usr/bin/env python3
"""First-contact kit for the N > 1 path (BASELINE configs[3]) that needs NO GPU: `bench.py`'s own step closure (`bench.make_step`) and
`DetectionGatherer` driven on the gloo backend by a STAND-IN engine -- no kernel runs, `select` writes synthetic detections that are a function of the
global image index only, so rank 0 can check the gathered batches against the unsharded expectation bit for bit.  What it exercises is the control flow
an 8-GPU lease would otherwise see first: the output slots (three, as in bench.py), `wait_slot_free` before a slot is rewritten, `join_into` + `submit` after every select, the
late read of the previous batch, uneven shards, and the capacity-slab vs compact-rows exchange.

    python tools/dryrun_dist.py --gpus 2 --steps 5 [--compact]        (also:  python bench.py --gpus 2 --dry-run-cpu)

This is test / bring-up infrastructure: the stand-in computes nothing and is not a CPU fallback of any kernel (the product has none)."""
import argparse
import json
import os
import socket
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from head_detector_amd.dist import DetectionGatherer, shard_batch  # noqa: E402


def shard_outputs(step, rank, world, total, keep=5, V=7):
    """What an engine would leave in its output slot for this rank's shard of global batch ``step``: a function of the GLOBAL image index only, so the
    expected gathered batch does not depend on how it was sharded."""
    lo, hi = shard_batch(total, rank, world)
    g = [torch.Generator().manual_seed(1000 * step + i) for i in range(lo, hi)]
    counts = torch.tensor([int(torch.randint(0, keep + 1, (1,), generator=gi)) for gi in g], dtype=torch.int32)
    boxes = torch.stack([torch.rand(keep, 4, generator=gi) for gi in g])
    scores = torch.stack([torch.rand(keep, generator=gi) for gi in g])
    flame = torch.stack([torch.rand(keep, 413, generator=gi) for gi in g])
    verts = torch.cat([torch.rand(int(c), V, 3, generator=gi) for c, gi in zip(counts, g)] + [torch.zeros(0, V, 3)])
    return boxes, scores, flame, counts, verts


class StandInEngine:
    """The engine surface `bench.make_step` touches; `select` writes what the real engine would leave in the output slot for this rank's shard of global
    batch `step`."""

    def __init__(self, rank, world, total, keep, V):
        self.rank, self.world, self.total, self.keep, self.V = rank, world, total, keep, V
        self.stream = None
        self.step = -1
        self.calls = []

    def forward_net(self, images, use_graph=False):
        self.step += 1
        self.calls.append("net")

    def candidates(self, B, lazy_flame=False):
        self.calls.append("cand")

    def select(self, B, confidence_threshold, iou_threshold, flame, unpad, n_heads_out, slot):
        b, sc, f, c, v = shard_outputs(self.step, self.rank, self.world, self.total, self.keep, self.V)
        nb = b.shape[0]
        slot["boxes"].fill_(7.0)  # junk beyond the shard, as an engine that owns fewer images than the slab would leave it
        slot["boxes"][:nb], slot["scores"][:nb], slot["flame"][:nb] = b, sc, f
        slot["counts"].zero_()
        slot["counts"][:nb] = c
        slot["n_heads"][0] = int(c.sum())
        slot["proj"][: v.shape[0]] = v
        n_heads_out[0] = int(c.sum())
        self.calls.append("select")
        return types.SimpleNamespace(boxes=slot["boxes"], scores=slot["scores"], flame_params=slot["flame"], counts=slot["counts"], n_heads=slot["n_heads"], vertices_cap=slot["proj"])

    def make_event(self):
        return types.SimpleNamespace(synchronize=lambda: self.calls.append("host_wait"))

    def record_select_done(self, event):
        self.calls.append("record")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def step_worker(rank, world, port, q, total, steps, compact=False):
    """One rank of the dry run: `steps` benchmark steps through bench.make_step, rank 0 collects every gathered batch (compacted to the one-shot layout)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    keep, V = 5, 7
    B = shard_batch(total, 0, world)[1]
    assert shard_batch(total, rank, world)[1] - shard_batch(total, rank, world)[0] == B, "bench.py is weak scaling: every rank owns B images"
    eng = StandInEngine(rank, world, total, keep, V)
    mk = lambda: dict(boxes=torch.zeros(B, keep, 4), scores=torch.zeros(B, keep), flame=torch.zeros(B, keep, 413), counts=torch.zeros(B, dtype=torch.int32),  # noqa: E731
                      n_heads=torch.zeros(1, dtype=torch.int32), proj=torch.zeros(B * keep, V, 3))
    NSLOTS = 3  # as bench.py: a forward waits for the exchange of three batches ago
    slots = [mk() for _ in range(NSLOTS)]
    gat = DetectionGatherer(B, keep, V, vertex_rows=B * keep, device="cpu", dst=0, compact_rows=B * keep if compact else 0, slots=NSLOTS)
    n_heads_all = torch.zeros(steps, dtype=torch.int32)
    step = bench.make_step(eng, None, None, None, 0.5, B, slots, gat, True, False, n_heads_all)
    got = []

    def collect(slot):
        out = gat.result(slot)
        if rank == 0:
            assert not gat.overflowed(out), "dry run: a rank cut survivors at its compact_rows cap"
            c = gat.compact(out)
            got.append({k: getattr(c, k).clone().numpy() for k in ("boxes", "scores", "flame_params", "counts", "vertices_3d")})

    t0 = time.perf_counter()
    LAG = 2  # bench.make_step hands batch k's exchange over at step k + 2 (after a host wait on its select): batch k can be collected from step k + 2 on
    for i in range(steps):
        step(i)
        if i >= LAG:
            collect((i - LAG) % NSLOTS)
    step.flush()
    for j in range(max(steps - LAG, 0), steps):
        collect(j % NSLOTS)
    dt = time.perf_counter() - t0
    per_step = ["net", "cand", "select", "record"]
    assert [c for c in eng.calls if c != "host_wait"] == per_step * steps and eng.calls.count("host_wait") == steps
    assert eng.calls[: 4 * LAG] == per_step * LAG and eng.calls[4 * LAG : 4 * LAG + 5] == per_step + ["host_wait"], "the first exchange is handed over at step LAG, behind a host wait"
    if rank == 0:
        q.put((got, n_heads_all.tolist(), dt))
    dist.barrier()
    dist.destroy_process_group()


def check(got, n_heads, world, total, compact=False):
    """Rank 0's gathered batches against the unsharded expectation (live rows only in the compact form: rows beyond an image's count are not sent)."""
    for s, out in enumerate(got):
        exp = shard_outputs(s, 0, 1, total)  # the unsharded global batch
        cnt = exp[3]
        assert torch.equal(torch.from_numpy(out["counts"]), cnt), (s, "counts")
        for k, e in zip(("boxes", "scores", "flame_params"), exp[:3]):
            g = torch.from_numpy(out[k])
            if compact:
                for i, c in enumerate(cnt.tolist()):
                    assert torch.equal(g[i, :c], e[i, :c]) and float(g[i, c:].abs().sum()) == 0.0, (s, k, i)
            else:
                assert torch.equal(g, e), (s, k)
        assert torch.equal(torch.from_numpy(out["vertices_3d"]), exp[4]), (s, "vertices")
        assert n_heads[s] == int(shard_outputs(s, 0, world, total)[3].sum())


def run(world: int, total: int, steps: int, compact: bool = False, timeout: float = 180.0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=step_worker, args=(r, world, port, q, total, steps, compact)) for r in range(world)]
    for p in procs:
        p.start()
    got, n_heads, dt = q.get(timeout=timeout)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == steps
    check(got, n_heads, world, total, compact)
    return dt


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2, help="ranks (processes on the gloo backend; no GPU is touched)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=3, help="images per rank")
    ap.add_argument("--compact", action="store_true", help="DetectionGatherer(compact_rows=...): packed survivor rows instead of the [B, keep, 418] capacity slab")
    args = ap.parse_args(argv)
    dt = run(args.gpus, args.gpus * args.batch, args.steps, args.compact)
    keep = 5
    slab = args.batch * keep * 418 * 4
    print(json.dumps({"dry_run": "bench.make_step + DetectionGatherer on gloo with a stand-in engine (no GPU, no kernels; control flow only)", "ranks": args.gpus, "steps": args.steps,
                      "images_per_rank": args.batch, "exchange": "compact rows" if args.compact else "capacity slab", "bytes_per_rank_per_step": slab,
                      "gathered_batches_equal_unsharded_expectation": True, "wall_s": round(dt, 3)}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
