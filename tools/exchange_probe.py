#!/usr/bin/env python3
"""What the N>1 exchange costs a rank, piece by piece, on ONE GPU (single-rank RCCL group): bench.make_step with
   (0) no gatherer; the EAGER step of r02 - r04 with (1) local copies instead of collectives, (2) collectives without the vertex slab, (3) the full exchange, (4) the compact
exchange, (5 - 7) parts of it switched off; (8, 9) the LAZY hand-over of bench.make_step (host waits for select(k - 2), no device-side wait).
ms per forward (wall) and per network part (HIP events), alternating rounds.     python tools/exchange_probe.py [variant] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from head_detector_amd.dist import DetectionGatherer, init_from_env, steer_collective_stream  # noqa: E402
from head_detector_amd.engine import VGHeadsEngine  # noqa: E402
from head_detector_amd.flame import FLAMELayer  # noqa: E402
from head_detector_amd.synthetic import synthetic_flame_model  # noqa: E402


def eager_step(eng, flame, images, unpad, conf, B, slots, gat, join, n_heads_all, ev0, ev1, ready):
    """The r02 - r04 step: batch k's exchange is queued right after its select; the communication stream waits for it ON THE DEVICE (join = True: on the detector's
    low-priority side stream through vgh_detector_join; False: on an event recorded on the engine stream)."""
    n = [0]
    ns = len(slots) if slots else 2

    def step(i=None):
        s = n[0] % ns
        n[0] += 1
        if gat is not None:
            gat.wait_slot_free(s, eng.stream)
        if i is not None:
            ev0[i].record(eng.stream)
        eng.forward_net(images)
        if i is not None:
            ev1[i].record(eng.stream)
        eng.candidates(B)
        k = i if i is not None else 0
        det = eng.select(B, confidence_threshold=conf, iou_threshold=0.5, flame=flame, unpad=unpad, n_heads_out=n_heads_all[k : k + 1], slot=slots[s] if slots else None)
        if gat is not None:
            ev = None
            if join:
                eng.join_into(gat.stream)
            else:
                ev = ready[s]
                ev.record(eng.stream)
            gat.submit(s, det.boxes, det.scores, det.flame_params, det.counts, det.n_heads, det.vertices_cap, ev)

    step.flush = lambda: None
    return step


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "vgg_heads_l"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    init_from_env(single_rank_group=True)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    flame = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=B * 100)
    eng = VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1)
    images = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    unpad = torch.tensor([[0.0, 0.0, 1.0]], device=dev).expand(B, 3).contiguous()
    _, scores, _ = eng.model(images)
    conf = float(torch.sort(scores.flatten(), descending=True).values[3 * B])
    eng.set_overlap(True)
    eng.set_split(2)
    NS, K = 3, 48
    slots = [eng.new_output_slot(flame) for _ in range(NS)]
    n_heads_all = torch.zeros(K, dtype=torch.int32, device=dev)
    comm = eng.acquire_stream()
    steered = steer_collective_stream(eng.streams_in_use())
    print(f"collective stream {'clear of' if steered else 'SHARES a queue with'} the engine's streams")
    rows = B * 5

    def variant_gat(kind):
        if kind == 0:
            return None
        return DetectionGatherer(B, eng.keep_k, flame.num_vertices, vertex_rows=0 if kind == 2 else rows, device=dev, dst=0, stream=comm, always_collective=kind >= 2,
                                 compact_rows=rows if kind == 4 else 0, slots=NS)

    names = ["no gatherer", "gatherer, local copies (no collectives)", "collectives, no vertex slab", "full exchange (capacity slab + vertex slab)", "compact exchange",
             "local copies, wait_slot_free disabled", "local copies, submit = record the done event only", "local copies, no join_into (ready event instead)",
             "LAZY (bench.make_step): host waits for select(k-2), no device-side wait; local copies", "LAZY (bench.make_step), full collectives"]
    NV = len(names)
    res = {k: [] for k in range(NV)}
    for rnd in range(3):
        for kind in range(NV):
            gat = variant_gat(3 if kind == 9 else min(kind, 1) if kind >= 5 else kind)
            if kind == 5:
                gat.wait_slot_free = lambda slot, stream=None: None
            if kind == 6:
                def only_done(slot, *a, _g=gat, **k):
                    _g.slots[slot]["done"].record(_g.stream)
                    _g.slots[slot]["busy"] = True
                gat.submit = only_done
            ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
            ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
            ready = [torch.cuda.Event() for _ in range(NS)]
            if kind >= 8:  # what bench.py does since r05
                step = bench.make_step(eng, flame, images, unpad, conf, B, slots, gat, True, False, n_heads_all, ev0, ev1, ready)
            else:
                step = eager_step(eng, flame, images, unpad, conf, B, slots if gat is not None else None, gat, kind != 7, n_heads_all, ev0, ev1, ready)
            for _ in range(12):
                step()
            eng.join()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                step(i)
            step.flush()
            eng.join()
            if gat is not None:
                for s in range(NS):
                    gat.result(s)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K * 1e3
            net = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / K
            res[kind].append((dt, net))
    for kind in range(NV):
        best = min(res[kind])
        print(f"{variant} B={B} {names[kind]:48s}: {best[0]:7.3f} ms per forward (network part {best[1]:7.3f})   all: {', '.join(f'{a:.3f}' for a, _ in res[kind])}")
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
