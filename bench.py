#!/usr/bin/env python3
"""Headline benchmark: images/sec at 640x640 for the VGGHeads forward path (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the whole hot path over one batch that is already resident in HBM:
    network (stem + 120/150 fused convs) -> box/score decode -> per-image top-k(1000) -> candidate gather + FLAME
    fix-up -> NMS (every image) -> compaction -> FLAME decode of every surviving head -> (N>1) RCCL gather to rank 0.
Workload at N=1 = BASELINE.json configs[2]: VGGHeads_L, bf16, batch 64 @ 640x640 with FLAME decode per detection (per GPU; weak
scaling: configs[3] = 8 x this).  configs[1] (VGGHeads_M, batch 32) is measured too and reported on stderr (and inside `config`).
Weights / FLAME constants are seeded synthetic tensors of the exact architecture (no network for the real assets).
The random-weight network's scores are arbitrary, so the NMS confidence threshold is calibrated ONCE (untimed) so that
about 3 heads per image survive; nothing is skipped inside the timed region.

The JSON line also carries
  roofline     : the conv implicit-GEMM kernel family against the dense bf16 MFMA peak, measured live with HIP events on
                 the engine's stream around the network part of every timed step;
  cpu_baseline : the oracle (torch-CPU fp32 restatement of the reference pipeline) on a bounded sample, rank 0 / N=1 only;
  config.bf16_vs_fp32 : deviation of the timed bf16 mode from the engine's fp32 parity mode on seeded inputs (untimed).
The run refuses to start with any VGH_* environment variable set (experiment knobs must not leak into a measurement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _timed(fn, iters: int):
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def cpu_baseline(variant: str, image_size: int, flame_model):
    """Oracle (kind "port": the torch-CPU fp32 restatement of the reference pipeline, oracle/) timed on the host cores, on a bounded
    sample: end to end at batch 1 / 8 / 32 (best reported as `value`), FLAME decode alone at n = 1 / 100, top-k + NMS
    alone on 1000 candidates; median and min of the iterations."""
    import torch

    from head_detector_amd import arch
    from oracle import flame_oracle as fo
    from oracle import net_oracle, postproc_oracle as po

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 32))  # torch's CPU conv stops scaling (and thrashes) far below the box's 256 hardware threads
    torch.set_num_threads(cores)
    sd = arch.random_state_dict(variant, 1)
    net = net_oracle.YoloHeadsOracle({"vgg_heads_m": "m", "vgg_heads_l": "l"}[variant])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    consts = fo.FlameConstants(flame_model, torch.float32)
    net(torch.rand(1, 3, 64, 64))  # spin up the thread pool / allocator on a tiny input (untimed)
    t_start = time.time()

    def end_to_end(x):
        b, s, f = net(x)
        conf = float(s[:, 3, 0].min())
        res = po.postprocess_batched(b, s, f, conf, 0.5)
        fo.reproject(consts, torch.cat([r[2] for r in res]))

    e2e, n_img = {}, 0
    for bs, iters in ((1, 10), (8, 2), (32, 1)):
        x = torch.rand(bs, 3, image_size, image_size, generator=torch.Generator().manual_seed(0))
        if bs == 1:
            end_to_end(x)  # warm-up at full size
        med, mn = _timed(lambda: end_to_end(x), iters)
        e2e[f"b{bs}"] = {"img_per_s_median": round(bs / med, 3), "img_per_s_best": round(bs / mn, 3), "iters": iters}
        n_img += bs * iters
    best = max(v["img_per_s_median"] for v in e2e.values())
    # FLAME decode alone (per-head mesh-decode metric)
    dec = {}
    for n in (1, 100):
        p = fo.synthetic_params(n, seed=2)
        fo.reproject(consts, p)
        med, mn = _timed(lambda: fo.reproject(consts, p), 10)
        dec[f"n{n}"] = {"us_per_head_median": round(med / n * 1e6, 1), "us_per_head_min": round(mn / n * 1e6, 1)}
    # top-k(1000) + NMS alone on one image's 8400 anchors
    bx, sc = po.synthetic_detections(1, seed=2, image_size=image_size)
    fl = fo.synthetic_params(bx.shape[1], seed=4)[None]

    def topk_nms():
        cb, cs, cf, _ = po.decoding_topk(bx, sc, fl, 1000)
        po.postprocess_batched(cb, cs, cf, 0.5, 0.5)

    topk_nms()
    med, mn = _timed(topk_nms, 10)
    return {"value": best, "unit": "images/sec", "cores": cores, "kind": "port", "cpu": _cpu_model(),
            "sample": f"{n_img} images ({variant} fp32 unfused torch-CPU net + top-k + NMS + FLAME decode at batch 1/8/32; median of the iterations, best batch reported) in {time.time() - t_start:.1f}s",
            "end_to_end": e2e, "flame_decode_alone": dec, "topk_nms_alone_1000cand": {"ms_median": round(med * 1e3, 3), "ms_min": round(mn * 1e3, 3)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--variant", default="vgg_heads_l")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--image-size", type=int, default=640)
    ap.add_argument("--heads-per-image", type=float, default=3.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the (untimed) bf16-vs-fp32 deviation report")
    ap.add_argument("--no-secondary", action="store_true", help="skip the (separately timed) VGGHeads_M batch-32 line on stderr")
    ap.add_argument("--per-layer", default=None, help="write a per-op timing table (json) to this path")
    ap.add_argument("--split", type=int, default=2, help="independent sub-batches per forward on the net's lane streams (1 = off)")
    ap.add_argument("--overlap", action="store_true", help="force the post-stage overlap on (default: on for batch >= 8)")
    ap.add_argument("--no-overlap", action="store_true", help="run NMS..FLAME decode on the network stream instead of the detector's (lowest-priority) side stream")
    ap.add_argument("--graph", action="store_true", help="replay the network through a captured hipGraph")
    ap.add_argument("--exchange", action="store_true", help="run the N>1 step (output slots + RCCL gather to rank 0 on the communication stream) even with one rank")
    ap.add_argument("--tuning", default=None, help="tile table to load instead of head_detector_amd/tuning/conv_cfg.json")
    args = ap.parse_args()
    leaked = sorted(k for k in os.environ if k.startswith("VGH_"))
    if leaked:
        sys.exit(f"bench.py: refusing to measure with experiment switches in the environment: {leaked}")

    import torch
    import torch.distributed as dist

    from head_detector_amd import _lib
    from head_detector_amd.dist import DetectionGatherer, init_from_env, steer_collective_stream
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer
    from head_detector_amd.synthetic import synthetic_flame_model

    rank, world, local = init_from_env(single_rank_group=args.exchange)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()
    S = args.image_size
    flame_model = synthetic_flame_model(seed=3)
    flame = FLAMELayer(model=flame_model, device=dev, max_heads=max(1024, max(args.batch, 32) * 100))
    nsplit = 1 if args.graph else max(1, min(4, args.split))
    # post stages of batch s under the network of batch s+1: pays from batch 8 up (measured r02: 2.89 vs 2.95 ms at B=8, 13.25 vs 13.42 at B=64), costs
    # at B=1 (2.2 vs 1.9 ms: the network itself is a chain of small latency-bound kernels there)
    overlap = (args.overlap or (args.batch >= 8 and not args.no_overlap)) and not args.graph

    steered = [False]

    def run_workload(variant: str, B: int, steps: int, warmup: int, per_layer_path=None) -> dict:
        """The timed region of the contract for one (variant, batch): W warm-up steps, barrier + synchronize, K steps, synchronize +
        barrier, max over ranks.  HIP events on the engine's stream bracket the network part of every timed step."""
        eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1)
        if args.tuning:
            eng.load_tuning(args.tuning)
        # synthetic images, seed 0 (+rank): u8 NHWC resident in HBM (what the letterbox stage hands over, detector.py:48-51)
        images = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(rank)).to(dev)
        unpad = torch.tensor([[0.0, 0.0, 1.0]], device=dev).expand(B, 3).contiguous()
        # calibrate the confidence threshold once (untimed): ~heads_per_image survivors per image
        _, scores, _ = eng.model(images)
        torch.cuda.synchronize()
        lo, hi = float(scores.min()), float(scores.max())
        conf = hi
        for _ in range(30):
            mid = 0.5 * (lo + hi)
            mean_heads = float(eng.detect(images, confidence_threshold=mid).counts.float().mean())
            conf = mid
            if mean_heads > args.heads_per_image:
                lo = mid
            else:
                hi = mid
            if abs(mean_heads - args.heads_per_image) < 0.25:
                break
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        n_heads_all = torch.zeros(max(steps, 1), dtype=torch.int32, device=dev)
        # throughput mode: NMS .. FLAME decode of batch s run on the detector's side stream underneath the network of batch s+1
        eng.set_overlap(overlap)
        eng.set_split(nsplit)
        if args.tuning:
            eng.load_tuning(args.tuning)

        # N>1: the detections of every rank go to rank 0 -- fixed-capacity slabs allocated once, two output slots, collectives queued
        # on a communication stream behind the detector's side stream: the gather of batch s runs under the network of batch s+1 and
        # nothing in the steady-state loop waits on the host (head_detector_amd/dist.py::DetectionGatherer)
        slots = gat = None
        if world > 1 or args.exchange:
            slots = [eng.new_output_slot(flame) for _ in range(2)]
            gat = DetectionGatherer(B, eng.keep_k, flame.num_vertices, vertex_rows=B * int(1.5 * args.heads_per_image + 1), device=dev, dst=0, stream=eng.acquire_stream(),
                                    always_collective=args.exchange)
            ready = [torch.cuda.Event() for _ in range(2)]
            if gat.collective and not steered[0]:
                # RCCL launches its kernels on a stream of torch's pool; on the hardware queue of the engine stream or of a lane they would
                # hold up the next batch's network (a queue is in-order).  Steer the pool before the first collective, then measure.
                steered[0] = True
                ok = steer_collective_stream(eng.streams_in_use())
                print(f"[bench] rank {rank}: collective stream {'clear of' if ok else 'SHARES a hardware queue with'} the engine's streams", file=sys.stderr)
        nstep = [0]

        def step(i=None):
            s = nstep[0] & 1
            nstep[0] += 1
            if gat is not None:
                gat.wait_slot_free(s, eng.stream)  # the exchange that last read this output slot (two batches ago) is over
            if i is not None:
                ev0[i].record(eng.stream)  # HIP events on the stream the kernels are launched on
            eng.forward_net(images, use_graph=args.graph)
            if i is not None:
                ev1[i].record(eng.stream)
            # post-network stages: decode/top-k/gather, then ONE library call for NMS + compaction + head list + FLAME decode of every
            # survivor (vgh_detector_select); the head count stays on the device, so the host queues ahead of the GPU
            eng.candidates(B)
            k = i if i is not None else 0
            det = eng.select(B, confidence_threshold=conf, iou_threshold=0.5, flame=flame, unpad=unpad, n_heads_out=n_heads_all[k : k + 1],
                             slot=slots[s] if slots else None)
            if gat is not None:
                if overlap:
                    eng.join_into(gat.stream)  # the communication stream (not the engine stream) waits for this batch's select
                    ev = None
                else:
                    ev = ready[s]
                    ev.record(eng.stream)
                gat.submit(s, det.boxes, det.scores, det.flame_params, det.counts, det.n_heads, det.vertices_cap, ev)

        for _ in range(warmup):
            step()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        eng.join()
        if gat is not None:
            for s in range(2):
                gat.result(s)  # the last two exchanges
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        net_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / max(steps, 1)
        heads = int(n_heads_all.sum().item())
        if per_layer_path and rank == 0:
            eng.join()
            eng.set_split(1)  # per-op events make sense on one stream only: the table is the single-stream view of every op
            json.dump(eng.profile_ops(images), open(per_layer_path, "w"), indent=0)
        out = dict(variant=variant, B=B, steps=steps, warmup=warmup, dt=dt, net_ms=net_ms, heads_per_img=heads / max(steps * B, 1),
                   value=B * world * steps / dt, flops_per_image=eng.flops_per_image, conv_tflops=eng.flops_per_image * B / (net_ms * 1e-3) / 1e12)
        eng.close()
        return out

    main_run = run_workload(args.variant, args.batch, args.steps, args.warmup, args.per_layer)
    B = args.batch

    # FLAME decode alone (second headline metric): us per head at n = 96
    p96 = torch.randn(96, 413, device=dev)
    for _ in range(3):
        flame.decode(p96, want_vertices=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        flame.decode(p96, want_vertices=False)
    e1.record()
    torch.cuda.synchronize()
    decode_us_per_head = e0.elapsed_time(e1) * 1e3 / (20 * 96)

    if rank == 0:
        traffic = None  # HBM bytes per conv launch from the committed PMC pass (profiles/), only for the exact workload it was taken on
        for rnd in ("r02", "r01"):
            tpath = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{args.variant[-1]}{B}.json")
            if traffic is None and os.path.exists(tpath) and S == 640:
                traffic = round(json.load(open(tpath))["traffic_bytes_per_launch"])
        config = {"workload": f"{args.variant} bf16 batch {B}/GPU @ {S}x{S}, u8 NHWC input resident in HBM, ~{main_run['heads_per_img']:.2f} heads/img decoded",
                  "global_batch": B * world, "image_size": S, "parallelism": f"dp{world}", "gflop_per_image": round(main_run["flops_per_image"] / 1e9, 2),
                  "graph": bool(args.graph), "exchange_to_rank0": bool(world > 1 or args.exchange), "overlap_post": overlap, "batch_split": nsplit, "flame_decode_us_per_head_n96": round(decode_us_per_head, 3),
                  "net_ms_per_step": round(main_run["net_ms"], 3)}
        if world == 1 and not args.no_secondary and (args.variant, B) != ("vgg_heads_m", 32):
            m = run_workload("vgg_heads_m", 32, max(50, args.steps // 2), args.warmup)
            config["secondary_vgg_heads_m_b32"] = {"images_per_sec": round(m["value"], 2), "ms_per_step": round(m["dt"] / m["steps"] * 1e3, 3), "net_ms_per_step": round(m["net_ms"], 3),
                                                   "conv_tflops": round(m["conv_tflops"], 2), "roofline_frac": round(m["conv_tflops"] / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "steps": m["steps"]}
            print(f"[bench] BASELINE configs[1] vgg_heads_m bf16 batch 32 @ {S}: {m['value']:.1f} img/s, {m['dt'] / m['steps'] * 1e3:.3f} ms/step, "
                  f"net {m['net_ms']:.3f} ms = {m['conv_tflops']:.1f} TFLOP/s ({m['conv_tflops'] / MFMA_BF16_DENSE_PEAK_TFLOPS:.3f} of the bf16 MFMA peak)", file=sys.stderr)
        if world == 1 and not args.no_accuracy:
            from head_detector_amd.accuracy import bf16_vs_fp32

            config["bf16_vs_fp32"] = [bf16_vs_fp32("vgg_heads_m", S, 2, flame, split=nsplit), bf16_vs_fp32("vgg_heads_l", S, 1, flame, split=1)]
        line = {
            "metric": "images/sec at 640x640 (VGGHeads forward path: net -> top-k/NMS -> FLAME decode)",
            "value": round(main_run["value"], 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(main_run["dt"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "config": config,
            "roofline": {"bound": "mfma", "achieved": round(main_run["conv_tflops"], 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(main_run["conv_tflops"] / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "kernel": "conv_igemm_kernel<*> + conv3x3_patch_kernel<*> + conv3x3_patch3_kernel<*> (all launches of one forward: algorithmic 2*MACs / HIP-event time of the network part; traffic = PMC HBM bytes per launch, mean over the conv + stem launches of a single-lane forward)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.variant, S, flame_model)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
