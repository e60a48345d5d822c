#!/usr/bin/env python3
"""Engines created one after another in ONE process, in bench.py's order (r05; profiles/r05_engine_sequence.txt): per engine the pipelined step (network + post stages on the
low-priority side stream), the same again, the pairwise overlap of its streams, and -- for the 8-bit link modes -- the step with the post stages serial, with one lane, the
network alone and the single-stream per-op sum.  Found with it: an int8-link engine can come up in a state where the side stream is STARVED (every forward stalls 2.5 - 3 ms at
the prediction guard: 14.1 vs 11.5 ms; persistent for the engine's life, deterministic in the creation history, per-op times and the measured stream overlaps unchanged, gone
with set_overlap(False)); bench.py therefore measures both before its timed steps (overlap_check).  The "tiny kernel under a never-empty queue" probe it also prints (one
workgroup spinning on stream i, a tiny kernel on stream j, the chip otherwise idle) finishes in 0.15 ms on starved and healthy engines alike: the queues themselves are not
the problem, the starvation needs the chip full."""
import os, sys, time, torch
sys.path.insert(0, ".")
from head_detector_amd.engine import VGHeadsEngine
from head_detector_amd._lib import check as _lib_check
from head_detector_amd.flame import FLAMELayer
from head_detector_amd.synthetic import synthetic_flame_model
dev = torch.device("cuda", 0)
flame = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=6400)
# SEQ_CLASS: hardware-queue CLASS of every stream an engine uses.  Two streams on one hardware queue never overlap (vgh_streams_overlap = 0), so a handful of reference streams
# created up front split into classes (= the runtime's 4 normal-priority queues), and any later stream belongs to the class of the reference it does not overlap with;
# "null" = the class of the legacy default stream.  Printed per engine next to its timing: does the starved state go with one class? (r06; no profiler needed)
REFS, REF_CLASS = [], []
def _ovl(a, b):
    return int(lib_.vgh_streams_overlap(a, b))
def stream_class(ptr):
    for r, c in zip(REFS, REF_CLASS):
        if r.cuda_stream == ptr: return c
        if not _ovl(r.cuda_stream, ptr): return c
    return -1
if os.environ.get("SEQ_CLASS"):
    from head_detector_amd import _lib as _l
    lib_ = _l.load()
    for _ in range(int(os.environ.get("SEQ_CLASS"))):
        s_ = torch.cuda.Stream(device=dev)
        c_ = next((c for r, c in zip(REFS, REF_CLASS) if not _ovl(r.cuda_stream, s_.cuda_stream)), None)
        REF_CLASS.append(len(set(REF_CLASS)) if c_ is None else c_); REFS.append(s_)
    print("SEQ_CLASS reference streams -> classes", REF_CLASS, "; class of the default (null) stream:", stream_class(0), "; of torch's current stream:", stream_class(torch.cuda.current_stream().cuda_stream), flush=True)
def run(variant, B, prec, S=640, nf=40):
    eng = VGHeadsEngine(variant, image_size=S, max_batch=B, seed=1, precision=prec)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
    unpad = torch.tensor([[0.0, 0.0, 1.0]], device=dev).expand(B, 3).contiguous()
    eng.detect(x, confidence_threshold=0.5)
    eng.set_overlap(B >= 8); eng.set_split(2 if B >= 16 else 1)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(nf)]; ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(nf)]
    def step(i=None):
        if i is not None: ev0[i].record(eng.stream)
        eng.forward_net(x)
        if i is not None: ev1[i].record(eng.stream)
        eng.candidates(B)
        eng.select(B, confidence_threshold=0.6, iou_threshold=0.5, flame=flame, unpad=unpad)
    for _ in range(5): step()
    def mark():  # one spin kernel of a role-specific length per stream (main 211 us, lane 223, side 239: far from the 1-us / 150-us spins of the stream acquisition): tools/queue_trace_summary.py finds them in a rocprofv3
        if os.environ.get("SEQ_MARK"):  # --kernel-trace of this script and reads the hardware queue (Queue_Id) each role's stream sits on; once before and once after the engine's loops
            eng.join()
            sts = eng.streams_in_use()  # [main, lanes of the batch split ..., side stream when the post stages overlap]
            for i, s_ in enumerate(sts):
                us = 211 if i == 0 else 239 if (B >= 8 and i == len(sts) - 1) else 223
                torch.cuda.synchronize(); _lib_check(eng.lib.vgh_stream_spin(s_.cuda_stream, us)); torch.cuda.synchronize()
    mark()
    eng.join(); torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(nf): step(i)
    eng.join(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / nf * 1e3
    net = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / nf
    st = eng.streams_in_use()
    # starvation probe: a never-empty queue of 1-workgroup spin kernels on one engine stream (the chip is idle otherwise), ONE tiny kernel on another: when does it finish?
    def lat(busy, probe_s, n=40, cyc=200000):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(busy):
            e0.record()
            for _ in range(n): torch.cuda._sleep(cyc)
            e2.record()
        with torch.cuda.stream(probe_s):
            torch.cuda._sleep(1000)
            e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1), 3), round(e0.elapsed_time(e2), 3)
    starv = {f"{i}->{j}": lat(st[i], st[j]) for i in range(len(st)) for j in range(len(st)) if i != j} if os.environ.get("SEQ_STARV") else {}
    ov = [[int(eng.lib.vgh_streams_overlap(a.cuda_stream, b.cuda_stream)) if a is not b else 1 for b in st] for a in st]
    if hasattr(eng.lib, "vgh_stream_blocked_behind") and len(st) > 1:  # r06: is the side stream (last) held up behind the pipe of main / a lane?
        print(f"HOL {variant} b{B} @{S} {prec:6s}: side stream blocked behind [main, lanes..] = {[int(eng.lib.vgh_stream_blocked_behind(a.cuda_stream, st[-1].cuda_stream)) for a in st[:-1]]}", flush=True)
    eng.join(); torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(nf): step(i)
    eng.join(); torch.cuda.synchronize(); dt2 = (time.perf_counter() - t) / nf * 1e3
    if REFS:
        print(f"CLS {variant} b{B} @{S} {prec:6s}: queue class of [main, lanes.., side] = {[stream_class(s_.cuda_stream) for s_ in st]} (side is low priority: its class is among the low-priority queues' images)", flush=True)
    print(f"SEQ {variant} b{B} @{S} {prec:6s}: {dt:7.3f} ms per forward, network part {net:7.3f}; again {dt2:7.3f}; stream overlap matrix {ov}; tiny kernel on stream j done at / busy chain on stream i done at (ms) {starv}; arena {eng.lib.vgh_net_buffer(eng._net, 0):#x}", flush=True)
    if prec in ("int8", "fp8"):
        def timed(label):
            for _ in range(3): step()
            eng.join(); torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(nf): step(i)
            eng.join(); torch.cuda.synchronize()
            print(f"   {label}: {(time.perf_counter() - t) / nf * 1e3:7.3f} ms per forward, network part {sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / nf:7.3f}", flush=True)
        eng.join(); eng.set_overlap(False); timed("split 2, post stages NOT overlapped")
        eng.join(); eng.set_overlap(True); eng.set_split(1); timed("split 1, overlapped")
        eng.join(); eng.set_split(2); timed("split 2, overlapped (back)")
        if hasattr(eng.lib, "vgh_detector_set_side_priority"):  # experiments build (VGH_LIB_PATH=head_detector_amd/libvgh_exp.so)
            eng.join(); _lib_check(eng.lib.vgh_detector_set_side_priority(eng._det, 0)); timed("split 2, overlapped, side stream at NORMAL priority")
            eng.join(); _lib_check(eng.lib.vgh_detector_set_side_priority(eng._det, 1)); timed("split 2, overlapped, low priority again (from the park)")
        def net_only(label):
            for _ in range(3): eng.forward_net(x)
            eng.join(); torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(nf): eng.forward_net(x)
            eng.join(); torch.cuda.synchronize()
            print(f"   {label}: {(time.perf_counter() - t) / nf * 1e3:7.3f} ms per forward", flush=True)
        net_only("split 2, network only (no candidates / select)")
        eng.join(); eng.set_split(1)
        rows = eng.profile_ops(x); rows = eng.profile_ops(x)
        tab = {r["name"]: r["ms"] for r in rows}
        PROF.setdefault(prec, []).append(tab)
        print(f"   single-stream sum {sum(tab.values()):.3f} ms", flush=True)
    mark()
    eng.close()
PROF = {}
seq = [("vgg_heads_l", 64, "bf16", 640), ("vgg_heads_m", 32, "bf16", 640), ("vgg_heads_l", 16, "bf16", 1280), ("vgg_heads_l", 32, "fp16x3", 640), ("vgg_heads_l", 8, "fp32", 640),
       ("vgg_heads_l", 64, "fp8", 640), ("vgg_heads_l", 64, "fp16", 640), ("vgg_heads_l", 64, "int8", 640), ("vgg_heads_l", 64, "bf16", 640), ("vgg_heads_l", 64, "int8", 640), ("vgg_heads_l", 64, "fp8", 640), ("vgg_heads_l", 64, "int8", 640)]
NF = int(os.environ.get("SEQ_NF", "40"))
for v, B, p, S in seq:
    run(v, B, p, S, nf=min(NF, 10) if p in ("fp32",) else NF)

for prec, tabs in PROF.items():
    base = tabs[0]
    for k, t in enumerate(tabs[1:], 1):
        diffs = [(t[n] - base[n], n, base[n], t[n]) for n in base if abs(t[n] - base[n]) > 0.15 * base[n] and abs(t[n] - base[n]) > 0.005]
        print(f"PROF {prec} engine {k} vs engine 0: {len(diffs)} ops differ by > 15 %:", [(n, round(a, 4), round(b, 4)) for _, n, a, b in sorted(diffs, reverse=True)[:12]], flush=True)
