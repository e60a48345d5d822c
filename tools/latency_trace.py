#!/usr/bin/env python3
"""Where does one synchronous detect() of ONE image go?  (r06; VERDICT r05 item 6: HeadDetector.__call__ is the reference's API shape, detector.py:97-102)

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d DIR -o lat -- python tools/latency_trace.py run [variant]      # 30 synchronous calls
    python tools/latency_trace.py summary DIR [OUT.txt]                                                                       # the LAST call's kernels with their gaps

`run` prints the host wall time of every call; `summary` takes the last call of the trace (the kernels after the last gap above 200 us) and lists every dispatch with
its start offset, duration and the idle gap in front of it, then the totals: kernel time, gaps, and the class split (network / post-network / FLAME)."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(variant):
    import torch

    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer
    from head_detector_amd.synthetic import synthetic_flame_model

    dev = torch.device("cuda", 0)
    flame = FLAMELayer(model=synthetic_flame_model(seed=3), device=dev, max_heads=128)
    eng = VGHeadsEngine(variant, image_size=640, max_batch=1, seed=1)
    img = torch.randint(0, 256, (1, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7)).to(dev)
    unp = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
    _, sc, _ = eng.model(img)
    conf = float(sc[0, 2, 0])
    for _ in range(20):
        eng.detect(img, confidence_threshold=conf, flame=flame, unpad=unp)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        time.sleep(0.002)  # an idle gap the summary can find between calls
        t0 = time.perf_counter()
        eng.detect(img, confidence_threshold=conf, flame=flame, unpad=unp)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("LAT", variant, "host wall per synchronous call (ms):", " ".join(f"{t:.3f}" for t in ts), "median", sorted(ts)[len(ts) // 2])
    eng.close()


def summary(d, out):
    from head_detector_amd import arch

    f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    start = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 200_000:
            start = i
    call = rows[start:]
    t0 = int(call[0]["Start_Timestamp"])
    print(f"# {f}: last synchronous detect() = {len(call)} dispatches; columns: start offset us, duration us, idle gap in front us, queue, kernel", file=out)
    ksum = gsum = 0.0
    cls = {"network": [0, 0.0], "post-network": [0, 0.0], "flame": [0, 0.0]}
    prev_end = t0
    for r in call:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = max(0, s - prev_end) / 1e3
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k[:k.find("(")] if "(" in k else k
        c = "network" if arch.is_net_kernel(r["Kernel_Name"]) else "flame" if "flame" in k else "post-network"
        cls[c][0] += 1
        cls[c][1] += (e - s) / 1e3
        ksum += (e - s) / 1e3
        gsum += gap
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap:7.2f}  q{r['Queue_Id']}  {k[-70:]}", file=out)
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print(f"# span first start -> last end {span:.1f} us; kernel time {ksum:.1f} us; idle gaps {gsum:.1f} us ({100 * gsum / span:.0f} %)", file=out)
    for c, (n, t) in cls.items():
        print(f"#   {c:13s} {n:4d} dispatches {t:8.1f} us", file=out)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "vgg_heads_l")
    else:
        summary(sys.argv[2], open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout)
