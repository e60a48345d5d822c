#!/usr/bin/env python3
"""The 8-bit link modes ("fp8": e4m3, "int8") side by side on one GPU (r05; profiles/r05_int8_links.txt):

    python tools/q8_probe.py snr      per link tensor of the bf16 network: quantisation SNR of e4m3 / int8 codes with the engine's per-tensor scale (and a per-channel one)
    python tools/q8_probe.py weights  (no GPU) per int8-input conv: SNR of the per-cout int8 weight image with and without the diagonal bypass, and of the e4m3 image
    python tools/q8_probe.py speed    network time of L b64 in bf16 / fp8 / int8 without / with the diagonal bypass (two lanes, alternating)
    python tools/q8_probe.py ops      single-stream time of every linked op of L b64 in bf16 / fp8 / int8 (which tile family pays what)
    python tools/q8_probe.py dev      deviation from the fp32 oracle (tests/test_gpu_split.py::network_vs_oracle) of the same modes, M b2 and L b1

`dev` imports the oracle through the test helper: this is a measurement script of the test infrastructure, not a product path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from head_detector_amd import _lib, arch  # noqa: E402


def snr_db(p, n):
    return 10.0 * float(np.log10(p / max(n, 1e-30)))


def cmd_snr():
    from head_detector_amd.engine import VGHeadsEngine

    for variant in ("vgg_heads_m", "vgg_heads_l"):
        B = 4
        x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
        eng = VGHeadsEngine(variant, image_size=640, max_batch=B, seed=1, precision="bf16", use_tuning=False)
        eng.forward_net(x)
        eng.stream.synchronize()
        tot = {k: [0.0, 0.0] for k in ("int8 per tensor", "int8 per channel", "e4m3 per tensor", "e4m3 per channel")}
        for link, (src, live) in arch.fp8_link_names(variant, 640, 40).items():
            t = eng.buffer(src, B)[..., :live].float()
            amax, cmax = t.abs().max(), t.abs().flatten(0, 2).max(0).values.clamp(min=1e-6)
            q_i8 = lambda s: torch.round(t / s).clamp(-127, 127) * s  # noqa: E731
            q_e4 = lambda s: (t / s).clamp(-448, 448).to(torch.float8_e4m3fn).float() * s  # noqa: E731
            res = {"int8 per tensor": q_i8(amax * arch.I8_HEADROOM / 127), "int8 per channel": q_i8(cmax * arch.I8_HEADROOM / 127),
                   "e4m3 per tensor": q_e4(amax * arch.FP8_HEADROOM / 448), "e4m3 per channel": q_e4(cmax * arch.FP8_HEADROOM / 448)}
            p = float((t ** 2).sum())
            line = f"{variant} {link:40s} max {float(amax):8.2f} rms {float(t.pow(2).mean().sqrt()):7.3f} | SNR dB:"
            for k, q in res.items():
                n = float(((q - t) ** 2).sum())
                tot[k][0] += p
                tot[k][1] += n
                line += f"  {k} {snr_db(p, n):5.1f}"
            print(line)
        print(variant, "all links, SNR dB:", {k: round(snr_db(*v), 1) for k, v in tot.items()})
        eng.close()


def cmd_weights():
    for variant in ("vgg_heads_m", "vgg_heads_l"):
        sd = arch.random_state_dict(variant, 21)
        P = arch.build_program(variant, sd, 640, "int8", fp8_scales={})
        w_all, _ = P.arrays()
        tot = {k: [0.0, 0.0] for k in ("int8", "int8, diagonal bypass", "e4m3", "bf16")}
        for op in P.ops:
            if not arch.op_touches_fp8(P, op) or P.bufs[op["in_buf"]]["is_f32"] != arch.FMT_I8:
                continue
            co, ci = op["cout_pad"], op["cin"]
            W = torch.from_numpy(w_all[op["w_off"] : op["w_off"] + co * 9 * ci].reshape(co, 9, ci).copy())

            def q_i8(Wx):
                mx = Wx.abs().flatten(1).max(1).values
                ws = torch.where(mx > 0, mx / 127, torch.ones_like(mx))
                return torch.round(Wx / ws[:, None, None]).clamp(-127, 127) * ws[:, None, None]

            mx = W.abs().flatten(1).max(1).values
            Wd = W.clone()
            idx = torch.arange(min(co, ci))
            Wd[idx, 4, idx] = 0
            s2 = torch.where(mx > 0, 2.0 ** torch.ceil(torch.log2(mx.clamp(min=1e-30) / 448)), torch.ones_like(mx))
            cands = {"int8": q_i8(W), "int8, diagonal bypass": q_i8(Wd) + (W - Wd), "e4m3": (W / s2[:, None, None]).to(torch.float8_e4m3fn).float() * s2[:, None, None], "bf16": W.to(torch.bfloat16).float()}
            p = float((W ** 2).sum())
            live = mx > 0
            line = f"{variant} {op['name']:48s} row max / rms {float((mx[live] / W[live].flatten(1).pow(2).mean(1).sqrt()).mean()):5.1f} | SNR dB:"
            for k, q in cands.items():
                n = float(((q - W) ** 2).sum())
                tot[k][0] += p
                tot[k][1] += n
                line += f"  {k} {snr_db(p, n):5.1f}"
            print(line)
        print(variant, "all int8-input convs, SNR dB:", {k: round(snr_db(*v), 1) for k, v in tot.items()})


def cmd_speed():
    from head_detector_amd.engine import VGHeadsEngine

    lib = _lib.load()
    B = 64
    NF = int(os.environ.get("Q8_FORWARDS", "20"))  # 20 = a quarter of a second per burst; 300 = long enough for the PPT power controller to settle
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
    for rnd in range(3):
        for prec, diag in (("bf16", 1), ("fp8", 1), ("int8", 0), ("int8", 1)):
            lib.vgh_net_set_i8_diag(diag)
            eng = VGHeadsEngine("vgg_heads_l", image_size=640, max_batch=B, seed=1, precision=prec, calib_images=x[:2])
            nd = sum(lib.vgh_net_op_has_diag(eng._net, i) for i in range(len(eng.program.ops)))
            eng.set_overlap(True)
            eng.set_split(2)
            for _ in range(5):
                eng.forward_net(x)
            eng.join()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(NF):
                eng.forward_net(x)
            eng.join()
            torch.cuda.synchronize()
            print(f"round {rnd} vgg_heads_l b64 {prec:5s} ops with the diagonal bypass {nd:2d}: {(time.perf_counter() - t) / NF * 1e3:7.3f} ms per forward (network only, two lanes, {NF} forwards back to back)", flush=True)
            eng.close()
    lib.vgh_net_set_i8_diag(1)


def cmd_ops():
    from head_detector_amd.engine import VGHeadsEngine

    lib = _lib.load()
    B = 64
    x = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
    tabs = {}
    for prec in ("bf16", "fp8", "int8"):
        eng = VGHeadsEngine("vgg_heads_l", image_size=640, max_batch=B, seed=1, precision=prec, calib_images=x[:2])
        for _ in range(3):
            eng.forward_net(x)
        eng.join()
        rows, rows2 = eng.profile_ops(x), eng.profile_ops(x)
        tabs[prec] = {r["name"]: min(r["ms"], r2["ms"]) for r, r2 in zip(rows, rows2)}
        if prec == "int8":
            P = eng.program
            diag = {op["name"]: lib.vgh_net_op_has_diag(eng._net, i) for i, op in enumerate(P.ops)}
            q8 = {op["name"]: (P.bufs[op["in_buf"]]["is_f32"], P.bufs[op["out_buf"]]["is_f32"], op["cin"], op["cout_pad"], P.bufs[op["out_buf"]]["h"]) for op in P.ops if arch.op_touches_fp8(P, op)}
        eng.close()
    tot = {k: 0.0 for k in tabs}
    print(f"{'op':52s} in out  cin cout  map diag   bf16    fp8   int8  (ms, single stream; in / out: VGH_FMT_* of the int8 program)")
    for name, (fi, fo, ci, co, h) in q8.items():
        print(f"{name:52s} {fi:2d} {fo:3d} {ci:4d} {co:4d} {h:4d} {diag[name]:4d} {tabs['bf16'][name]:7.4f} {tabs['fp8'][name]:7.4f} {tabs['int8'][name]:7.4f}")
        for k in tabs:
            tot[k] += tabs[k][name]
    print("linked ops total:", {k: round(v, 4) for k, v in tot.items()}, " whole net:", {k: round(sum(t.values()), 4) for k, t in tabs.items()})


def cmd_dev():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_split import network_vs_oracle

    from head_detector_amd.synthetic import synthetic_flame_model

    fm = synthetic_flame_model(seed=3)
    lib = _lib.load()
    keys = ("kept_iou_min", "kept_param_max_rel_err", "vertex_l2_metric_max", "vertex_l2_metric_mean", "dense_score_max_abs_err", "dense_iou_min")
    for variant, okey, B in (("vgg_heads_m", "m", 2), ("vgg_heads_l", "l", 1)):
        for prec, diag, onin in (("bf16", 1, False), ("fp16", 1, False), ("fp8", 1, False), ("int8", 0, False), ("int8", 1, False), ("int8", 1, True)):
            lib.vgh_net_set_i8_diag(diag)
            r = network_vs_oracle(variant, okey, prec, 640, B, fm, calibrate_on_inputs=onin)
            print(f"{variant} b{B} {prec:5s} diagonal bypass {diag} calibrated on {'the measured images' if onin else 'two other random images'}: ", {k: round(r[k], 5) for k in keys}, flush=True)
    lib.vgh_net_set_i8_diag(1)


if __name__ == "__main__":
    {"snr": cmd_snr, "weights": cmd_weights, "speed": cmd_speed, "ops": cmd_ops, "dev": cmd_dev}[sys.argv[1] if len(sys.argv) > 1 else "weights"]()
