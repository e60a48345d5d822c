#!/usr/bin/env python3
"""First contact with the REAL assets, in one command (r06).  None of them exists in this image: the released weights (hub download,
head_detector/detector.py:25-30), the licensed FLAME pickle (flame.py:18-24), an exported .onnx (README.md:23,199), cv2.  Whatever is supplied is checked; every
check prints PASS / FAIL / SKIPPED with the number behind it, and the exit code is the number of FAILs.

    python tools/first_contact.py [--model vgg_heads_l] [--trcd vgg_heads_l.trcd] [--onnx vgg_heads_l.onnx] [--pkl generic_model.pkl] [--images a.jpg b.jpg ...] [--cv2]

  1. --trcd / --onnx : load_weights -> weight_manifest_diff (every missing / unexpected / mis-shaped tensor; an .onnx also says whether it was read by name or by graph
                       position and which Conv nodes stayed unbound) -> build_program in every precision.
  2. --pkl           : FLAMELayer from the pickle; the reference's known-answer fixture (yolo_head_training/tests/1.json, committed as tests/golden/fixture_1json.npz):
                       its 413 parameters through vgh_flame_decode against its vertices_3d (the one full known answer the reference holds; 1e-4).
  3. --cv2           : the letterbox kernel (csrc/letterbox.hip, OpenCV's 8-bit LANCZOS4 restated) against cv2.resize + copyMakeBorder on seeded images and on --images,
                       bit for bit (detector.py:40-52).
  4. --images + weights (+ --pkl): bf16 / fp16 / int8 / fp8 against the fp16x3 parity mode on those photographs -- dense boxes IoU, scores, the kept detections'
                       parameters and vertices; int8 / fp8 calibrated on the same photographs.
GPU needed for 2 - 4 (the library has no CPU path)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

FAILS = []


def report(name, ok, detail=""):
    print(f"[{'PASS' if ok else 'FAIL'}] {name}: {detail}", flush=True)
    if not ok:
        FAILS.append(name)


def skipped(name, why):
    print(f"[SKIPPED] {name}: {why}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vgg_heads_l")
    ap.add_argument("--trcd")
    ap.add_argument("--onnx")
    ap.add_argument("--pkl")
    ap.add_argument("--images", nargs="*", default=[])
    ap.add_argument("--cv2", action="store_true")
    ap.add_argument("--image-size", type=int, default=640)
    args = ap.parse_args()
    from head_detector_amd import arch
    from head_detector_amd import detector as det

    # ---- 1. archives ----
    sd = None
    for label, path in (("trcd", args.trcd), ("onnx", args.onnx)):
        if not path:
            skipped(f"{label} ingest", "no file given")
            continue
        try:
            got = det.load_weights(path, args.model)
        except Exception as e:  # noqa: BLE001
            report(f"{label} ingest", False, f"{type(e).__name__}: {e}")
            continue
        diff = det.weight_manifest_diff(args.model, got)
        how = dict(det.LAST_LOAD_REPORT) if label == "onnx" else {}
        report(f"{label} manifest", not any(diff.values()), f"{len(got)} tensors; missing {diff['missing'][:6]} unexpected {diff['unexpected'][:6]} shape {diff['shape'][:4]}"
               + (f"; read {how.get('how')}, unbound Conv nodes {how.get('conv_nodes_unbound', how.get('not_weights', []))[:4]}" if how else ""))
        if not any(diff.values()):
            for prec in ("bf16", "fp16", "fp16x3", "fp32"):
                try:
                    arch.build_program(args.model, got, args.image_size, prec)
                    report(f"{label} lowers ({prec})", True, "op program built")
                except Exception as e:  # noqa: BLE001
                    report(f"{label} lowers ({prec})", False, f"{type(e).__name__}: {e}")
            sd = sd or got
    import torch

    gpu = torch.cuda.is_available()
    # ---- 2. FLAME pickle + the reference's known answer ----
    flame_model = None
    if args.pkl and gpu:
        from head_detector_amd.flame import FLAMELayer

        dev = torch.device("cuda", 0)
        try:
            fl = FLAMELayer(flame_path=args.pkl, device=dev, max_heads=256)
            flame_model = fl
            g = np.load(os.path.join(ROOT, "tests", "golden", "fixture_1json.npz"))
            p = torch.from_numpy(g["params"]).float()[None].to(dev)
            out = fl.decode(p)  # vertices in model space after pose, rotation and scale (flame.py:122-177)
            v3 = torch.as_tensor(out[0] if isinstance(out, (tuple, list)) else out).float().cpu().numpy().reshape(-1, 3)
            ref = g["vertices_3d"].astype(np.float64)
            err = float(np.abs(v3[: ref.shape[0]] - ref).max()) if v3.shape[0] >= ref.shape[0] else float("inf")
            report("1.json known answer through vgh_flame_decode", err < 1e-4, f"max |vertex - fixture| = {err:.3e} m over {ref.shape[0]} vertices (bar 1e-4)")
        except Exception as e:  # noqa: BLE001
            report("FLAME pickle", False, f"{type(e).__name__}: {e}")
    else:
        skipped("1.json known answer", "needs --pkl and a GPU")
    # ---- 3. letterbox against cv2 ----
    if args.cv2 and gpu:
        try:
            import cv2
        except ImportError:
            cv2 = None
            report("letterbox vs cv2", False, "cv2 is not importable here")
        if cv2 is not None:
            from head_detector_amd.letterbox import letterbox

            dev = torch.device("cuda", 0)
            rng = np.random.default_rng(0)
            imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((480, 640), (1080, 1920), (333, 517), (640, 640), (97, 1201))]
            for pth in args.images:
                im = cv2.imread(pth)
                if im is not None:
                    imgs.append(cv2.cvtColor(im, cv2.COLOR_BGR2RGB))
            worst, nbad = 0, 0
            S = args.image_size
            for im in imgs:
                got = letterbox(im, S, dev)[0].cpu().numpy()[0]
                h, w = im.shape[:2]
                sc = S / max(h, w)
                if sc != 1.0:  # detector.py:40-52
                    rs = cv2.resize(im, (int(round(w * sc)), int(round(h * sc))), interpolation=cv2.INTER_LANCZOS4)
                else:
                    rs = im
                ph, pw = S - rs.shape[0], S - rs.shape[1]
                ref = cv2.copyMakeBorder(rs, 0, ph, 0, pw, cv2.BORDER_CONSTANT, value=127)
                d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
                worst = max(worst, int(d.max()))
                nbad += int((d > 0).sum())
            report("letterbox vs cv2 (bit for bit)", worst == 0, f"{len(imgs)} images, max |diff| {worst}, {nbad} differing bytes")
    else:
        skipped("letterbox vs cv2", "needs --cv2 and a GPU")
    # ---- 4. throughput modes against the parity mode on real photographs ----
    if gpu and sd is not None and args.images:
        from head_detector_amd.engine import VGHeadsEngine
        from head_detector_amd.letterbox import letterbox

        dev = torch.device("cuda", 0)
        S = args.image_size
        ims = []
        for pth in args.images:
            from PIL import Image

            ims.append(letterbox(np.array(Image.open(pth).convert("RGB")), S, dev)[0])
        x = torch.cat(ims).contiguous()
        B = x.shape[0]
        ref_eng = VGHeadsEngine(args.model, state_dict=sd, image_size=S, max_batch=B, precision="fp16x3")
        rb, rs_, rf = [t.clone() for t in ref_eng.model(x)]
        dense_b, dense_s = ref_eng.boxes_all[:B].clone(), ref_eng.scores_all[:B].clone()
        ref_eng.close()

        def iou(a, b):
            lt, rb2 = torch.maximum(a[..., :2], b[..., :2]), torch.minimum(a[..., 2:], b[..., 2:])
            wh = (rb2 - lt).clamp(min=0)
            inter = wh[..., 0] * wh[..., 1]
            return inter / ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter)

        for prec in ("bf16", "fp16", "int8", "fp8"):
            eng = VGHeadsEngine(args.model, state_dict=sd, image_size=S, max_batch=B, precision=prec, calib_images=x if prec in ("int8", "fp8") else None)
            eng.model(x)
            conf = dense_s > 0.5  # the anchors the parity mode would keep at the reference's default threshold
            i_min = float(iou(eng.boxes_all[:B][conf], dense_b[conf]).min()) if bool(conf.any()) else float("nan")
            s_err = float((eng.scores_all[:B] - dense_s).abs().max())
            eng.close()
            print(f"[INFO] {prec} vs fp16x3 on {B} photographs: IoU min over the {int(conf.sum())} anchors with score > 0.5 = {i_min:.5f}; max |score diff| {s_err:.2e}", flush=True)
    else:
        skipped("precision modes on photographs", "needs weights, --images and a GPU")
    print(f"{len(FAILS)} FAIL(s): {FAILS}")
    return len(FAILS)


if __name__ == "__main__":
    sys.exit(main())
