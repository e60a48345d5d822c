"""`.vghpack`: everything `vgh_create` (include/vgh.h) needs to build the pipeline without Python -- the lowered op program, the
folded fp32 weights, the per-op tile choices (by name) and the FLAME constants behind a versioned header.

    python -m head_detector_amd.pack <variant> <weights.trcd | seed:N> <generic_model.pkl | seed:N | none> out.vghpack
                                     [--image-size 640] [--batch 64] [--split 2] [--precision bf16]

`weights.trcd` is the released TorchScript archive HeadDetector downloads (head_detector/detector.py:25-30); `seed:N` packs the
seeded synthetic weights of the exact architecture (what the benchmarks use: the released assets are not in this image).
`--batch/--split` select the tile-table bucket (head_detector_amd/tuning/conv_cfg.json) the per-op choices are resolved for.
Pure host code: packing needs neither a GPU nor libvgh.so.

Layout (little-endian):  header (128 B: magic "VGHPACK\\0", version 3, header_bytes, variant[32], image_size, precision,
n_bufs, n_ops, n_levels, shape_c, expr_c, has_flame, tune_batch, reserved, flops_per_image f64, n_weights i64, n_biases i64,
V, NB, NJ, F) | vgh_buf_desc[n_bufs] | vgh_op_desc[n_ops] | char tile_name[n_ops][32] | level[n_levels]{buf,h,w,pitch,stride} |
f32 weights | f32 biases | FLAME: v_template[V,3] shapedirs[V,3,NB] posedirs[(NJ-1)*9,3V] J_regressor[NJ,V] parents[NJ] i32
lbs_weights[V,NJ] faces[F,3] i32.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import struct
import sys
from typing import Any, Dict, Optional

import numpy as np

from . import _lib, arch

MAGIC = b"VGHPACK\0"
VERSION = 3  # 2: vgh_op_desc grew grp_cout / grp_in_stride; precision = VGH_FMT_* of the activation buffers; 3 (r05): vgh_buf_desc grew `scale` (VGH_FMT_FP8 links)
HEADER_BYTES = 128
_HDR = "<8sII32s8i2idqq4i"
assert struct.calcsize(_HDR) == HEADER_BYTES


def _flame_arrays(model: Dict[str, Any]):
    """The buffers FLAMELayer.__init__ registers (head_detector/flame.py:75-95), as vgh_flame_create wants them."""
    from .flame import _to_np

    v_template = _to_np(model["v_template"]).astype(np.float32)
    shapedirs = _to_np(model["shapedirs"]).astype(np.float32)
    pd = _to_np(model["posedirs"])
    posedirs = np.reshape(pd, [-1, pd.shape[-1]]).T.astype(np.float32).copy()
    jreg = _to_np(model["J_regressor"]).astype(np.float32)
    parents = _to_np(model["kintree_table"], np.int64)[0].astype(np.int32).copy()
    parents[0] = -1
    weights = _to_np(model["weights"]).astype(np.float32)
    faces = _to_np(model["f"], np.int64).astype(np.int32) if "f" in model else np.zeros((0, 3), np.int32)
    return v_template, shapedirs, posedirs, jreg, parents, weights, faces


def write_pack(path: str, program: "arch.Program", flame_model: Optional[Dict[str, Any]] = None, tile_names: Optional[Dict[int, str]] = None, tune_batch: int = 0) -> int:
    P = program
    w, b = P.arrays()
    fields = [f for f, _ in _lib.OpDesc._fields_]
    ops = (_lib.OpDesc * len(P.ops))(*[_lib.OpDesc(**{f: (op.get(f, 0) if f != "in_buf" else max(op[f], 0)) for f in fields}) for op in P.ops])
    bufs = (_lib.BufDesc * len(P.bufs))(*[_lib.BufDesc(bf["h"], bf["w"], bf["pitch"], bf["is_f32"], float(bf.get("scale", 0.0))) for bf in P.bufs])
    names = np.zeros((len(P.ops), 32), dtype=np.uint8)
    for i, nm in (tile_names or {}).items():
        raw = nm.encode()[:31]
        names[i, : len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    levels = np.array([[lv["buf"], lv["h"], lv["w"], lv["pitch"], lv["stride"]] for lv in P.levels], dtype=np.int32)
    V = NB = NJ = F = 0
    fl = None
    if flame_model is not None:
        fl = _flame_arrays(flame_model)
        V, NB, NJ, F = fl[0].shape[0], fl[1].shape[2], fl[3].shape[0], fl[6].shape[0]
    hdr = struct.pack(_HDR, MAGIC, VERSION, HEADER_BYTES, P.variant.encode()[:31], P.image_size, arch.PRECISION_FMT[P.precision], len(P.bufs), len(P.ops), len(P.levels),
                      P.shape_c, P.expr_c, int(fl is not None), tune_batch, 0, float(P.flops), int(w.size), int(b.size), V, NB, NJ, F)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(bytes(bufs))
        f.write(bytes(ops))
        f.write(names.tobytes())
        f.write(levels.tobytes())
        f.write(np.ascontiguousarray(w, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(b, dtype=np.float32).tobytes())
        if fl is not None:
            for a in fl:
                f.write(np.ascontiguousarray(a).tobytes())
    return os.path.getsize(path)


def read_header(path: str) -> Dict[str, Any]:
    with open(path, "rb") as f:
        raw = f.read(HEADER_BYTES)
    if len(raw) != HEADER_BYTES or raw[:8] != MAGIC:
        raise ValueError(f"{path} is not a .vghpack file")
    v = struct.unpack(_HDR, raw)
    keys = ("magic", "version", "header_bytes", "variant", "image_size", "precision", "n_bufs", "n_ops", "n_levels", "shape_c", "expr_c", "has_flame", "tune_batch", "reserved",
            "flops_per_image", "n_weights", "n_biases", "V", "NB", "NJ", "F")
    h = dict(zip(keys, v))
    h["variant"] = h["variant"].split(b"\0")[0].decode()
    if h["version"] != VERSION:
        raise ValueError(f"{path}: pack version {h['version']}, this package reads version {VERSION}")
    return h


def tile_names_for(program: "arch.Program", batch: int, nsplit: int = 1, table_path: Optional[str] = None) -> Dict[int, str]:
    """Per-op tile choices of the measured table for this batch bucket / lane count (what VGHeadsEngine.load_tuning applies)."""
    from .engine import TUNING_DIR, tuning_lookup

    path = table_path or os.path.join(TUNING_DIR, "conv_cfg.json")
    if not os.path.exists(path):
        return {}
    table = json.load(open(path))
    pre = "" if program.precision in ("bf16", "fp8", "int8") else program.precision + ":"  # the split-precision modes have their own keys and tile names
    out = {}
    for i, op in enumerate(program.ops):
        if op["kind"] != 1 or arch.op_touches_fp8(program, op):  # e4m3 links run on the g tile the library picks
            continue
        name = tuning_lookup(table, op, batch, nsplit, pre)
        if name:
            out[i] = name
    return out


def _weights_arg(spec: str, variant: str) -> Dict[str, np.ndarray]:
    if spec.startswith("seed:"):
        return arch.random_state_dict(variant, int(spec[5:]))
    from .detector import load_weights

    return load_weights(spec, variant)


def _flame_arg(spec: str) -> Optional[Dict[str, Any]]:
    if spec == "none":
        return None
    if spec.startswith("seed:"):
        from .synthetic import synthetic_flame_model

        return synthetic_flame_model(seed=int(spec[5:]))
    from .flame import get_flame_model

    return get_flame_model(spec)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m head_detector_amd.pack", description=__doc__.split("\n\n")[0])
    ap.add_argument("variant", choices=sorted(arch.VARIANTS))
    ap.add_argument("weights", help="released .trcd archive, or seed:N for seeded synthetic weights")
    ap.add_argument("flame", help="FLAME generic_model.pkl, seed:N for the synthetic model, or none")
    ap.add_argument("out")
    ap.add_argument("--image-size", type=int, default=640)
    ap.add_argument("--batch", type=int, default=64, help="batch the per-op tile choices are resolved for")
    ap.add_argument("--split", type=int, default=1, help="lane count the tile choices are resolved for")
    ap.add_argument("--precision", default="bf16", choices=sorted(arch.PRECISION_FMT))
    ap.add_argument("--fp8-scales", default=None, help="--precision fp8: json {e4m3 link name: max|activation|} from head_detector_amd.engine.calibrate_fp8")
    args = ap.parse_args(argv)
    scales = None
    if args.precision in arch.Q8_PRECISIONS:  # the e4m3 links need calibrated activation maxima (engine.calibrate_fp8 on a GPU box, saved as {link name: max|x|}); packing itself needs no GPU
        if not args.fp8_scales:
            sys.exit(f"pack: --precision {args.precision} needs --fp8-scales <json from head_detector_amd.engine.calibrate_fp8>")
        scales = json.load(open(args.fp8_scales))
    P = arch.build_program(args.variant, _weights_arg(args.weights, args.variant), args.image_size, args.precision, fp8_scales=scales)
    names = tile_names_for(P, args.batch, args.split) if args.precision != "fp32" else {}
    n = write_pack(args.out, P, _flame_arg(args.flame), names, args.batch)
    print(f"{args.out}: {n / 2 ** 20:.1f} MiB, {len(P.ops)} ops, {len(P.bufs)} buffers, {len(names)} tuned tile choices, {P.flops / 1e9:.2f} GFLOP/image")


if __name__ == "__main__":
    main()
