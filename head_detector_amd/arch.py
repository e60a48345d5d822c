"""VGGHeads network -> flat op program for libvgh.so.

The reference never spells the network out: it ships a TorchScript blob (head_detector/detector.py:25-30)
whose graph is YoloHeads (yolo_head_training/yolo_head/yolo_heads.py:89-112) built from
yolo_head_training/configs/arch_params/yolo_heads_{m,l}_arch_params.yaml:4-137 with super_gradients
blocks.  This module (1) enumerates that module tree with super_gradients' parameter names
(``layer_specs``), (2) folds eval-mode BN / QARepVGG branches into one conv + bias per block in fp64
(``fold_state_dict``), and (3) lowers the folded network to the op program the C engine runs
(``build_program``): NHWC bf16 buffers, concat-by-offset (no torch.cat), fused sibling convs
(CSP conv1|conv2, head stems, cls|reg towers, first layers of the six FLAME branches, block-diagonal
prediction convs), residual-add and ConvTranspose pixel-shuffle in the conv epilogue.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

BN_EPS = 1e-6  # arch yaml :139

PRED_CLS_OFF, PRED_FLAME_OFF = 68, 72  # VGH_PRED_CLS_OFF / VGH_PRED_FLAME_OFF (include/vgh.h)

VARIANTS: Dict[str, dict] = {
    # yolo_heads_l_arch_params.yaml:4-137
    "vgg_heads_l": dict(
        stem=48,
        stages=[(96, 2, 96, True), (192, 3, 128, True), (384, 5, 256, True), (768, 2, 512, True)],  # (out, blocks, hidden, concat_intermediates)
        spp_out=768,
        neck=[(192, 4, 128), (96, 4, 128), (192, 4, 128), (384, 4, 256)],  # (out, blocks, hidden)
        head=dict(bbox=(128, 256, 512), flame=256, blocks=3, shape_inter=256, expr_inter=128, shape_out=128, expr_out=64, tr_inter=32, width_mult=1.0),
    ),
    # yolo_heads_m_arch_params.yaml (diff vs L: :16,37-38,55-56,65-66,75-76,84,95-134)
    "vgg_heads_m": dict(
        stem=48,
        stages=[(96, 2, 64, True), (192, 3, 128, True), (384, 5, 256, True), (768, 2, 384, False)],
        spp_out=768,
        neck=[(192, 2, 192), (96, 3, 64), (192, 2, 192), (384, 3, 256)],
        head=dict(bbox=(256, 256, 256), flame=256, blocks=2, shape_inter=128, expr_inter=64, shape_out=64, expr_out=32, tr_inter=16, width_mult=0.75),
    ),
}
# activation storage formats (VGH_FMT_* of include/vgh.h; the buffer field keeps its historical name is_f32) per precision mode
FMT_BF16, FMT_F32, FMT_BF16X2, FMT_F16X2, FMT_FP8, FMT_F16, FMT_I8 = 0, 1, 2, 3, 4, 5, 6
# "fp8" (r05) is the bf16 program with OCP-e4m3 LINKS: a tensor written by one 3x3 / stride-1 conv and read by one (bottleneck cv1 -> cv2, the middle layers of the
# FLAME shape / expression branches) is stored as e4m3 bytes with a calibrated per-tensor scale, and its consumer runs v_mfma_f32_32x32x64_f8f6f4 (csrc/conv_pp.hip)
# "fp16" (r05): ONE fp16 plane per value -- the reference's own FP16 export (exportable_mesh_model.py:177,299,409) -- same bytes and MFMA count as bf16, 11 significand bits
# "int8" (r05): the same links as signed bytes in [-127, 127] -- the reference exporter's QuantizationMode.INT8 (exportable_mesh_model.py:175-178,398-411) -- through
# v_mfma_i32_32x32x32_i8: a uniform grid (relative step 1 / 127 of the tensor's range instead of e4m3's 1 / 16 of the value) and an exact int32 accumulator
PRECISION_FMT = {"bf16": FMT_BF16, "fp32": FMT_F32, "bf16x3": FMT_BF16X2, "fp16x3": FMT_F16X2, "fp8": FMT_BF16, "fp16": FMT_F16, "int8": FMT_BF16}
FMT_BYTES = {FMT_BF16: 2, FMT_F32: 4, FMT_BF16X2: 4, FMT_F16X2: 4, FMT_FP8: 1, FMT_F16: 2, FMT_I8: 1}  # bytes per logical element
FP8_MAX, FP8_HEADROOM = 448.0, 2.0  # scale of an e4m3 link = calibrated max|activation| * headroom / 448
I8_MAX, I8_HEADROOM = 127.0, 1.25   # scale of an int8 link = calibrated max|activation| * headroom / 127 (a uniform grid pays for head-room linearly: less of it)
Q8_FMTS = (FMT_FP8, FMT_I8)
Q8_PRECISIONS = {"fp8": FMT_FP8, "int8": FMT_I8}  # the bf16 program with 8-bit links
TR_OUTS = (("rotation", 6), ("jaw", 3), ("translation", 3), ("scale", 1))  # order of the transform branches in the prediction buffer
STRIDES = (8, 16, 32)


def _wm(ch: int, factor: float, divisor: int = 8) -> int:
    return int(math.ceil(int(ch * factor) / divisor) * divisor)  # super_gradients width_multiplier


def head_dims(v: dict, level: int) -> dict:
    h = v["head"]
    return dict(bbox=_wm(h["bbox"][level], h["width_mult"]), fl=_wm(h["flame"], h["width_mult"]), **{k: h[k] for k in ("blocks", "shape_inter", "expr_inter", "shape_out", "expr_out", "tr_inter")})


# ------------------------------------------------------------------------------------------------------
# 1. module tree with super_gradients parameter names
# ------------------------------------------------------------------------------------------------------
@dataclass
class Spec:
    kind: str  # qarep | conv | cbr | plain | convT | alpha
    name: str
    cin: int = 0
    cout: int = 0
    k: int = 1
    stride: int = 1
    residual: bool = False
    use_alpha: bool = False


def _csp_specs(p: str, cin: int, cout: int, n: int, hidden: int, ci: bool, block: str) -> List[Spec]:
    s = [Spec("conv", f"{p}.conv1", cin, hidden, 1), Spec("conv", f"{p}.conv2", cin, hidden, 1), Spec("conv", f"{p}.conv3", hidden * (2 + (n if ci else 0)), cout, 1)]
    for i in range(n):
        for cv in ("cv1", "cv2"):
            if block == "qarep":
                s.append(Spec("qarep", f"{p}.bottlenecks.{i}.{cv}", hidden, hidden, 3, 1, residual=True))
            else:
                s.append(Spec("conv", f"{p}.bottlenecks.{i}.{cv}", hidden, hidden, 3, 1))
        s.append(Spec("alpha", f"{p}.bottlenecks.{i}.alpha"))
    return s


def layer_specs(variant: str) -> List[Spec]:
    v = VARIANTS[variant]
    s: List[Spec] = [Spec("qarep", "backbone.stem.conv", 3, v["stem"], 3, 2)]
    c = v["stem"]
    for i, (co, n, hid, ci) in enumerate(v["stages"]):
        p = f"backbone.stage{i + 1}"
        s.append(Spec("qarep", f"{p}.downsample", c, co, 3, 2))
        s += _csp_specs(f"{p}.blocks", co, co, n, hid, ci, "qarep")
        c = co
    s += [Spec("conv", "backbone.context_module.cv1", c, c // 2, 1), Spec("conv", "backbone.context_module.cv2", c // 2 * 4, v["spp_out"], 1)]
    c2, c3, c4 = (st[0] for st in v["stages"][:3])
    c5 = v["spp_out"]
    (o1, n1, h1), (o2, n2, h2), (o3, n3, h3), (o4, n4, h4) = v["neck"]
    for p, cin, s1, s2, o, n, h in (("neck.neck1", c5, c4, c3, o1, n1, h1), ("neck.neck2", o1, c3, c2, o2, n2, h2)):
        s += [
            Spec("conv", f"{p}.reduce_skip1", s1, o, 1),
            Spec("conv", f"{p}.reduce_skip2", s2, o, 1),
            Spec("conv", f"{p}.conv", cin, o, 1),
            Spec("convT", f"{p}.upsample", o, o, 2, 2),
            Spec("conv", f"{p}.downsample", o, o, 3, 2),
            Spec("conv", f"{p}.reduce_after_concat", 3 * o, o, 1),
        ]
        s += _csp_specs(f"{p}.blocks", o, o, n, h, False, "qarep")
    for p, cin, skip, o, n, h in (("neck.neck3", o2, o2, o3, n3, h3), ("neck.neck4", o3, o1, o4, n4, h4)):
        s.append(Spec("conv", f"{p}.conv", cin, o // 2, 3, 2))
        s += _csp_specs(f"{p}.blocks", o // 2 + skip, o, n, h, False, "conv")
    for lv, cin in enumerate((o2, o3, o4)):
        d = head_dims(v, lv)
        p = f"heads.head{lv + 1}"
        s += [
            Spec("cbr", f"{p}.pose_stem", cin, d["fl"], 1),
            Spec("cbr", f"{p}.bbox_stem", cin, d["bbox"], 1),
            Spec("cbr", f"{p}.cls_convs.0", d["bbox"], d["bbox"], 3),
            Spec("cbr", f"{p}.reg_convs.0", d["bbox"], d["bbox"], 3),
            Spec("plain", f"{p}.reg_pred", d["bbox"], 68, 1),
            Spec("plain", f"{p}.cls_pred", d["bbox"], 1, 1),
        ]
        branches = [("shape", d["shape_inter"], d["shape_out"]), ("expression", d["expr_inter"], d["expr_out"])] + [(n_, d["tr_inter"], o_) for n_, o_ in TR_OUTS]
        for bn, inter, out in branches:
            c_in = d["fl"]
            for b in range(d["blocks"]):
                s.append(Spec("qarep", f"{p}.flame_{bn}_pred.{b}", c_in, inter, 3, 1, residual=False, use_alpha=True))
                c_in = inter
            s.append(Spec("plain", f"{p}.flame_{bn}_pred.{d['blocks']}", inter, out, 1))
    return s


def module_graph(variant: str) -> List[dict]:
    """The reference network as a MODULE-level dataflow graph in forward (= export) order: one entry per conv module of ``layer_specs`` plus the tensor ops
    between them.  ``{"op": "conv" | "convT" | "add" | "concat" | "maxpool", "name": output tensor (= the module name for convs), "inputs": [tensor names, "image" first],
    "spec": Spec of a conv, "relu": a ReLU follows, "alpha": name of the bottleneck-alpha spec scaling inputs[0] of an add, "k": pool size}``.
    What it restates (the forward bodies of super_gradients' YoloNASStage / CSPLayer / Bottleneck / SPP / UpStage / DownStage as oracle/net_oracle.py states them, and
    yolo_head_training/yolo_head/yolo_head_dfl_head.py:143-164 for the heads); used by onnx_graph.py to bind the anonymous Conv nodes of a simplified ONNX export by
    graph position, and by the tests' exporter stand-in.  ``tests/test_host_logic.py`` pins it against the oracle's module tree run under forward hooks."""
    v = VARIANTS[variant]
    sp = {s.name: s for s in layer_specs(variant)}
    g: List[dict] = []

    def conv(name, x, relu=True):
        s = sp[name]
        g.append(dict(op="convT" if s.kind == "convT" else "conv", name=name, inputs=[x], spec=s, relu=relu and s.kind not in ("plain", "convT")))
        return name

    def csp(p, x, n, ci):
        x1 = conv(f"{p}.conv1", x)
        outs = [x1]
        for i in range(n):
            a = conv(f"{p}.bottlenecks.{i}.cv1", x1)
            b = conv(f"{p}.bottlenecks.{i}.cv2", a)
            g.append(dict(op="add", name=f"{p}.bottlenecks.{i}", inputs=[x1, b], alpha=f"{p}.bottlenecks.{i}.alpha"))
            x1 = f"{p}.bottlenecks.{i}"
            outs.append(x1)
        if not ci:
            outs = outs[-1:]
        x2 = conv(f"{p}.conv2", x)
        g.append(dict(op="concat", name=f"{p}.cat", inputs=outs + [x2]))
        return conv(f"{p}.conv3", f"{p}.cat")

    x = conv("backbone.stem.conv", "image")
    feats = []
    for i, (co, n, hid, ci) in enumerate(v["stages"]):
        p = f"backbone.stage{i + 1}"
        x = csp(f"{p}.blocks", conv(f"{p}.downsample", x), n, ci)
        feats.append(x)
    c2, c3, c4, _ = feats
    y = conv("backbone.context_module.cv1", x)
    pools = []
    for k in (5, 9, 13):
        g.append(dict(op="maxpool", name=f"backbone.context_module.m{k}", inputs=[y], k=k))
        pools.append(f"backbone.context_module.m{k}")
    g.append(dict(op="concat", name="backbone.context_module.cat", inputs=[y] + pools))
    c5 = conv("backbone.context_module.cv2", "backbone.context_module.cat")

    def up(p, x, s1, s2, n):
        a, b = conv(f"{p}.reduce_skip1", s1), conv(f"{p}.reduce_skip2", s2)
        inter = conv(f"{p}.conv", x)
        u = conv(f"{p}.upsample", inter)
        d = conv(f"{p}.downsample", b)
        g.append(dict(op="concat", name=f"{p}.cat", inputs=[u, a, d]))
        return inter, csp(f"{p}.blocks", conv(f"{p}.reduce_after_concat", f"{p}.cat"), n, False)

    def down(p, x, skip, n):
        c = conv(f"{p}.conv", x)
        g.append(dict(op="concat", name=f"{p}.cat", inputs=[c, skip]))
        return csp(f"{p}.blocks", f"{p}.cat", n, False)

    (_, n1, _), (_, n2, _), (_, n3, _), (_, n4, _) = v["neck"]
    i1, x = up("neck.neck1", c5, c4, c3, n1)
    i2, p3 = up("neck.neck2", x, c3, c2, n2)
    p4 = down("neck.neck3", p3, i2, n3)
    p5 = down("neck.neck4", p4, i1, n4)
    for lv, f in enumerate((p3, p4, p5)):
        p = f"heads.head{lv + 1}"
        nb = head_dims(v, lv)["blocks"]
        pose, bb = conv(f"{p}.pose_stem", f), conv(f"{p}.bbox_stem", f)
        conv(f"{p}.cls_pred", conv(f"{p}.cls_convs.0", bb))  # yolo_head_dfl_head.py:147-153: the cls tower first, then the reg tower
        conv(f"{p}.reg_pred", conv(f"{p}.reg_convs.0", bb))
        for br in ("shape", "expression", "rotation", "jaw", "translation", "scale"):  # :155-160
            t = pose
            for b in range(nb + 1):
                t = conv(f"{p}.flame_{br}_pred.{b}", t)
    return g


def _bn_names(p: str) -> List[str]:
    return [f"{p}.weight", f"{p}.bias", f"{p}.running_mean", f"{p}.running_var"]


def random_state_dict(variant: str, seed: int = 1) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights of the architecture's exact shapes, variance-preserving so
    that activations stay O(1) through ~150 layers like a trained network's (random BN statistics that are NOT matched
    to the data would otherwise multiply the signal by ~1.1 per layer and overflow the exp() in the scale branch):
      conv+BN+ReLU     W ~ N(0, 2/K), BN scale gamma/sqrt(var) = s,        s = exp(U(-0.1, 0.1))
      QARepVGG         W3 ~ N(0, 1/K), W1 ~ N(0, 0.25/Cin), post-BN scale 1.26 s (0.85 s with the identity branch)
      bottleneck alpha ~ U(0.05, 0.2);  BN beta, mean ~ N(0, 0.1), var ~ U(0.5, 1.5)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}

    def conv_w(co, ci, k, gain=2.0):
        return (rng.standard_normal((co, ci, k, k)) * math.sqrt(gain / (ci * k * k))).astype(np.float32)

    def bn(p, c, target=1.0):
        var = rng.uniform(0.5, 1.5, c)
        scale = target * np.exp(rng.uniform(-0.1, 0.1, c))
        sd[f"{p}.weight"] = (scale * np.sqrt(var + BN_EPS)).astype(np.float32)
        sd[f"{p}.bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[f"{p}.running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[f"{p}.running_var"] = var.astype(np.float32)

    for sp in layer_specs(variant):
        if sp.kind == "qarep":
            sd[f"{sp.name}.branch_3x3.conv.weight"] = conv_w(sp.cout, sp.cin, 3, 1.0)
            bn(f"{sp.name}.branch_3x3.bn", sp.cout)
            sd[f"{sp.name}.branch_1x1.weight"] = conv_w(sp.cout, sp.cin, 1, 0.25)
            sd[f"{sp.name}.branch_1x1.bias"] = (rng.standard_normal(sp.cout) * 0.1).astype(np.float32)
            bn(f"{sp.name}.post_bn", sp.cout, 0.85 if sp.residual else 1.26)
            if sp.use_alpha:
                sd[f"{sp.name}.alpha"] = rng.uniform(0.5, 1.5, 1).astype(np.float32)
        elif sp.kind == "conv":
            sd[f"{sp.name}.conv.weight"] = conv_w(sp.cout, sp.cin, sp.k)
            bn(f"{sp.name}.bn", sp.cout)
        elif sp.kind == "cbr":
            sd[f"{sp.name}.seq.conv.weight"] = conv_w(sp.cout, sp.cin, sp.k)
            bn(f"{sp.name}.seq.bn", sp.cout)
        elif sp.kind == "plain":
            sd[f"{sp.name}.weight"] = conv_w(sp.cout, sp.cin, sp.k, 1.0)
            sd[f"{sp.name}.bias"] = (rng.standard_normal(sp.cout) * 0.1).astype(np.float32)
        elif sp.kind == "convT":
            sd[f"{sp.name}.weight"] = (rng.standard_normal((sp.cin, sp.cout, 2, 2)) * math.sqrt(1.0 / sp.cin)).astype(np.float32)
            sd[f"{sp.name}.bias"] = (rng.standard_normal(sp.cout) * 0.1).astype(np.float32)
        elif sp.kind == "alpha":
            sd[sp.name] = rng.uniform(0.05, 0.2, 1).astype(np.float32)
    for lv in range(3):
        # cls bias prior of YoloHeadsDFLHead._initialize_biases (yolo_head_dfl_head.py:188-190); modest logit spread
        sd[f"heads.head{lv + 1}.cls_pred.bias"][:] = -math.log((1 - 1e-2) / 1e-2)
        sd[f"heads.head{lv + 1}.cls_pred.weight"] *= np.float32(0.02)
        for k in list(sd):  # keep every raw FLAME prediction O(1) like a trained head's (scale = exp(x)/0.05, 3*tanh(x), ...)
            if re.fullmatch(rf"heads\.head{lv + 1}\.flame_(scale|translation|rotation|jaw|shape|expression)_pred\.\d+\.(weight|bias)", k):
                sd[k] = (sd[k] * 0.05).astype(np.float32)
    return sd


# ------------------------------------------------------------------------------------------------------
# 2. folding (fp64)
# ------------------------------------------------------------------------------------------------------
def _bn_affine(sd, p) -> Tuple[np.ndarray, np.ndarray]:
    g, b, m, var = (np.asarray(sd[n], dtype=np.float64) for n in _bn_names(p))
    s = g / np.sqrt(var + BN_EPS)
    return s, b - m * s


def fold_state_dict(variant: str, sd: Dict[str, np.ndarray]) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """name -> (W [Cout,Cin,k,k] f64, b [Cout] f64); bottleneck alphas as name -> (alpha, None).
    Raises KeyError / ValueError on missing keys or shape mismatches (the manifest check of SURVEY.md N1)."""
    out: Dict[str, Tuple[np.ndarray, Optional[np.ndarray]]] = {}

    def need(key, shape):
        a = np.asarray(sd[key])
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        return a.astype(np.float64)

    for sp in layer_specs(variant):
        n = sp.name
        if sp.kind == "qarep" and f"{n}.branch_3x3.conv.weight" not in sd and f"{n}.rbr_reparam.weight" in sd:
            # an archive exported after super_gradients' QARepVGGBlock fusion.  `rbr_reparam` then holds the fused
            # 3x3 conv -- branches + alpha (+ identity) folded -- and, after FULL fusion, post_bn too (post_bn is an Identity: no keys);
            # after PARTIAL fusion post_bn is still a BatchNorm and is folded here.
            K = need(f"{n}.rbr_reparam.weight", (sp.cout, sp.cin, 3, 3))
            bias = need(f"{n}.rbr_reparam.bias", (sp.cout,))
            if f"{n}.post_bn.running_var" in sd:
                sp_, tp = _bn_affine(sd, f"{n}.post_bn")
                K, bias = K * sp_[:, None, None, None], bias * sp_ + tp
            out[n] = (K, bias)
        elif sp.kind == "qarep":
            w3 = need(f"{n}.branch_3x3.conv.weight", (sp.cout, sp.cin, 3, 3))
            s3, t3 = _bn_affine(sd, f"{n}.branch_3x3.bn")
            w1 = need(f"{n}.branch_1x1.weight", (sp.cout, sp.cin, 1, 1))
            b1 = need(f"{n}.branch_1x1.bias", (sp.cout,))
            alpha = float(np.asarray(sd[f"{n}.alpha"]).reshape(-1)[0]) if sp.use_alpha else 1.0
            K = w3 * s3[:, None, None, None]
            K[:, :, 1, 1] += alpha * w1[:, :, 0, 0]
            if sp.residual:
                K[np.arange(sp.cout), np.arange(sp.cout), 1, 1] += 1.0
            bias = t3 + alpha * b1
            sp_, tp = _bn_affine(sd, f"{n}.post_bn")
            out[n] = (K * sp_[:, None, None, None], bias * sp_ + tp)
        elif sp.kind in ("conv", "cbr"):
            p = n if sp.kind == "conv" else f"{n}.seq"
            w = need(f"{p}.conv.weight", (sp.cout, sp.cin, sp.k, sp.k))
            if f"{p}.conv.bias" in sd and f"{p}.bn.running_var" not in sd:
                # u8: the exporter already merged the eval-mode BatchNorm into the conv (an ONNX export that kept the conv's name): nothing left to fold
                out[n] = (w, need(f"{p}.conv.bias", (sp.cout,)))
            else:
                s, t = _bn_affine(sd, f"{p}.bn")
                out[n] = (w * s[:, None, None, None], t)
        elif sp.kind == "plain":
            out[n] = (need(f"{n}.weight", (sp.cout, sp.cin, sp.k, sp.k)), need(f"{n}.bias", (sp.cout,)))
        elif sp.kind == "convT":
            out[n] = (need(f"{n}.weight", (sp.cin, sp.cout, 2, 2)), need(f"{n}.bias", (sp.cout,)))
        elif sp.kind == "alpha":
            out[n] = (float(np.asarray(sd[n]).reshape(-1)[0]), None)
    return out


# ------------------------------------------------------------------------------------------------------
# 3. lowering to the op program
# ------------------------------------------------------------------------------------------------------
def _r32(c: int) -> int:
    return (c + 31) // 32 * 32


@dataclass
class View:
    buf: int
    coff: int
    c: int  # channels visible through this view (padded to 32 where it feeds a conv)


@dataclass
class Program:
    variant: str
    image_size: int
    bufs: List[dict] = field(default_factory=list)
    ops: List[dict] = field(default_factory=list)
    weights: List[np.ndarray] = field(default_factory=list)
    biases: List[np.ndarray] = field(default_factory=list)
    w_elems: int = 0
    b_elems: int = 0
    levels: List[dict] = field(default_factory=list)  # prediction buffers per level
    shape_c: int = 0
    expr_c: int = 0
    flops: float = 0.0  # algorithmic 2*MACs per image (fused-conv accounting)
    precision: str = "bf16"

    def buf(self, name: str, h: int, w: int, pitch: int, f32: bool = False, fp8_scale: Optional[float] = None, q8_fmt: int = FMT_FP8) -> int:
        """fp8_scale: the buffer is an 8-bit link (q8_fmt: FMT_FP8 e4m3 / FMT_I8 int8), value = stored * fp8_scale."""
        self.bufs.append(dict(name=name, h=h, w=w, pitch=pitch, is_f32=int(f32)) if fp8_scale is None else dict(name=name, h=h, w=w, pitch=pitch, is_f32=q8_fmt, scale=float(fp8_scale)))
        return len(self.bufs) - 1

    def _push_w(self, W: np.ndarray, b: np.ndarray) -> Tuple[int, int]:
        wo, bo = self.w_elems, self.b_elems
        self.weights.append(np.ascontiguousarray(W, dtype=np.float32).reshape(-1))
        self.biases.append(np.ascontiguousarray(b, dtype=np.float32).reshape(-1))
        self.w_elems += self.weights[-1].size
        self.b_elems += self.biases[-1].size
        return wo, bo

    def conv(self, name: str, src: View, dst: View, W: np.ndarray, b: np.ndarray, k: int, stride: int = 1, act: int = 1, res: Optional[Tuple[View, float]] = None,
             split: Optional[Tuple[int, int]] = None, shuffle: bool = False, cout_store: Optional[int] = None, flops_macs: Optional[float] = None,
             groups: Optional[Tuple[int, int]] = None):
        """W: [rows, k, k, cin_view] (rows padded to 32 here); b: [rows].  groups = (grp_cout, grp_in_stride): rows [g*grp_cout, (g+1)*grp_cout)
        read the cin_view channels starting at src.coff + g*grp_in_stride (sibling branches as one block-diagonal launch)."""
        rows, kk1, kk2, cin = W.shape
        assert kk1 == k and kk2 == k and cin == src.c and cin % 32 == 0, (name, W.shape, src)
        rp = _r32(rows)
        Wp = np.zeros((rp, k, k, cin), dtype=np.float64)
        Wp[:rows] = W
        bp = np.zeros(rp, dtype=np.float64)
        bp[:rows] = b
        wo, bo = self._push_w(Wp, bp)
        ib = self.bufs[src.buf]
        ho = (ib["h"] + 2 * (k // 2) - k) // stride + 1
        wo_ = (ib["w"] + 2 * (k // 2) - k) // stride + 1
        self.ops.append(dict(
            name=name, kind=1, in_buf=src.buf, in_coff=src.coff, cin=cin, out_buf=dst.buf, out_coff=dst.coff, cout_pad=rp,
            cout_store=cout_store if cout_store is not None else rows, out_split=split[0] if split else rp, out_coff2=split[1] if split else 0,
            res_buf=res[0].buf if res else -1, res_coff=res[0].coff if res else 0, alpha=float(res[1]) if res else 0.0,
            ksize=k, stride=stride, act=act, shuffle=int(shuffle), w_off=wo, b_off=bo, force_cfg=-1,
            grp_cout=groups[0] if groups else 0, grp_in_stride=groups[1] if groups else 0,
            macs=float(ho * wo_) * float(flops_macs if flops_macs is not None else rows * k * k * cin),
            gemm=(ho * wo_, rp, k * k * cin),
        ))
        if groups:
            assert rp % groups[0] == 0 and groups[0] % 32 == 0, (name, rp, groups)

    def arrays(self):
        return np.concatenate(self.weights), np.concatenate(self.biases)


def _ohwi(W: np.ndarray, cin_pad: int) -> np.ndarray:
    """torch OIHW -> [O,H,W,I] with the input channels zero-padded to cin_pad."""
    o, i, kh, kw = W.shape
    out = np.zeros((o, kh, kw, cin_pad), dtype=np.float64)
    out[..., :i] = np.transpose(W, (0, 2, 3, 1))
    return out


def _stack(parts: List[Tuple[np.ndarray, np.ndarray]], pad_to: int = 1) -> Tuple[np.ndarray, np.ndarray, List[int]]:
    """Stack [rows_i,k,k,cin] blocks along rows, each padded to a multiple of pad_to. Returns W, b, row offsets."""
    ws, bs, offs, cur = [], [], [], 0
    for W, b in parts:
        r = W.shape[0]
        rp = (r + pad_to - 1) // pad_to * pad_to
        Wp = np.zeros((rp,) + W.shape[1:], dtype=np.float64)
        Wp[:r] = W
        bp = np.zeros(rp, dtype=np.float64)
        bp[:r] = b
        ws.append(Wp)
        bs.append(bp)
        offs.append(cur)
        cur += rp
    return np.concatenate(ws), np.concatenate(bs), offs


STEM_PITCH_BF16 = 48  # 64 restores the zero-padded stem tensor of r01 - r03 (tools/ab_stem_pitch.py measures one against the other)


def fp8_link_names(variant: str, image_size: int = 640, fp8_min_px: int = 40) -> Dict[str, Tuple[str, int]]:
    """8-bit link buffer of the "fp8" (and "int8": same links) program -> (buffer of the bf16 program that holds the same tensor, its leading channels): what a calibration forward of the
    bf16 engine has to look at (engine.calibrate_fp8)."""
    P = build_program(variant, random_state_dict(variant, 0), image_size, "fp8", fp8_scales={}, fp8_min_px=fp8_min_px)
    out = {}
    for bf in P.bufs:
        if bf["is_f32"] in Q8_FMTS:
            out[bf["name"]] = (bf["name"][:-1], bf["live"]) if bf["name"].endswith("q") else (bf["name"], bf["live"])
    return out


def build_program(variant: str, sd: Dict[str, np.ndarray], image_size: int = 640, precision: str = "bf16", head_lanes: bool = False,
                  fp8_scales: Optional[Dict[str, float]] = None, fp8_min_px: int = 40) -> Program:
    """precision: 'bf16' (throughput mode: bf16 activations/weights, fp32 accumulate); 'fp16x3' (matrix-core parity mode: two fp16 planes
    per value, three MFMAs per product, csrc/conv_split.hip); 'bf16x3' (the same with bf16 planes: 16 significand bits, kept for the
    comparison); 'fp32' (VALU parity mode: no 16-bit format anywhere); 'fp8' (bf16 with e4m3 links between 3x3 / stride-1 convs on maps of at least
    fp8_min_px pixels a side -- the 8 x 8 sub-patches of the ping-pong tiles waste a third of a 20-wide map; fp8_scales: link buffer name -> max|activation| from a
    calibration forward, see engine.calibrate_fp8; a missing entry takes max = 8)."""
    assert precision in PRECISION_FMT, precision
    fp8 = precision in Q8_PRECISIONS  # 8-bit links (e4m3 or int8)
    q8_fmt = Q8_PRECISIONS.get(precision, FMT_FP8)
    fp8_scales = fp8_scales or {}

    def link_scale(name: str) -> float:
        amax = max(float(fp8_scales.get(name, 8.0)), 1e-6)
        return amax * I8_HEADROOM / I8_MAX if q8_fmt == FMT_I8 else amax * FP8_HEADROOM / FP8_MAX

    def r64(c: int) -> int:
        return (c + 63) // 64 * 64
    v = VARIANTS[variant]
    F = fold_state_dict(variant, sd)
    P = Program(variant=variant, image_size=image_size)
    S = image_size
    assert S % 32 == 0

    # ---------------- stem (own kernel: exact fp32, K = 27) ----------------
    W, b = F["backbone.stem.conv"]
    assert v["stem"] == 48
    wo, bo = P._push_w(np.transpose(W, (0, 2, 3, 1)), b)  # [48][ky][kx][ci]
    # bf16 throughput mode: the stem tensor is stored at its own 48-channel pitch (96-byte pixels); the stage-1 downsample still reads 64-channel K
    # windows, whose last 16 channels (the next pixel's first 16) meet the 16 zero weight columns _ohwi pads in below -- the executor verifies that at
    # vgh_net_create.  The parity modes keep the 64-channel pitch with stored zeros (a split pixel is [hi | lo] planes: no such window).
    stem_pitch = STEM_PITCH_BF16 if precision in ("bf16", "fp8", "int8") else 64
    stem_buf = P.buf("stem", S // 2, S // 2, stem_pitch)
    P.ops.append(dict(name="backbone.stem.conv", kind=0, in_buf=-1, in_coff=0, cin=3, out_buf=stem_buf, out_coff=0, cout_pad=64, cout_store=stem_pitch, out_split=64,
                      out_coff2=0, res_buf=-1, res_coff=0, alpha=0.0, ksize=3, stride=2, act=1, shuffle=0, w_off=wo, b_off=bo, force_cfg=-1,
                      macs=float((S // 2) ** 2) * 48 * 27, gemm=((S // 2) ** 2, 48, 27)))
    x = View(stem_buf, 0, 64)
    x_real = 48

    def csp(p: str, xin: View, cin_real: int, out: View, n: int, hid: int, ci: bool, res_px: int):
        """YoloNASCSPLayer. One buffer holds every tensor of the layer, each written exactly once:
        ci -> [x1_0 | b_1 .. b_n | x2] (conv3 reads all of it); else [b_n | x2 | x1_0 | b_1 .. b_{n-1}] (conv3 reads the first two)."""
        assert hid % 32 == 0
        slots = (n + 2) if ci else 2
        cat = P.buf(f"{p}.cat", res_px, res_px, hid * (n + 2))
        w1, b1 = F[f"{p}.conv1"]
        w2, b2 = F[f"{p}.conv2"]
        Wf, bf, _ = _stack([(_ohwi(w1, xin.c), b1), (_ohwi(w2, xin.c), b2)])
        x2_off = hid * (n + 1) if ci else hid
        x1_off = 0 if ci else hid * 2
        P.conv(f"{p}.conv1|conv2", xin, View(cat, x1_off, hid), Wf, bf, 1, split=(hid, x2_off), flops_macs=2 * hid * cin_real)
        prev = x1_off
        for i in range(n):
            dst = hid * (i + 1) if ci else (0 if i == n - 1 else hid * (3 + i))
            wa, ba = F[f"{p}.bottlenecks.{i}.cv1"]
            wb, bb = F[f"{p}.bottlenecks.{i}.cv2"]
            alpha = F[f"{p}.bottlenecks.{i}.alpha"][0]
            if fp8 and res_px >= fp8_min_px:
                # cv1 -> cv2 is a single-writer / single-reader link: e4m3 bytes, K blocks of 64 channels (a 96-channel tensor sits in a 128-byte pixel whose last 32
                # bytes are never written: the arena's zero bytes are e4m3 +0 and meet zero weight columns)
                mid = P.buf(f"{p}.mid{i}", res_px, res_px, r64(hid), fp8_scale=link_scale(f"{p}.mid{i}"), q8_fmt=q8_fmt)
                P.bufs[mid]["live"] = hid
                P.conv(f"{p}.bottlenecks.{i}.cv1", View(cat, prev, hid), View(mid, 0, hid), _ohwi(wa, hid), ba, 3)
                P.conv(f"{p}.bottlenecks.{i}.cv2", View(mid, 0, r64(hid)), View(cat, dst, hid), _ohwi(wb, r64(hid)), bb, 3, res=(View(cat, prev, hid), alpha), flops_macs=hid * 9 * hid)
            else:
                mid = P.buf(f"{p}.mid{i}", res_px, res_px, hid)
                P.conv(f"{p}.bottlenecks.{i}.cv1", View(cat, prev, hid), View(mid, 0, hid), _ohwi(wa, hid), ba, 3)
                P.conv(f"{p}.bottlenecks.{i}.cv2", View(mid, 0, hid), View(cat, dst, hid), _ohwi(wb, hid), bb, 3, res=(View(cat, prev, hid), alpha))
            prev = dst
        w3, b3 = F[f"{p}.conv3"]
        P.conv(f"{p}.conv3", View(cat, 0, hid * slots), out, _ohwi(w3, hid * slots), b3, 1)

    # ---------------- backbone stages ----------------
    feats: List[View] = []
    res_px = S // 2
    for i, (co, n, hid, ci) in enumerate(v["stages"]):
        p = f"backbone.stage{i + 1}"
        res_px //= 2
        ds = P.buf(f"{p}.ds", res_px, res_px, co)
        W, b = F[f"{p}.downsample"]
        P.conv(f"{p}.downsample", x, View(ds, 0, co), _ohwi(W, x.c), b, 3, stride=2, flops_macs=co * 9 * x_real)
        ob = P.buf(f"{p}.out", res_px, res_px, co)
        csp(f"{p}.blocks", View(ds, 0, co), co, View(ob, 0, co), n, hid, ci, res_px)
        x, x_real = View(ob, 0, co), co
        feats.append(x)
    # ---------------- SPP ----------------
    c = v["stages"][3][0]
    hidden = c // 2
    spp = P.buf("spp.cat", res_px, res_px, hidden * 4)
    W, b = F["backbone.context_module.cv1"]
    P.conv("backbone.context_module.cv1", x, View(spp, 0, hidden), _ohwi(W, c), b, 1)
    P.ops.append(dict(name="backbone.context_module.m", kind=2, in_buf=spp, in_coff=0, cin=hidden, out_buf=spp, out_coff=hidden, cout_pad=0, cout_store=0, out_split=0,
                      out_coff2=0, res_buf=-1, res_coff=0, alpha=0.0, ksize=5, stride=1, act=0, shuffle=0, w_off=0, b_off=0, force_cfg=-1, macs=0.0, gemm=(0, 0, 0)))
    c5b = P.buf("c5", res_px, res_px, v["spp_out"])
    W, b = F["backbone.context_module.cv2"]
    P.conv("backbone.context_module.cv2", View(spp, 0, hidden * 4), View(c5b, 0, v["spp_out"]), _ohwi(W, hidden * 4), b, 1)
    c2, c3, c4 = feats[0], feats[1], feats[2]
    c5 = View(c5b, 0, v["spp_out"])
    px = {2: S // 4, 3: S // 8, 4: S // 16, 5: S // 32}

    # ---------------- neck ----------------
    (o1, n1, h1), (o2, n2, h2), (o3, n3, h3), (o4, n4, h4) = v["neck"]
    # concat buffers of the two down stages are allocated first: the up stages' x_inter land directly inside them
    n3cat = P.buf("neck3.cat", px[4], px[4], o3 // 2 + o2)  # [conv(p3) | x_n2_inter]
    n4cat = P.buf("neck4.cat", px[5], px[5], o4 // 2 + o1)  # [conv(p4) | x_n1_inter]

    def up_stage(p: str, xin: View, s1: View, s2: View, o: int, n: int, hid: int, res_lo: int, inter_dst: View, out: View):
        """YoloNASUpStage (3 inputs, reduce_channels): cat = [upsample(conv(x)) | reduce_skip1(s1) | downsample(reduce_skip2(s2))]."""
        res_hi = res_lo * 2
        cat = P.buf(f"{p}.cat", res_hi, res_hi, 3 * o)
        W, b = F[f"{p}.conv"]
        P.conv(f"{p}.conv", xin, inter_dst, _ohwi(W, xin.c), b, 1)
        Wt, bt = F[f"{p}.upsample"]  # [cin, cout, 2, 2]
        Wg = np.zeros((4 * o, 1, 1, o), dtype=np.float64)
        for dy in range(2):
            for dx in range(2):
                Wg[(dy * 2 + dx) * o : (dy * 2 + dx + 1) * o, 0, 0, :] = Wt[:, :, dy, dx].T
        P.conv(f"{p}.upsample", inter_dst, View(cat, 0, o), Wg, np.tile(bt, 4), 1, act=0, shuffle=True)
        W, b = F[f"{p}.reduce_skip1"]
        P.conv(f"{p}.reduce_skip1", s1, View(cat, o, o), _ohwi(W, s1.c), b, 1)
        rs2 = P.buf(f"{p}.rs2", res_hi * 2, res_hi * 2, o)
        W, b = F[f"{p}.reduce_skip2"]
        P.conv(f"{p}.reduce_skip2", s2, View(rs2, 0, o), _ohwi(W, s2.c), b, 1)
        W, b = F[f"{p}.downsample"]
        P.conv(f"{p}.downsample", View(rs2, 0, o), View(cat, 2 * o, o), _ohwi(W, o), b, 3, stride=2)
        rac = P.buf(f"{p}.rac", res_hi, res_hi, o)
        W, b = F[f"{p}.reduce_after_concat"]
        P.conv(f"{p}.reduce_after_concat", View(cat, 0, 3 * o), View(rac, 0, o), _ohwi(W, 3 * o), b, 1)
        csp(f"{p}.blocks", View(rac, 0, o), o, out, n, hid, False, res_hi)

    n1out = P.buf("neck1.out", px[4], px[4], o1)
    up_stage("neck.neck1", c5, c4, c3, o1, n1, h1, px[5], View(n4cat, o4 // 2, o1), View(n1out, 0, o1))
    p3b = P.buf("p3", px[3], px[3], o2)
    up_stage("neck.neck2", View(n1out, 0, o1), c3, c2, o2, n2, h2, px[4], View(n3cat, o3 // 2, o2), View(p3b, 0, o2))
    p3 = View(p3b, 0, o2)
    W, b = F["neck.neck3.conv"]
    P.conv("neck.neck3.conv", p3, View(n3cat, 0, o3 // 2), _ohwi(W, o2), b, 3, stride=2)
    p4b = P.buf("p4", px[4], px[4], o3)
    csp("neck.neck3.blocks", View(n3cat, 0, o3 // 2 + o2), o3 // 2 + o2, View(p4b, 0, o3), n3, h3, False, px[4])
    p4 = View(p4b, 0, o3)
    W, b = F["neck.neck4.conv"]
    P.conv("neck.neck4.conv", p4, View(n4cat, 0, o4 // 2), _ohwi(W, o3), b, 3, stride=2)
    p5b = P.buf("p5", px[5], px[5], o4)
    csp("neck.neck4.blocks", View(n4cat, 0, o4 // 2 + o1), o4 // 2 + o1, View(p5b, 0, o4), n4, h4, False, px[5])
    p5 = View(p5b, 0, o4)

    # ---------------- heads: three independent branches.  head_lanes=True puts levels 1/2 on side HIP streams (VGH_OP_FORK);
    # measured SLOWER on MI355X (M b32: 4577 vs 4960 img/s -- the tuned tiles already fill the chip and concurrent kernels
    # thrash each other's L2), so it is off by default and kept only as an executor feature. ----
    if head_lanes:
        P.ops.append(dict(name="fork.heads", kind=3, in_buf=0, in_coff=0, cin=0, out_buf=0, out_coff=0, cout_pad=0, cout_store=0, out_split=0, out_coff2=0, res_buf=-1,
                          res_coff=0, alpha=0.0, ksize=0, stride=0, act=0, shuffle=0, w_off=0, b_off=0, force_cfg=-1, lane=0, macs=0.0, gemm=(0, 0, 0)))
    for lv, (feat, stride) in enumerate(zip((p3, p4, p5), STRIDES)):
        first_head_op = len(P.ops)
        d = head_dims(v, lv)
        p = f"heads.head{lv + 1}"
        r = S // stride
        fl, bb, nb = d["fl"], d["bbox"], d["blocks"]
        Sc, Ec, tr = d["shape_out"], d["expr_out"], d["tr_inter"]
        trp = _r32(tr)
        FO = PRED_FLAME_OFF  # [reg 68 | cls 1 | 3 unused | shape | expr | rot 6 jaw 3 trans 3 scale 1]: 16-byte aligned segments
        pred_pitch = (FO + Sc + Ec + 13 + 3) // 4 * 4
        pred = P.buf(f"{p}.pred", r, r, pred_pitch, f32=True)
        # stems: [pose | bbox]
        hs = P.buf(f"{p}.stems", r, r, fl + bb)
        (wp, bp_), (wb, bb_) = F[f"{p}.pose_stem"], F[f"{p}.bbox_stem"]
        Wf, bf, _ = _stack([(_ohwi(wp, feat.c), bp_), (_ohwi(wb, feat.c), bb_)])
        P.conv(f"{p}.pose_stem|bbox_stem", feat, View(hs, 0, fl + bb), Wf, bf, 1)
        # towers: [cls_feat | reg_feat]
        cr = P.buf(f"{p}.clsreg", r, r, 2 * bb)
        (wc, bc), (wr, br) = F[f"{p}.cls_convs.0"], F[f"{p}.reg_convs.0"]
        Wf, bf, _ = _stack([(_ohwi(wc, bb), bc), (_ohwi(wr, bb), br)])
        P.conv(f"{p}.cls_convs|reg_convs", View(hs, fl, bb), View(cr, 0, 2 * bb), Wf, bf, 3)
        # reg_pred (rows 0..67 <- reg_feat) + cls_pred (row 68 <- cls_feat): block-diagonal 1x1, fp32 out
        (wrp, brp), (wcp, bcp) = F[f"{p}.reg_pred"], F[f"{p}.cls_pred"]
        Wd = np.zeros((69, 1, 1, 2 * bb), dtype=np.float64)
        Wd[:68, 0, 0, bb:] = wrp[:, :, 0, 0]
        Wd[68, 0, 0, :bb] = wcp[0, :, 0, 0]
        P.conv(f"{p}.reg_pred|cls_pred", View(cr, 0, 2 * bb), View(pred, 0, 69), Wd, np.concatenate([brp, bcp]), 1, act=0, flops_macs=69 * bb)
        # FLAME branches: layer 0 of all six fused (shared input pose_features)
        names = ["shape", "expression"] + [n_ for n_, _ in TR_OUTS]
        inters = [d["shape_inter"], d["expr_inter"]] + [tr] * 4
        outs = [Sc, Ec] + [o_ for _, o_ in TR_OUTS]
        parts = [(_ohwi(F[f"{p}.flame_{n_}_pred.0"][0], fl), F[f"{p}.flame_{n_}_pred.0"][1]) for n_ in names]
        Wf, bf, offs = _stack(parts, pad_to=32)
        width = Wf.shape[0]
        cur = P.buf(f"{p}.f0", r, r, width)
        P.conv(f"{p}.flame_*_pred.0", View(hs, 0, fl), View(cur, 0, width), Wf, bf, 3, cout_store=width, flops_macs=sum(inters) * 9 * fl)
        assert offs[3] == offs[2] + trp and offs[5] == offs[2] + 3 * trp
        cur_q = None  # fp8 mode: the shape | expression part of `cur` as an e4m3 link
        for bi in range(1, nb):
            nxt = P.buf(f"{p}.f{bi}", r, r, width)
            # layer bi -> layer bi + 1 of the shape / expression branches is a 3x3 -> 3x3 link as long as layer bi + 1 is not the 1x1 prediction conv
            link_out = fp8 and r >= fp8_min_px and bi + 1 < nb and all(_r32(c) % 64 == 0 for c in inters[:2])
            nxt_q = None
            if link_out:
                nxt_q = P.buf(f"{p}.f{bi}q", r, r, offs[2], fp8_scale=link_scale(f"{p}.f{bi}q"), q8_fmt=q8_fmt)
                P.bufs[nxt_q]["live"] = offs[2]
            for n_, inter, off in zip(names[:2], inters[:2], offs[:2]):
                W, b = F[f"{p}.flame_{n_}_pred.{bi}"]
                ip = _r32(inter)
                P.conv(f"{p}.flame_{n_}_pred.{bi}", View(cur_q if cur_q is not None else cur, off, ip), View(nxt_q if nxt_q is not None else nxt, off, ip), _ohwi(W, ip), b, 3, cout_store=ip,
                       flops_macs=inter * 9 * inter)
            # the four transform branches (rotation / jaw / translation / scale: 3x3, tr -> tr each) are ONE grouped launch: cout group g
            # reads its own trp-channel window of the previous layer (four launches of a [M,32,288] GEMM could not fill the chip)
            parts = [(_ohwi(F[f"{p}.flame_{n_}_pred.{bi}"][0], trp), F[f"{p}.flame_{n_}_pred.{bi}"][1]) for n_, _ in TR_OUTS]
            if precision == "fp32":  # the fp32 VALU kernel (csrc/conv_f32.hip) has no grouped mode: one launch per branch
                for j, ((n_, _), (Wj, bj)) in enumerate(zip(TR_OUTS, parts)):
                    P.conv(f"{p}.flame_{n_}_pred.{bi}", View(cur, offs[2] + j * trp, trp), View(nxt, offs[2] + j * trp, trp), Wj, bj, 3, cout_store=trp, flops_macs=tr * 9 * tr)
            else:
                Wg, bg, _ = _stack(parts, pad_to=trp)
                P.conv(f"{p}.flame_transform_pred.{bi}", View(cur, offs[2], trp), View(nxt, offs[2], 4 * trp), Wg, bg, 3, cout_store=4 * trp, flops_macs=4 * tr * 9 * tr,
                       groups=(trp, trp))
            cur, cur_q = nxt, nxt_q
        # final 1x1 predictions -> fp32 prediction buffer [reg68 | cls1 | .. | shape | expr | rot6 | jaw3 | trans3 | scale1].  (r03 measured ONE
        # block-diagonal GEMM over the whole last-layer buffer instead of these three: 324 + 86 + 27 us vs 238 + 67 + 36 us at L b64 -- the zero
        # blocks cost more MFMA time than the two saved launches return; kept as three launches.)
        W, b = F[f"{p}.flame_shape_pred.{nb}"]
        P.conv(f"{p}.flame_shape_pred.{nb}", View(cur, offs[0], _r32(inters[0])), View(pred, FO, Sc), _ohwi(W, _r32(inters[0])), b, 1, act=0)
        W, b = F[f"{p}.flame_expression_pred.{nb}"]
        P.conv(f"{p}.flame_expression_pred.{nb}", View(cur, offs[1], _r32(inters[1])), View(pred, FO + Sc, Ec), _ohwi(W, _r32(inters[1])), b, 1, act=0)
        Wd = np.zeros((13, 1, 1, 4 * trp), dtype=np.float64)
        bd = np.zeros(13, dtype=np.float64)
        row = 0
        for j, (n_, o_) in enumerate(TR_OUTS):
            W, b = F[f"{p}.flame_{n_}_pred.{nb}"]
            Wd[row : row + o_, 0, 0, j * trp : j * trp + tr] = W[:, :, 0, 0]
            bd[row : row + o_] = b
            row += o_
        P.conv(f"{p}.flame_transform_pred.{nb}", View(cur, offs[2], 4 * trp), View(pred, FO + Sc + Ec, 13), Wd, bd, 1, act=0, flops_macs=13 * tr)
        P.levels.append(dict(buf=pred, h=r, w=r, pitch=pred_pitch, stride=stride))
        P.shape_c, P.expr_c = Sc, Ec
        if head_lanes:
            for op in P.ops[first_head_op:]:
                op["lane"] = lv

    P.flops = 2.0 * sum(o["macs"] for o in P.ops)
    fmt = PRECISION_FMT[precision]
    if fmt != FMT_BF16:
        for bf in P.bufs:
            if bf["is_f32"] == FMT_BF16:  # the fp32 prediction buffers stay fp32 in every mode
                bf["is_f32"] = fmt
    P.precision = precision
    return P


# substrings of the HIP kernel names that run the op program (stem / conv / pool ops): what the PMC tooling (bench.py live_traffic, tools/pmc_*.py)
# sums over.  Every op is ONE launch per lane, so a forward shows `ops x lanes` such dispatches -- the tools check that count.
NET_KERNEL_MARKERS = ("conv_igemm", "pp_kernel", "patch_kernel", "patch3_kernel", "conv1x1_stream", "conv_f32", "stem_kernel", "stem_mfma", "stem_f32", "stem_ds", "spp_pool", "ds_b2b_kernel", "ds_conv_kernel", "w_conv_kernel")


def is_net_kernel(kernel_name: str) -> bool:
    return any(m in kernel_name for m in NET_KERNEL_MARKERS)


def op_touches_fp8(P: "Program", op: dict) -> bool:
    """The op reads or writes an e4m3 link (it then runs on the ping-pong tile the library picks: no table entry applies)."""
    return op["kind"] == 1 and (P.bufs[op["in_buf"]]["is_f32"] in Q8_FMTS or P.bufs[op["out_buf"]]["is_f32"] in Q8_FMTS)


def op_algorithmic_bytes(P: "Program", op: dict, batch: int) -> Dict[str, float]:
    """HBM bytes one op moves when every tensor is read / written exactly once (SURVEY.md 8(d) "algorithmic bytes"): the input view, the
    residual view and the packed weights read, the stored channels written.  What `roofline.traffic` (PMC FETCH_SIZE / WRITE_SIZE) is
    compared with: 3x3 halos and multi-consumer tensors that miss L2 / MALL show up as measured > algorithmic."""
    kind = op["kind"]
    if kind == 3:
        return dict(read=0.0, write=0.0)
    if kind == 0:  # stem: u8 image in; 48 channels out (bf16 mode), or 48 + 16 stored zeros (parity modes: 64-channel pitch)
        ob = P.bufs[op["out_buf"]]
        return dict(read=float(batch * P.image_size * P.image_size * 3), write=float(batch * ob["h"] * ob["w"] * op["cout_store"] * FMT_BYTES[ob["is_f32"]]))
    ib = P.bufs[op["in_buf"]]
    eb_in = FMT_BYTES[ib["is_f32"]]
    if kind == 2:  # SPP pools: C channels in, 3 x C out
        px = batch * ib["h"] * ib["w"]
        return dict(read=float(px * op["cin"] * eb_in), write=float(3 * px * op["cin"] * eb_in))
    ob = P.bufs[op["out_buf"]]
    eb_out = FMT_BYTES[ob["is_f32"]]
    groups = op["cout_pad"] // op["grp_cout"] if op.get("grp_cout") else 1
    rd = batch * ib["h"] * ib["w"] * min(op["cin"], ib["pitch"] - op["in_coff"]) * groups * eb_in  # a K window wider than the pitch (stem tensor) re-reads its neighbour
    out_px = batch * ob["h"] * ob["w"]
    store = op["cout_store"] if not op["shuffle"] else op["cout_pad"] // 4
    wr = out_px * store * eb_out
    if op["res_buf"] >= 0:
        rd += out_px * store * FMT_BYTES[P.bufs[op["res_buf"]]["is_f32"]]
    rd += op["cout_pad"] * op["ksize"] ** 2 * op["cin"] * (1 if ib["is_f32"] in Q8_FMTS else 2 if ib["is_f32"] in (FMT_BF16, FMT_F16) else 4 if ib["is_f32"] == FMT_F32 else 6)
    return dict(read=float(rd), write=float(wr))


def schedule_latency(P: "Program") -> "Program":
    """Single-image latency (r06; HeadDetector.__call__ is the reference's API shape, head_detector/detector.py:97-102): at batch 1 the network is a chain of ~133
    launches of 5 - 20 us each with the chip mostly idle (profiles/r06_latency_trace_l1.txt), and the three detection heads -- 39 launches, ~440 us -- depend on nothing
    but their own pyramid level.  This pass (i) moves each head's ops right behind the op that produces its level (p3 is ready before neck3 / neck4 run), and (ii) puts
    them on the executor's lane streams with EXACT dependencies: `lane` = stream index | (1 + index of the ONE op this op waits for) << 8 (csrc/net.hip records an event
    behind that op and makes this op's stream wait for it).  Per head: [stems -> FLAME layer 0 -> shape branch] on one lane, [towers -> box / score predictions ->
    transform branch] on a second, [expression branch] on a third; heads 1 and 2 use the three side lanes while the neck goes on on the caller's stream, head 3 starts on
    it.  Same kernels, same tiles, same bits; the order of P.ops stays a valid serial order (the batch-split executor and the per-op profiler run it as such)."""
    ops = P.ops
    if any(op["kind"] == 3 for op in ops):
        return P  # already forked (head_lanes)
    heads = {lv: [op for op in ops if op["name"].startswith(f"heads.head{lv + 1}.")] for lv in range(3)}
    anchors = {0: "neck.neck2.blocks.conv3", 1: "neck.neck3.blocks.conv3", 2: "neck.neck4.blocks.conv3"}
    if not all(heads.values()) or not all(any(op["name"] == a for op in ops) for a in anchors.values()):
        return P
    new: List[dict] = []
    for op in ops:
        if op["name"].startswith("heads."):
            continue
        new.append(op)
        for lv, a in anchors.items():
            if op["name"] == a:
                new += heads[lv]
    pos = {op["name"]: i for i, op in enumerate(new)}
    for lv in range(3):
        p = f"heads.head{lv + 1}"
        la, lb, lc = (1, 2, 3) if lv < 2 else (0, 1, 2)
        stem, f0, towers = f"{p}.pose_stem|bbox_stem", f"{p}.flame_*_pred.0", f"{p}.cls_convs|reg_convs"
        for op in heads[lv]:
            n = op["name"]
            if n in (stem, f0):
                lane, dep = la, (anchors[lv] if n == stem else None)
            elif n in (towers, f"{p}.reg_pred|cls_pred"):
                lane, dep = lb, (stem if n == towers else None)
            else:
                br = "shape" if ".flame_shape_pred." in n else "expr" if ".flame_expression_pred." in n else "tr"
                lane = {"shape": la, "expr": lc, "tr": lb}[br]
                # the first layer of a branch that does not share FLAME layer 0's lane waits for it (the ungrouped fp32 program has four transform ops per layer: each first one)
                first = n.rsplit(".", 1)[-1] == "1"
                dep = f0 if (first and lane != la) else None
            op["lane"] = lane | (((pos[dep] + 1) << 8) if dep is not None and (new[pos[dep]].get("lane", 0) & 0xFF) != lane else 0)
    # the side lanes only ever wait for the caller's stream or for lane `la`: the wait-for relation among them is acyclic.  (It has to be: under stream capture the HIP
    # runtime links a non-origin stream to the stream of EVERY event it waits for, and two side streams waiting for each other's events make hipStreamEndCapture recurse
    # without end -- measured r06, ROCm 7.0's libamdhip64; csrc/net.hip refuses to capture such a program.)
    P.ops = new
    return P


def b2b_pairs(P: "Program") -> List[int]:
    """Indices i of the ops the executor runs with op i + 1 as ONE back-to-back-GEMM launch (csrc/net.hip, vgh_net_create; r06): a plain bf16 conv with all of its 96
    output channels in one tile whose whole output tensor is read by exactly one op, the next one, a plain 1x1 / stride-1 bf16 conv with 128, 192 or 256 output channels.
    The same predicate as the library's (the engine checks the counts agree): PMC tools and the algorithmic-bytes accounting use it."""
    out = []
    for i in range(len(P.ops) - 1):
        a, b = P.ops[i], P.ops[i + 1]
        if a["kind"] != 1 or b["kind"] != 1:
            continue
        ab, ai, bo = P.bufs[a["out_buf"]], P.bufs[a["in_buf"]], P.bufs[b["out_buf"]]
        ok = (ab["is_f32"] == FMT_BF16 and ai["is_f32"] == FMT_BF16 and bo["is_f32"] == FMT_BF16 and a["ksize"] in (1, 3) and a["cout_pad"] == 96 and b["cout_pad"] in (128, 192, 256)
              and a["cout_store"] == a["cout_pad"] and a["out_coff"] == 0 and a["out_split"] >= a["cout_pad"] and ab["pitch"] == a["cout_pad"] and a["res_buf"] < 0 and not a["shuffle"]
              and not a.get("grp_cout") and a["act"] != 2 and b["ksize"] == 1 and b["stride"] == 1 and b["in_buf"] == a["out_buf"] and b["in_coff"] == 0 and b["cin"] == a["cout_pad"]
              and b["res_buf"] < 0 and not b["shuffle"] and not b.get("grp_cout") and b["act"] != 2 and b["out_coff"] % 8 == 0 and b["out_coff2"] % 8 == 0 and b["out_split"] % 8 == 0
              and b["cout_store"] % 8 == 0 and bo["pitch"] % 8 == 0 and (not out or out[-1] != i - 1))
        for j, o in enumerate(P.ops):
            if ok and j not in (i, i + 1) and o["kind"] in (1, 2) and a["out_buf"] in (o["in_buf"], o["out_buf"], o.get("res_buf", -1)):
                ok = False
        if ok:
            out.append(i)
    return out


def program_algorithmic_bytes(P: "Program", batch: int, fused_stem: Optional[bool] = None, b2b: bool = True) -> Dict[str, float]:
    """fused_stem (engine.stem_fused: the default for u8 images since late r06 -- the stem conv runs inside the stage-1 pair's launch): the stem tensor is neither written nor read.  b2b (r06, the executor's default): the tensor between
    the two convs of a back-to-back pair is neither written nor read -- the pair is ONE op of the engine's layer-by-layer accounting."""
    fused_stem = bool(fused_stem)
    tot = dict(read=0.0, write=0.0)
    pairs = set(b2b_pairs(P)) if b2b else set()
    for i, op in enumerate(P.ops):
        b = op_algorithmic_bytes(P, op, batch)
        if i in pairs:
            b["write"] = 0.0
        if i - 1 in pairs:
            ib = P.bufs[op["in_buf"]]
            b["read"] -= batch * ib["h"] * ib["w"] * op["cin"] * FMT_BYTES[ib["is_f32"]]
        if fused_stem and op["kind"] == 0:
            b["write"] = 0.0
        if fused_stem and i > 0 and P.ops[i - 1]["kind"] == 0 and op["kind"] == 1 and op["in_buf"] == P.ops[i - 1]["out_buf"]:
            ib = P.bufs[op["in_buf"]]
            b["read"] -= batch * ib["h"] * ib["w"] * min(op["cin"], ib["pitch"] - op["in_coff"]) * FMT_BYTES[ib["is_f32"]]
        tot["read"] += b["read"]
        tot["write"] += b["write"]
    return tot
