"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  "parity unpinned" -- see below.

torch-CPU fp32 restatement of the VGGHeads network that lives inside the released
TorchScript blob (head_detector/detector.py:25-30,58-59).  The *definition* is

  yolo_head_training/yolo_head/yolo_heads.py:89-112            YoloHeads(CustomizableDetector)
  yolo_head_training/configs/arch_params/yolo_heads_{m,l}_arch_params.yaml:4-141
  yolo_head_training/yolo_head/yolo_head_dfl_head.py:17-186    YoloHeadsDFLHead
  yolo_head_training/yolo_head/yolo_head_ndfl_heads.py:117-175 YoloHeadsNDFLHeads.forward
  yolo_head_training/yolo_head/yolo_heads.py:44-86             VGGHeadDecodingModule

with every backbone/neck block (NStageBackbone, YoloNASStem/Stage/UpStage/DownStage,
YoloNASCSPLayer, YoloNASBottleneck, SPP, QARepVGGBlock, Conv, ConvBNReLU,
YoloNASPANNeckWithC2) defined in super_gradients>=3.7 (yolo_head_training/requirements.txt:1),
which is NOT vendored in /root/reference and not installed in this image.  Their published
forward semantics are restated here in UNFUSED form (separate 3x3/1x1/identity/post-BN
branches, separate conv1/conv2, real torch.cat) so that the product's weight folding and
launch fusion are checked against something structurally different from itself.

Parameter names follow the super_gradients module tree so a real ``state_dict`` of the
released model would load with ``strict=True`` modulo the prefix.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .postproc_oracle import assemble_flame_channels, decoding_topk, ndfl_decode

BN_EPS = 1e-6  # arch yaml :139 (applied to every BatchNorm by CustomizableDetector)

VARIANTS: Dict[str, dict] = {
    # yolo_heads_l_arch_params.yaml
    "l": dict(
        stem=48,
        stages=[(96, 2, 96, True), (192, 3, 128, True), (384, 5, 256, True), (768, 2, 512, True)],
        spp_out=768,
        neck=[(192, 4, 128), (96, 4, 128), (192, 4, 128), (384, 4, 256)],
        head=dict(bbox=(128, 256, 512), flame=256, blocks=3, shape_inter=256, expr_inter=128, shape_out=128, expr_out=64, tr_inter=32, width_mult=1.0),
    ),
    # yolo_heads_m_arch_params.yaml
    "m": dict(
        stem=48,
        stages=[(96, 2, 64, True), (192, 3, 128, True), (384, 5, 256, True), (768, 2, 384, False)],
        spp_out=768,
        neck=[(192, 2, 192), (96, 3, 64), (192, 2, 192), (384, 3, 256)],
        head=dict(bbox=(256, 256, 256), flame=256, blocks=2, shape_inter=128, expr_inter=64, shape_out=64, expr_out=32, tr_inter=16, width_mult=0.75),
    ),
}


def width_multiplier(original: int, factor: float, divisor: int = None) -> int:
    """super_gradients.modules.utils.width_multiplier."""
    import math

    if divisor is None:
        return int(original * factor)
    return math.ceil(int(original * factor) / divisor) * divisor


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS)


class Conv(nn.Module):
    """SG yolo_nas ``Conv``: conv(bias=False, pad=k//2) + BN + ReLU."""

    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False)
        self.bn = _bn(cout)

    def forward(self, x):
        return torch.relu(self.bn(self.conv(x)))


class ConvBNReLU(nn.Module):
    """SG ``ConvBNReLU``: .seq = [conv, bn, act]."""

    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        self.seq = nn.Sequential()
        self.seq.add_module("conv", nn.Conv2d(cin, cout, k, s, p, bias=False))
        self.seq.add_module("bn", _bn(cout))

    def forward(self, x):
        return torch.relu(self.seq(x))


class QARepVGGBlock(nn.Module):
    """SG ``QARepVGGBlock`` unfused forward: relu(post_bn(bn(conv3x3(x)) + alpha*conv1x1_bias(x) [+ x]))."""

    def __init__(self, cin, cout, stride=1, use_residual_connection=True, use_alpha=False):
        super().__init__()
        self.branch_3x3 = nn.Sequential()
        self.branch_3x3.add_module("conv", nn.Conv2d(cin, cout, 3, stride, 1, bias=False))
        self.branch_3x3.add_module("bn", _bn(cout))
        self.branch_1x1 = nn.Conv2d(cin, cout, 1, stride, 0, bias=True)
        self.has_identity = bool(use_residual_connection)
        if use_residual_connection:
            assert cin == cout and stride == 1
        if use_alpha:
            self.alpha = nn.Parameter(torch.tensor([1.0]))
        else:
            self.alpha = 1.0
        self.post_bn = _bn(cout)

    def forward(self, x):
        y = self.branch_3x3(x) + self.alpha * self.branch_1x1(x)
        if self.has_identity:
            y = y + x
        return torch.relu(self.post_bn(y))


class Bottleneck(nn.Module):
    """SG ``YoloNASBottleneck``: alpha * x + cv2(cv1(x)) (shortcut, use_alpha=True)."""

    def __init__(self, c, block):
        super().__init__()
        self.cv1 = block(c, c)
        self.cv2 = block(c, c)
        self.alpha = nn.Parameter(torch.tensor([1.0]))

    def forward(self, x):
        return self.alpha * x + self.cv2(self.cv1(x))


class CSPLayer(nn.Module):
    """SG ``YoloNASCSPLayer``: conv3(cat([*bottlenecks(conv1(x)), conv2(x)]))."""

    def __init__(self, cin, cout, n, block, hidden, concat_intermediates):
        super().__init__()
        self.conv1 = Conv(cin, hidden, 1, 1)
        self.conv2 = Conv(cin, hidden, 1, 1)
        self.conv3 = Conv(hidden * (2 + concat_intermediates * n), cout, 1, 1)
        self.bottlenecks = nn.ModuleList([Bottleneck(hidden, block) for _ in range(n)])
        self.concat_intermediates = concat_intermediates

    def forward(self, x):
        x1 = self.conv1(x)
        outs = [x1]
        for b in self.bottlenecks:
            x1 = b(x1)
            outs.append(x1)
        if not self.concat_intermediates:
            outs = outs[-1:]
        x2 = self.conv2(x)
        return self.conv3(torch.cat((*outs, x2), dim=1))


def _rep(cin, cout):
    return QARepVGGBlock(cin, cout)  # default: residual connection when legal (cin == cout, stride 1)


def _conv3(cin, cout):
    return Conv(cin, cout, 3, 1)


class Stem(nn.Module):
    def __init__(self, cout):
        super().__init__()
        self.conv = QARepVGGBlock(3, cout, stride=2, use_residual_connection=False)

    def forward(self, x):
        return self.conv(x)


class Stage(nn.Module):
    def __init__(self, cin, cout, n, hidden, ci):
        super().__init__()
        self.downsample = QARepVGGBlock(cin, cout, stride=2, use_residual_connection=False)
        self.blocks = CSPLayer(cout, cout, n, _rep, hidden, ci)

    def forward(self, x):
        return self.blocks(self.downsample(x))


class SPP(nn.Module):
    def __init__(self, cin, cout, k=(5, 9, 13)):
        super().__init__()
        hidden = cin // 2
        self.cv1 = Conv(cin, hidden, 1, 1)
        self.cv2 = Conv(hidden * (len(k) + 1), cout, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def forward(self, x):
        x = self.cv1(x)
        return self.cv2(torch.cat([x] + [m(x) for m in self.m], 1))


class Backbone(nn.Module):
    def __init__(self, v):
        super().__init__()
        self.stem = Stem(v["stem"])
        c = v["stem"]
        for i, (co, n, hid, ci) in enumerate(v["stages"]):
            setattr(self, f"stage{i + 1}", Stage(c, co, n, hid, ci))
            c = co
        self.context_module = SPP(c, v["spp_out"])

    def forward(self, x):
        x = self.stem(x)
        c2 = self.stage1(x)
        c3 = self.stage2(c2)
        c4 = self.stage3(c3)
        c5 = self.context_module(self.stage4(c4))
        return c2, c3, c4, c5


class UpStage(nn.Module):
    """SG ``YoloNASUpStage`` with 3 inputs and reduce_channels=True."""

    def __init__(self, cin, skip1, skip2, cout, n, hidden):
        super().__init__()
        self.reduce_skip1 = Conv(skip1, cout, 1, 1)
        self.reduce_skip2 = Conv(skip2, cout, 1, 1)
        self.conv = Conv(cin, cout, 1, 1)
        self.upsample = nn.ConvTranspose2d(cout, cout, kernel_size=2, stride=2)
        self.downsample = Conv(cout, cout, 3, 2)
        self.reduce_after_concat = Conv(3 * cout, cout, 1, 1)
        self.blocks = CSPLayer(cout, cout, n, _rep, hidden, False)

    def forward(self, x, s1, s2):
        s1, s2 = self.reduce_skip1(s1), self.reduce_skip2(s2)
        x_inter = self.conv(x)
        x = torch.cat([self.upsample(x_inter), s1, self.downsample(s2)], 1)
        return x_inter, self.blocks(self.reduce_after_concat(x))


class DownStage(nn.Module):
    """SG ``YoloNASDownStage``."""

    def __init__(self, cin, skip, cout, n, hidden):
        super().__init__()
        self.conv = Conv(cin, cout // 2, 3, 2)
        self.blocks = CSPLayer(cout // 2 + skip, cout, n, _conv3, hidden, False)

    def forward(self, x, skip):
        return self.blocks(torch.cat([self.conv(x), skip], 1))


class Neck(nn.Module):
    """SG ``YoloNASPANNeckWithC2`` (arch yaml :49-88)."""

    def __init__(self, v, cs):
        super().__init__()
        c2, c3, c4, c5 = cs
        (o1, n1, h1), (o2, n2, h2), (o3, n3, h3), (o4, n4, h4) = v["neck"]
        self.neck1 = UpStage(c5, c4, c3, o1, n1, h1)
        self.neck2 = UpStage(o1, c3, c2, o2, n2, h2)
        self.neck3 = DownStage(o2, o2, o3, n3, h3)
        self.neck4 = DownStage(o3, o1, o4, n4, h4)

    def forward(self, c2, c3, c4, c5):
        x_n1_inter, x = self.neck1(c5, c4, c3)
        x_n2_inter, p3 = self.neck2(x, c3, c2)
        p4 = self.neck3(p3, x_n2_inter)
        p5 = self.neck4(p4, x_n1_inter)
        return p3, p4, p5


class DFLHead(nn.Module):
    """YoloHeadsDFLHead (yolo_head_dfl_head.py:23-186), shared_stem=False, first_conv_group_size=0."""

    def __init__(self, cin, h, bbox_inter):
        super().__init__()
        bbox = width_multiplier(bbox_inter, h["width_mult"], 8)
        fl = width_multiplier(h["flame"], h["width_mult"], 8)
        self.pose_stem = ConvBNReLU(cin, fl, 1, 1, 0)
        self.bbox_stem = ConvBNReLU(cin, bbox, 1, 1, 0)
        self.cls_convs = nn.Sequential(ConvBNReLU(bbox, bbox, 3, 1, 1))
        self.reg_convs = nn.Sequential(ConvBNReLU(bbox, bbox, 3, 1, 1))
        self.reg_pred = nn.Conv2d(bbox, 4 * 17, 1, 1, 0)
        self.cls_pred = nn.Conv2d(bbox, 1, 1, 1, 0)

        def branch(inter, out):
            layers, c = [], fl
            for _ in range(h["blocks"]):
                layers.append(QARepVGGBlock(c, inter, use_residual_connection=False, use_alpha=True))
                c = inter
            layers.append(nn.Conv2d(inter, out, 1, 1, 0))
            return nn.Sequential(*layers)

        self.flame_shape_pred = branch(h["shape_inter"], h["shape_out"])
        self.flame_expression_pred = branch(h["expr_inter"], h["expr_out"])
        self.flame_rotation_pred = branch(h["tr_inter"], 6)
        self.flame_jaw_pred = branch(h["tr_inter"], 3)
        self.flame_scale_pred = branch(h["tr_inter"], 1)
        self.flame_translation_pred = branch(h["tr_inter"], 3)

    def forward(self, x):
        pose = self.pose_stem(x)
        bb = self.bbox_stem(x)
        cls = self.cls_pred(self.cls_convs(bb))
        reg = self.reg_pred(self.reg_convs(bb))
        raw = dict(
            shape=self.flame_shape_pred(pose),
            expr=self.flame_expression_pred(pose),
            rot=self.flame_rotation_pred(pose),
            jaw=self.flame_jaw_pred(pose),
            trans=self.flame_translation_pred(pose),
            scale=self.flame_scale_pred(pose),
        )
        flame = assemble_flame_channels(raw["shape"], raw["expr"], raw["rot"], raw["jaw"], raw["trans"], raw["scale"])
        return reg, cls, flame, raw


class Heads(nn.Module):
    def __init__(self, v, cs):
        super().__init__()
        for i, c in enumerate(cs):
            setattr(self, f"head{i + 1}", DFLHead(c, v["head"], v["head"]["bbox"][i]))

    def forward(self, feats):
        return [getattr(self, f"head{i + 1}")(f) for i, f in enumerate(feats)]


class YoloHeadsOracle(nn.Module):
    """backbone -> neck -> heads -> NDFL decode -> top-k(1000) decoding module."""

    def __init__(self, variant: str):
        super().__init__()
        v = VARIANTS[variant]
        self.variant = variant
        self.backbone = Backbone(v)
        cs = [s[0] for s in v["stages"][:3]] + [v["spp_out"]]
        self.neck = Neck(v, cs)
        self.heads = Heads(v, [v["neck"][1][0], v["neck"][2][0], v["neck"][3][0]])
        self.eval()

    @torch.no_grad()
    def features(self, x):
        return self.neck(*self.backbone(x))

    @torch.no_grad()
    def raw_heads(self, x):
        """per level: (reg [B,68,H,W], cls [B,1,H,W], flame [B,413,H,W], raw branch outputs)."""
        return self.heads(self.features(x))

    @torch.no_grad()
    def dense(self, x):
        """Network output before the decoding module: boxes [B,A,4], scores [B,A,1], flame [B,A,413]."""
        lv = self.raw_heads(x)
        return ndfl_decode([(r, c, f) for r, c, f, _ in lv])

    @torch.no_grad()
    def forward(self, x, k: int = 1000):
        """What HeadDetector._process returns (detector.py:58-59): [B,k,4], [B,k,1], [B,k,413]."""
        b, s, f = self.dense(x)
        bb, ss, ff, _ = decoding_topk(b, s, f, k)
        return bb, ss, ff


def conv_flops(model: YoloHeadsOracle, size: int = 640) -> float:
    """2 * MACs of every conv in FUSED-layer accounting: a QARepVGG block counts as
    one 3x3 conv (its 1x1 / identity / BN branches fold away)."""
    total = 0.0
    handles = []

    def hook(mod, inp, out):
        nonlocal total
        name = names[mod]
        if name.endswith("branch_1x1"):
            return
        if isinstance(mod, nn.ConvTranspose2d):
            total += 2.0 * inp[0].shape[2] * inp[0].shape[3] * mod.in_channels * mod.out_channels * 4
        else:
            kh, kw = mod.kernel_size
            total += 2.0 * out.shape[2] * out.shape[3] * mod.out_channels * mod.in_channels * kh * kw

    names = {m: n for n, m in model.named_modules()}
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            handles.append(m.register_forward_hook(hook))
    model.raw_heads(torch.zeros(1, 3, size, size))
    for h in handles:
        h.remove()
    return total
