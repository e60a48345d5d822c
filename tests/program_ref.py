"""TEST INFRASTRUCTURE: executes a head_detector_amd.arch.Program with plain torch CPU ops (F.conv2d / max_pool2d),
optionally emulating the engine's bf16 storage (inputs, weights and every stored activation rounded to bf16,
fp32 accumulation).  Used (a) on CPU to prove that fold + fusion + concat-by-offset lowering reproduces the
unfused oracle network, (b) on the GPU box as the per-op checker of the HIP kernels on the engine's own inputs."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def rb(t: torch.Tensor, bf16: bool) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32) if bf16 else t


def alloc(P, B: int):
    return [torch.zeros(B, bf["h"], bf["w"], bf["pitch"], dtype=torch.float32) for bf in P.bufs]


def run_op(P, op, bufs, image, bf16: bool, w_all=None, b_all=None, f64: bool = False):
    """Executes one op in place on `bufs` (list of [B,h,w,pitch] float tensors).  f64: the arithmetic of the op in float64 (inputs and
    stored result stay float32) -- the reference for the parity modes, whose own error is then not mixed with torch's fp32 summation order."""
    up = (lambda t: t.double()) if f64 else (lambda t: t)
    if w_all is None:
        w_all, b_all = P.arrays()
    kind = op["kind"]
    if kind == 3:  # FORK: scheduling only
        return
    if kind == 0:  # stem: exact fp32 conv on the image, ReLU, stored as bf16
        W = torch.from_numpy(w_all[op["w_off"] : op["w_off"] + 48 * 27].reshape(48, 3, 3, 3)).permute(0, 3, 1, 2).contiguous()
        b = torch.from_numpy(b_all[op["b_off"] : op["b_off"] + 48])
        x = image if image.dtype == torch.float32 else image.permute(0, 3, 1, 2).float() / 255.0
        y = torch.relu(F.conv2d(up(x), up(W), up(b), stride=2, padding=1)).permute(0, 2, 3, 1).float()
        out = bufs[op["out_buf"]]
        out[..., op["out_coff"] : op["out_coff"] + 48] = rb(y, bf16)
        if op["cout_store"] > 48:  # 64-channel pitch (parity modes): 16 stored zeros; the bf16 mode stores the 48 channels at a 48-channel pitch
            out[..., op["out_coff"] + 48 : op["out_coff"] + 64] = 0
        return
    if kind == 2:  # SPP pools
        buf = bufs[op["in_buf"]]
        C, c0 = op["cin"], op["in_coff"]
        x = buf[..., c0 : c0 + C].permute(0, 3, 1, 2)
        for i, k in enumerate((5, 9, 13)):
            buf[..., c0 + (i + 1) * C : c0 + (i + 2) * C] = F.max_pool2d(x, k, 1, k // 2).permute(0, 2, 3, 1)
        return
    k, cin, rp = op["ksize"], op["cin"], op["cout_pad"]
    W = torch.from_numpy(w_all[op["w_off"] : op["w_off"] + rp * k * k * cin].reshape(rp, k, k, cin)).permute(0, 3, 1, 2).contiguous()
    b = torch.from_numpy(b_all[op["b_off"] : op["b_off"] + rp])
    gc = op.get("grp_cout", 0)
    if gc:  # grouped conv: cout group g reads its own cin-channel window
        ys = []
        for g in range(rp // gc):
            c0 = op["in_coff"] + g * op["grp_in_stride"]
            xg = bufs[op["in_buf"]][..., c0 : c0 + cin].permute(0, 3, 1, 2)
            ys.append(F.conv2d(up(rb(xg, bf16)), up(rb(W[g * gc : (g + 1) * gc], bf16)), None, stride=op["stride"], padding=k // 2))
        y = torch.cat(ys, 1) + up(b)[None, :, None, None]
    else:
        x = bufs[op["in_buf"]][..., op["in_coff"] : op["in_coff"] + cin]
        if x.shape[-1] < cin:
            # a K window wider than the pitch (the 48-channel stem tensor read as 64 channels): the engine's window runs on into the next pixel and meets
            # all-zero weight columns (checked here, and by vgh_net_create); zeros stand in for those finite values
            assert float(W[:, x.shape[-1] :].abs().max()) == 0.0
            x = torch.cat([x, torch.zeros(*x.shape[:-1], cin - x.shape[-1], dtype=x.dtype)], -1)
        x = x.permute(0, 3, 1, 2)
        y = F.conv2d(up(rb(x, bf16)), up(rb(W, bf16)), None, stride=op["stride"], padding=k // 2) + up(b)[None, :, None, None]
    if op["act"] == 1:
        y = torch.relu(y)
    elif op["act"] == 2:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1)  # [B,ho,wo,rp]
    out = bufs[op["out_buf"]]
    is_f32 = P.bufs[op["out_buf"]]["is_f32"]
    if op["shuffle"]:
        C = rp // 4
        B_, h, w, _ = y.shape
        z = torch.zeros(B_, 2 * h, 2 * w, C, dtype=y.dtype)
        for d in range(4):
            z[:, d // 2 :: 2, d % 2 :: 2, :] = y[..., d * C : (d + 1) * C]
        y = z
    if op["res_buf"] >= 0:
        n = min(y.shape[-1], op["cout_store"])
        r = bufs[op["res_buf"]][..., op["res_coff"] : op["res_coff"] + n]
        y = y.clone()
        y[..., :n] = y[..., :n] + (float(np.float32(op["alpha"])) * up(r) if f64 else np.float32(op["alpha"]) * r)
    store = op["cout_store"] if not op["shuffle"] else rp // 4
    split = min(op["out_split"], store)
    y = y.float()
    y = y if is_f32 else rb(y, bf16)
    out[..., op["out_coff"] : op["out_coff"] + split] = y[..., :split]
    if store > split:
        out[..., op["out_coff2"] : op["out_coff2"] + store - split] = y[..., split:store]


def run_program(P, image, bf16: bool):
    bufs = alloc(P, image.shape[0])
    w_all, b_all = P.arrays()
    for op in P.ops:
        run_op(P, op, bufs, image, bf16, w_all, b_all)
    return bufs


def head_outputs(P, bufs):
    """per level: (reg [B,68,H,W], cls [B,1,H,W], raw branch dict) as NCHW torch tensors (oracle format)."""
    out = []
    S, E = P.shape_c, P.expr_c
    for lv in P.levels:
        t = bufs[lv["buf"]].permute(0, 3, 1, 2)
        o = 72  # VGH_PRED_FLAME_OFF: [reg 68 | cls 1 | 3 unused | shape | expr | rot jaw trans scale]
        raw = dict(shape=t[:, o : o + S], expr=t[:, o + S : o + S + E], rot=t[:, o + S + E : o + S + E + 6], jaw=t[:, o + S + E + 6 : o + S + E + 9],
                   trans=t[:, o + S + E + 9 : o + S + E + 12], scale=t[:, o + S + E + 12 : o + S + E + 13])
        out.append((t[:, :68], t[:, 68:69], raw))
    return out
