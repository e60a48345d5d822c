"""GPU (-m gpu): SURVEY.md 8(f) N1 on archive shapes the release may have (head_detector/detector.py:25-30 loads a TorchScript .trcd,
head_detector/flame.py:18-24,75-95 a chumpy / scipy-sparse pickle): archive -> detector.load_weights -> manifest check -> arch.build_program
-> pack.write_pack -> vgh_create -> vgh_ctx_detect must equal the engine built straight from the state dict."""
import ctypes as C
import pickle
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def module_from_state_dict(sd):
    """A bare nn.Module tree whose state_dict() is `sd` (dotted names -> nested modules, leaves as buffers): what torch.jit.script needs
    to write a TorchScript archive with exactly these keys."""
    root = torch.nn.Module()
    for k, v in sd.items():
        m = root
        parts = k.split(".")
        for p in parts[:-1]:
            if not hasattr(m, p):
                m.add_module(p, torch.nn.Module())
            m = getattr(m, p)
        m.register_buffer(parts[-1], torch.as_tensor(v).clone())
    return root


def fused_variants(variant, sd):
    """The same weights as an archive exported AFTER QARepVGG fusion: full fusion = rbr_reparam only; partial = rbr_reparam
    (branches + alpha + identity) with post_bn kept as BatchNorm."""
    from head_detector_amd import arch

    F = arch.fold_state_dict(variant, sd)
    full, partial = {}, {}
    qarep = {sp.name: sp for sp in arch.layer_specs(variant) if sp.kind == "qarep"}
    for k, v in sd.items():
        blk = next((n for n in qarep if k.startswith(n + ".")), None)
        if blk is None:
            full[k] = partial[k] = v
        elif k.startswith(blk + ".post_bn."):
            partial[k] = v
    for n, sp in qarep.items():
        W, b = F[n]
        full[f"{n}.rbr_reparam.weight"], full[f"{n}.rbr_reparam.bias"] = W.astype(np.float32), b.astype(np.float32)
        # partial fusion: undo post_bn on the fully folded conv (exact up to fp32 rounding of the stored tensors)
        s_, t_ = arch._bn_affine(sd, f"{n}.post_bn")
        partial[f"{n}.rbr_reparam.weight"] = (W / s_[:, None, None, None]).astype(np.float32)
        partial[f"{n}.rbr_reparam.bias"] = ((b - t_) / s_).astype(np.float32)
    return full, partial


def flame_pickle(path, m):
    """generic_model.pkl as FLAME ships it: protocol-2 pickle, chumpy-wrapped arrays, scipy-sparse J_regressor."""
    import scipy.sparse as sp

    mod, sub = types.ModuleType("chumpy"), types.ModuleType("chumpy.ch")

    class Ch:
        def __init__(self, x):
            self.x = x

    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    sub.Ch = Ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
    try:
        blob = dict(m)
        blob["v_template"], blob["shapedirs"] = Ch(m["v_template"]), Ch(m["shapedirs"])
        blob["J_regressor"] = sp.csc_matrix(m["J_regressor"])
        with open(path, "wb") as f:
            pickle.dump(blob, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]


def test_release_shaped_archives_through_the_pack(gpu_lib, flame_model, tmp_path):
    from head_detector_amd import _lib, arch, pack
    from head_detector_amd.detector import load_weights, weight_manifest_diff
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer, get_flame_model

    variant, S, B = "vgg_heads_m", 256, 2
    sd = arch.random_state_dict(variant, 17)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(4)).to(_dev())
    unpad = torch.tensor([[2.0, 3.0, 1.25], [0.0, 8.0, 0.75]], device=_dev())
    # the FLAME asset through its pickle form
    pkl = str(tmp_path / "generic_model.pkl")
    flame_pickle(pkl, flame_model)
    fm = get_flame_model(pkl)
    fl = FLAMELayer(model=fm, device=_dev(), max_heads=B * 100)
    eng = VGHeadsEngine(variant, state_dict=sd, image_size=S, max_batch=B, use_tuning=False)
    _, sc, _ = eng.model(x)
    conf = float(sc[:, 5, 0].min())
    ref = eng.detect(x, confidence_threshold=conf, flame=fl, unpad=unpad)
    n_ref = ref.num_heads
    assert n_ref >= B
    ref_dense = eng.scores_all[:B].clone()
    eng.close()

    full, partial = fused_variants(variant, sd)
    archives = {}
    # (i) TorchScript archive with the unfused super_gradients keys under `model.` (ConvertableCompletePipelineModel), plus an unused rbr_reparam conv
    with_model = {f"model.{k}": v for k, v in sd.items()}
    with_model["model.backbone.stem.conv.rbr_reparam.weight"] = np.zeros((48, 3, 3, 3), np.float32)
    with_model["model.backbone.stem.conv.rbr_reparam.bias"] = np.zeros(48, np.float32)
    p = str(tmp_path / "unfused.trcd")
    torch.jit.script(module_from_state_dict(with_model)).save(p)
    archives["torchscript unfused"] = (p, True)
    # a training checkpoint: ema_net wins over net
    p = str(tmp_path / "ckpt.pth")
    torch.save({"net": {k: torch.from_numpy(v) * 0 for k, v in sd.items()}, "ema_net": {k: torch.from_numpy(v) for k, v in sd.items()}}, p)
    archives["checkpoint ema_net"] = (p, True)
    # (ii) exported after fusion: rbr_reparam only (full) / rbr_reparam + post_bn (partial)
    p = str(tmp_path / "fused.trcd")
    torch.jit.script(module_from_state_dict({f"model.{k}": v for k, v in full.items()})).save(p)
    archives["torchscript fully fused"] = (p, True)
    p = str(tmp_path / "partial.trcd")
    torch.jit.script(module_from_state_dict({f"model.{k}": v for k, v in partial.items()})).save(p)
    archives["torchscript partially fused"] = (p, False)  # post_bn is re-applied to fp32-rounded tensors: equal to round-off, not bit for bit

    kk, V = 100, fl.num_vertices
    for what, (path, exact) in archives.items():
        got = load_weights(path)
        diff = weight_manifest_diff(variant, got)
        assert not any(diff.values()), (what, {k: v[:4] for k, v in diff.items()})
        P = arch.build_program(variant, got, S)
        pk = str(tmp_path / "a.vghpack")
        pack.write_pack(pk, P, fm, {}, B)
        h = C.c_void_p()
        cfg = _lib.Config(device=torch.cuda.current_device(), pack_path=pk.encode(), max_batch=B)
        _lib.check(gpu_lib.vgh_create(C.byref(cfg), C.byref(h)))
        f32 = dict(dtype=torch.float32, device=_dev())
        ob, os_, of = torch.zeros(B, kk, 4, **f32), torch.zeros(B, kk, **f32), torch.zeros(B, kk, 413, **f32)
        oc, nh, hi = torch.zeros(B, dtype=torch.int32, device=_dev()), torch.zeros(1, dtype=torch.int32, device=_dev()), torch.zeros(B * kk, dtype=torch.int32, device=_dev())
        proj, rpy = torch.zeros(B * kk, V, 3, **f32), torch.zeros(B * kk, 3, **f32)
        o = _lib.DetectOut(boxes_dev=ob.data_ptr(), scores_dev=os_.data_ptr(), flame_dev=of.data_ptr(), counts_dev=oc.data_ptr(), n_heads_dev=nh.data_ptr(),
                           head_image_dev=hi.data_ptr(), head_capacity=B * kk, unpad_dev=unpad.data_ptr(), verts_dev=None, rot_dev=None, rpy_dev=rpy.data_ptr(), proj_dev=proj.data_ptr())
        st = torch.cuda.current_stream().cuda_stream
        rc = gpu_lib.vgh_ctx_detect(h, x.data_ptr(), _lib.VGH_IMG_U8_NHWC, B, conf, 0.5, C.byref(o), st)
        assert rc == 0, gpu_lib.vgh_ctx_last_error(h)
        _lib.check(gpu_lib.vgh_ctx_join(h, st))
        torch.cuda.synchronize()
        if exact:
            assert torch.equal(oc, ref.counts) and int(nh) == n_ref, what
            for b in range(B):
                n = int(oc[b])
                assert torch.equal(ob[b, :n], ref.boxes[b, :n]) and torch.equal(os_[b, :n], ref.scores[b, :n]) and torch.equal(of[b, :n], ref.flame_params[b, :n]), what
            assert torch.equal(proj[:n_ref], ref.vertices_3d) and torch.equal(rpy[:n_ref], ref.head_pose), what
        else:
            # same network up to fp32 round-off of the re-derived tensors, seen through bf16 activations: the dense scores agree to bf16 noise
            from head_detector_amd.engine import _alias

            dense = _alias(gpu_lib.vgh_detector_scratch(gpu_lib.vgh_ctx_detector(h), _lib.SCRATCH_SCORES_ALL), tuple(ref_dense.shape), "<f4", _dev()).clone()
            assert float((dense - ref_dense).abs().max()) < 2e-3, what
        gpu_lib.vgh_destroy(h)


def _iou(a, b):
    x1, y1 = torch.maximum(a[..., 0], b[..., 0]), torch.maximum(a[..., 1], b[..., 1])
    x2, y2 = torch.minimum(a[..., 2], b[..., 2]), torch.minimum(a[..., 3], b[..., 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    return inter / ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter)


@pytest.mark.parametrize("variant,okey", [("vgg_heads_m", "m"), ("vgg_heads_l", "l")])
def test_archives_against_the_unfused_oracle_at_north_star_tolerances(gpu_lib, flame_model, tmp_path, variant, okey):
    """N1 as an ORACLE test: the network the library builds FROM THE ARCHIVE (load_weights -> build_program(fp16x3) -> pack -> vgh_create) against
    oracle/net_oracle.YoloHeadsOracle loaded from the SAME archive's state dict -- strictly (every key of the unfused archive must be an oracle key and vice versa;
    head_detector/detector.py:25-30 loads the blob, the oracle is the unfused super_gradients module graph).  Bars: north_star's IoU >= 0.999, scores 1e-5, parameters 1e-4."""
    from head_detector_amd import _lib, arch, pack
    from head_detector_amd.detector import load_weights, weight_manifest_diff
    from head_detector_amd.engine import _alias
    from oracle import net_oracle

    S, B = 256, 2
    sd = arch.random_state_dict(variant, 23)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(6))
    full, _ = fused_variants(variant, sd)
    archives = {}
    p = str(tmp_path / "unfused.trcd")
    torch.jit.script(module_from_state_dict({f"model.{k}": v for k, v in sd.items()})).save(p)
    archives["torchscript unfused"] = (p, True)
    p = str(tmp_path / "ckpt.pth")
    torch.save({"net": {k: torch.from_numpy(v) * 0 for k, v in sd.items()}, "ema_net": {k: torch.from_numpy(v) for k, v in sd.items()}}, p)
    archives["checkpoint ema_net"] = (p, True)
    p = str(tmp_path / "fused.trcd")
    torch.jit.script(module_from_state_dict({f"model.{k}": v for k, v in full.items()})).save(p)
    archives["torchscript fully fused"] = (p, False)  # its keys are rbr_reparam.*: the oracle (unfused graph) is loaded from the archive's unfused twin below
    # N4 (r05): the same network as ONNX initializers (head_detector_amd/onnx_wire.py: README.md:199's other published format), once with every tensor under its
    # state_dict name and once as an exporter leaves the Conv + BatchNorm blocks it merged (conv.weight + conv.bias, no BN tensors)
    from head_detector_amd import onnx_wire

    p = str(tmp_path / "unfused.onnx")
    onnx_wire.write_model(p, sd, {k: "float_data" for k in list(sd)[::7]}, prefix="model.")
    archives["onnx initializers"] = (p, True)
    folded = dict(sd)
    for sp in arch.layer_specs(variant):
        if sp.kind in ("conv", "cbr"):
            pfx = sp.name if sp.kind == "conv" else f"{sp.name}.seq"
            s_, t_ = arch._bn_affine(sd, f"{pfx}.bn")
            folded[f"{pfx}.conv.weight"] = (sd[f"{pfx}.conv.weight"].astype(np.float64) * s_[:, None, None, None]).astype(np.float32)
            folded[f"{pfx}.conv.bias"] = t_.astype(np.float32)
            for k in [k for k in folded if k.startswith(pfx + ".bn.")]:
                del folded[k]
    p = str(tmp_path / "bn_folded.onnx")
    onnx_wire.write_model(p, folded, prefix="model.")
    archives["onnx, Conv+BN merged by the exporter"] = (p, False)
    # N4, the graph half (r06): what the reference's exporter actually leaves -- RepVGG blocks fused, BatchNorms merged, onnxsim-simplified, `onnx::Conv_NNN` names
    # (exportable_mesh_model.py:392-393,440-453,483-488) -- bound to the architecture by graph position (head_detector_amd/onnx_graph.py), L and M
    import os as _os

    sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
    from onnx_export_standin import write_simplified_export

    p = str(tmp_path / "simplified.onnx")
    write_simplified_export(p, variant, sd, seed=8)
    archives["onnx, fused + simplified + anonymous names (graph ingest)"] = (p, False)
    if variant == "vgg_heads_l":  # L: the graph ingest (this round's row) and one archive of each other kind
        archives = {k: v for k, v in archives.items() if k in ("torchscript unfused", "onnx, fused + simplified + anonymous names (graph ingest)")}

    for what, (path, unfused) in archives.items():
        got = load_weights(path)
        assert not any(weight_manifest_diff(variant, got).values()), what
        oracle = net_oracle.YoloHeadsOracle(okey)
        if unfused:
            res = oracle.load_state_dict({k: torch.from_numpy(v) for k, v in got.items()}, strict=False)
            # strict in both directions up to BatchNorm's bookkeeping counters (no tensor of the archive is ignored, no oracle weight keeps its initial value)
            assert not [k for k in res.missing_keys if "num_batches_tracked" not in k], (what, res.missing_keys[:5])
            assert not res.unexpected_keys, (what, res.unexpected_keys[:5])
        else:
            oracle.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        with torch.no_grad():
            ob_, os_, _ = oracle.dense(x)
        P = arch.build_program(variant, got, S, "fp16x3")
        pk = str(tmp_path / "a.vghpack")
        pack.write_pack(pk, P, flame_model, {}, B)
        h = C.c_void_p()
        cfg = _lib.Config(device=torch.cuda.current_device(), pack_path=pk.encode(), max_batch=B)
        _lib.check(gpu_lib.vgh_create(C.byref(cfg), C.byref(h)))
        kk = 100
        f32 = dict(dtype=torch.float32, device=_dev())
        ob, osc, of = torch.zeros(B, kk, 4, **f32), torch.zeros(B, kk, **f32), torch.zeros(B, kk, 413, **f32)
        oc, nh, hi = torch.zeros(B, dtype=torch.int32, device=_dev()), torch.zeros(1, dtype=torch.int32, device=_dev()), torch.zeros(B * kk, dtype=torch.int32, device=_dev())
        o = _lib.DetectOut(boxes_dev=ob.data_ptr(), scores_dev=osc.data_ptr(), flame_dev=of.data_ptr(), counts_dev=oc.data_ptr(), n_heads_dev=nh.data_ptr(), head_image_dev=hi.data_ptr(),
                           head_capacity=B * kk, unpad_dev=None, verts_dev=None, rot_dev=None, rpy_dev=None, proj_dev=None)
        conf = float(torch.sort(os_[:, :, 0].flatten(), descending=True).values[4 * B])  # a handful of detections per image
        st = torch.cuda.current_stream().cuda_stream
        xd = x.to(_dev()).contiguous()
        assert gpu_lib.vgh_ctx_detect(h, xd.data_ptr(), _lib.VGH_IMG_F32_NCHW, B, conf, 0.5, C.byref(o), st) == 0, gpu_lib.vgh_ctx_last_error(h)
        _lib.check(gpu_lib.vgh_ctx_join(h, st))
        torch.cuda.synchronize()
        det = gpu_lib.vgh_ctx_detector(h)
        A = ob_.shape[1]
        dense_b = _alias(gpu_lib.vgh_detector_scratch(det, _lib.SCRATCH_BOXES_ALL), (B, A, 4), "<f4", _dev()).cpu()
        dense_s = _alias(gpu_lib.vgh_detector_scratch(det, _lib.SCRATCH_SCORES_ALL), (B, A), "<f4", _dev()).cpu()
        assert float(_iou(dense_b, ob_).min()) >= 0.999, (what, float(_iou(dense_b, ob_).min()))
        assert float((dense_s - os_[..., 0]).abs().max()) < 1e-5, what
        # the detections it kept: each is some anchor of the oracle, with that anchor's box and parameters
        with torch.no_grad():
            _, _, of_ = oracle.dense(x)
        assert int(nh) >= B
        for b in range(B):
            for i in range(int(oc[b])):
                a = int((ob_[b] - ob[b, i].cpu()).abs().sum(-1).argmin())
                assert float(_iou(ob[b, i].cpu(), ob_[b, a])) >= 0.999
                rel = (of[b, i, :412].cpu() - of_[b, a, :412]).abs() / (of_[b, a, :412].abs() + 1.0)
                assert float(rel.max()) < 1e-4, (what, float(rel.max()))
        gpu_lib.vgh_destroy(h)
