"""CPU: host-side logic of the product -- weight folding + lowering to the op program (vs the unfused oracle
network), API containers, error behaviour, and the C ABI's exported symbol set.  No GPU compute."""
import os
import re

import numpy as np
import pytest
import torch

import program_ref as pr
from conftest import ROOT
from head_detector_amd import _lib, arch
from head_detector_amd.head_info import FLAME_CONSTS, FlameParams
from oracle import net_oracle, postproc_oracle as po


def _oracle_with(variant, sd):
    m = net_oracle.YoloHeadsOracle({"vgg_heads_m": "m", "vgg_heads_l": "l"}[variant])
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing[:3], unexpected[:3])
    return m


@pytest.mark.parametrize("variant,gflop", [("vgg_heads_m", 103.7721216), ("vgg_heads_l", 166.6826496)])
def test_program_flops_match_survey(variant, gflop):
    sd = arch.random_state_dict(variant, 1)
    P = arch.build_program(variant, sd, 640)
    assert abs(P.flops / 1e9 - gflop) < 1e-6  # 51.89 / 83.34 GMAC
    assert abs(net_oracle.conv_flops(_oracle_with(variant, sd)) - P.flops) < 1.0
    for op in P.ops:  # structural invariants the kernels rely on
        if op["kind"] == 1:
            assert op["cin"] % 32 == 0 and op["cout_pad"] % 32 == 0 and op["in_coff"] % 8 == 0
            if not P.bufs[op["out_buf"]]["is_f32"]:
                assert op["out_coff"] % 4 == 0 and op["out_coff2"] % 4 == 0 and op["out_split"] % 4 == 0 and op["cout_store"] % 4 == 0


@pytest.mark.parametrize("variant", ["vgg_heads_m", "vgg_heads_l"])
def test_lowering_reproduces_unfused_oracle(variant):
    """fold (BN, RepVGG branches, alpha) + sibling fusion + concat-by-offset + pixel-shuffle ConvTranspose, executed in fp32
    with torch ops, must equal the unfused module graph."""
    S = 128
    sd = arch.random_state_dict(variant, 3)
    P = arch.build_program(variant, sd, S)
    oracle = _oracle_with(variant, sd)
    x = torch.rand(2, 3, S, S, generator=torch.Generator().manual_seed(0))
    bufs = pr.run_program(P, x, bf16=False)
    ref = oracle.raw_heads(x)
    for (reg, cls, raw), (oreg, ocls, _, oraw) in zip(pr.head_outputs(P, bufs), ref):
        for a, b, name in [(reg, oreg, "reg"), (cls, ocls, "cls")] + [(raw[k], oraw[k], k) for k in raw]:
            scale = float(b.abs().max()) + 1e-6
            err = float((a - b).abs().max()) / scale
            assert err < 2e-4, (variant, name, err)
    # the K padding of the stage-1 downsample: in the bf16 program the stem tensor has a 48-channel pitch and the conv's 64-channel window meets 16 all-zero
    # weight columns; in the parity programs the pitch is 64 and the padded channels are stored zeros
    assert P.bufs[0]["pitch"] == 48 and P.ops[1]["cin"] == 64 and P.ops[1]["in_buf"] == 0
    w_all, _ = P.arrays()
    ds = P.ops[1]
    Wds = w_all[ds["w_off"] : ds["w_off"] + ds["cout_pad"] * 9 * 64].reshape(-1, 64)
    assert float(np.abs(Wds[:, 48:]).max()) == 0.0 and float(np.abs(Wds[:, :48]).max()) > 0.0
    P32 = arch.build_program(variant, sd, S, precision="fp16x3")
    assert P32.bufs[0]["pitch"] == 64 and P32.ops[0]["cout_store"] == 64
    assert float(pr.run_program(P32, x, bf16=False)[0][..., 48:].abs().max()) == 0.0


def test_u8_input_equals_float_input_in_program_ref():
    S = 64
    sd = arch.random_state_dict("vgg_heads_m", 5)
    P = arch.build_program("vgg_heads_m", sd, S)
    u8 = torch.randint(0, 256, (1, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    f = u8.permute(0, 3, 1, 2).float() / 255.0  # detector.py:51
    a = pr.alloc(P, 1)
    b = pr.alloc(P, 1)
    pr.run_op(P, P.ops[0], a, u8, False)
    pr.run_op(P, P.ops[0], b, f, False)
    assert torch.equal(a[0], b[0])


def test_fold_rejects_bad_state_dict():
    sd = arch.random_state_dict("vgg_heads_m", 1)
    bad = dict(sd)
    bad.pop("backbone.stage1.downsample.post_bn.weight")
    with pytest.raises(KeyError):
        arch.fold_state_dict("vgg_heads_m", bad)
    bad = dict(sd)
    bad["heads.head1.reg_pred.weight"] = np.zeros((68, 7, 1, 1), np.float32)
    with pytest.raises(ValueError):
        arch.fold_state_dict("vgg_heads_m", bad)


def test_flame_params_container_matches_reference_layout():
    from conftest import golden

    g = golden("layout.npz")
    x = torch.arange(413, dtype=torch.float32)[None]
    fp = FlameParams.from_3dmm(x)
    for k in ("shape", "expression", "jaw", "rotation", "eyeballs", "neck", "translation", "scale"):
        assert np.array_equal(getattr(fp, k).numpy().astype(np.int32)[0], g[f"read_{k}"])
    assert np.array_equal(fp.to_3dmm_tensor().numpy().astype(np.int32)[0], g["perm"])
    assert sum(FLAME_CONSTS.values()) == 413
    with pytest.raises(ValueError, match="Invalid number of parameters. Expected: 413. Got: 412."):
        FlameParams.from_3dmm(torch.zeros(1, 412))
    z = FlameParams.from_3dmm(torch.ones(2, 413), zero_expr=True)
    assert float(z.expression.abs().sum()) == 0.0 and z.shape.shape == (2, 300)


def test_host_rotation_helpers_match_reference():
    from conftest import golden
    from head_detector_amd.utils import calculate_rpy, limit_angle, rot_mat_from_6dof

    g = golden("rotation.npz")
    np.testing.assert_allclose(rot_mat_from_6dof(torch.from_numpy(g["v6"])).numpy(), g["R"], atol=1e-6)
    np.testing.assert_allclose([limit_angle(a) for a in g["lim_in"]], g["lim_out"])
    for i in range(1, 16):
        p = FlameParams.from_3dmm(torch.zeros(1, 413))
        p.rotation = torch.from_numpy(g["v6"][i : i + 1])
        np.testing.assert_allclose(np.array(calculate_rpy(p)), g["rpy"][i], atol=1e-3)


def test_c_abi_library_exports_every_declared_symbol():
    """The drop-in boundary: libvgh.so loads without a GPU and exports exactly what include/vgh.h declares."""
    hdr = open(os.path.join(ROOT, "include", "vgh.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    product, _, experiments = hdr.partition("#ifdef VGH_EXPERIMENTS")
    declared = set(re.findall(r"\b(vgh_[a-z0-9_]+)\s*\(", product))
    knobs = set(re.findall(r"\b(vgh_[a-z0-9_]+)\s*\(", experiments))
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert knobs == set(_lib.EXPERIMENT_SYMBOLS) and not (knobs & declared)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    # r06: the A/B knobs live in the -DVGH_EXPERIMENTS build only; the product library exports exactly the declared symbols
    import subprocess

    exported = {ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout.splitlines()
                if " T " in ln and ln.split()[-1].startswith("vgh_")}  # the C symbols (mangled C++ internals do not start with vgh_)
    assert exported == declared, (exported ^ declared)
    assert len(declared) <= 85
    assert lib.vgh_version().startswith(b"vgh")
    assert lib.vgh_conv_num_cfgs() >= 4 and lib.vgh_conv_cfg_name(0)
    # host-only entry point: weight packing is a pure permutation + bf16 rounding of the dense weights
    rng = np.random.default_rng(0)
    w = rng.standard_normal((64, 3, 3, 32)).astype(np.float32)
    out = np.zeros(w.size, dtype=np.uint16)
    assert lib.vgh_pack_conv_weights(_lib.ptr(w), 64, 3, 32, _lib.ptr(out)) == 0
    ref = torch.from_numpy(w).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert sorted(out.tolist()) == sorted(ref.reshape(-1).tolist())
    img = out.reshape(9, 64, 4, 8)  # [kb][cout][slot][8], slot = chunk ^ ((cout>>2)&3)
    for co in (0, 5, 13, 63):
        for chunk in range(4):
            assert np.array_equal(img[4, co, chunk ^ ((co >> 2) & 3)], ref[co, 1, 1, chunk * 8 : chunk * 8 + 8])
    assert lib.vgh_pack_conv_weights(_lib.ptr(w), 64, 3, 31, _lib.ptr(out)) != 0 and b"multiples of 32" in lib.vgh_last_error()


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from head_detector_amd.detector import HeadDetector
    from head_detector_amd.engine import VGHeadsEngine
    from head_detector_amd.flame import FLAMELayer
    from head_detector_amd.utils import nms

    with pytest.raises(_lib.VghError):
        HeadDetector()
    with pytest.raises(_lib.VghError):
        VGHeadsEngine("vgg_heads_m")
    with pytest.raises(_lib.VghError):
        nms(torch.zeros(1, 4, 4), torch.zeros(1, 4, 1), torch.zeros(1, 4, 413))
    with pytest.raises(FileNotFoundError):
        FLAMELayer()  # generic_model.pkl is a user-supplied asset (flame.py:18-24)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "head_detector_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_trcd_ingest_roundtrip(tmp_path):
    """N1: a TorchScript archive shaped like the released .trcd (module tree under `model.`, exportable_mesh_model.py:421-427)
    -> detector.load_weights -> fold: the state dict survives bit-exactly and folds without complaint."""
    from head_detector_amd.detector import load_weights

    variant = "vgg_heads_m"
    sd = arch.random_state_dict(variant, 2)
    oracle = _oracle_with(variant, sd)

    class Pipeline(torch.nn.Module):  # ConvertableCompletePipelineModel keeps the network under `.model`
        def __init__(self, m):
            super().__init__()
            self.model = m

        def forward(self, x):
            return self.model.neck(*self.model.backbone(x))

    ts = torch.jit.trace(Pipeline(oracle), torch.zeros(1, 3, 64, 64), check_trace=False)
    path = str(tmp_path / "vgg_heads_m.trcd")
    ts.save(path)
    got = load_weights(path)
    keys = set(sd)
    assert keys <= set(got), sorted(keys - set(got))[:5]
    for k in keys:
        assert np.array_equal(got[k], sd[k]), k
    folded = arch.fold_state_dict(variant, got)
    assert "backbone.stem.conv" in folded and folded["heads.head3.reg_pred"][0].shape == (68, 192, 1, 1)


def test_flame_pickle_ingest(tmp_path):
    """N1: FLAME pickle semantics of head_detector/flame.py:18-24,75-95 -- latin1 pickle, chumpy-wrapped arrays, scipy-sparse
    J_regressor, posedirs reshaped [V*3, P] then transposed, parents from kintree_table[0] with parents[0] = -1."""
    import pickle
    import sys
    import types

    import scipy.sparse as sp

    from head_detector_amd.flame import FLAMELayer
    from head_detector_amd.synthetic import synthetic_flame_model
    from oracle import flame_oracle as fo

    m = synthetic_flame_model(seed=5, V=97, NB=400, NJ=5)
    mod = types.ModuleType("chumpy")
    sub = types.ModuleType("chumpy.ch")

    class Ch:  # minimal stand-in for chumpy.ch.Ch: numeric payload in `.x`
        def __init__(self, x):
            self.x = x

    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    sub.Ch = Ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
    try:
        blob = dict(m)
        blob["v_template"] = Ch(m["v_template"])
        blob["shapedirs"] = Ch(m["shapedirs"])
        blob["J_regressor"] = sp.csc_matrix(m["J_regressor"])
        path = str(tmp_path / "generic_model.pkl")
        with open(path, "wb") as f:
            pickle.dump(blob, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    layer = FLAMELayer(flame_path=path)  # must not need chumpy installed
    ref = fo.FlameConstants(m, torch.float32)
    for name in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        assert torch.equal(getattr(layer, name), getattr(ref, name)), name
    assert layer.parents.tolist() == [-1, 0, 1, 1, 1] and layer.faces_tensor.shape == (9976, 3)
    with pytest.raises(FileNotFoundError):
        FLAMELayer(flame_path=str(tmp_path / "missing.pkl"))


def test_c_abi_argument_validation_reports_errors_without_a_gpu():
    """Every entry point validates its arguments before touching the device: a bad call returns a negative code and leaves a
    message in vgh_last_error() (never throws, never computes on the CPU).  Exercised here without a GPU."""
    import ctypes as C

    from head_detector_amd import _lib

    lib = _lib.load()

    def err(rc):
        assert rc < 0
        return lib.vgh_last_error().decode()

    assert "null" in err(lib.vgh_net_forward(None, None, 0, 1, None))
    assert "null" in err(lib.vgh_flame_decode(None, None, 1, 300, 100, None, None, None, None, None))
    fake = 0x1000  # non-null placeholders: validation fails before any of them would be dereferenced
    geo = dict(B=1, H=8, W=8, in_dev=fake, wpack_dev=fake, bias_dev=fake, out_dev=fake, in_pitch=64, out_pitch=64, cout_store=64, out_split=64)
    # vgh_conv2d looks at the current device first: without a GPU that is the (loud) error, with one the shape check is
    m = err(lib.vgh_conv2d(C.byref(_lib.ConvCall(cin=48, cout_pad=64, ksize=3, stride=1, **geo)), None))
    assert "cin=48" in m or "no ROCm-capable device" in m
    m = err(lib.vgh_conv2d(C.byref(_lib.ConvCall(cin=32, cout_pad=64, ksize=5, stride=1, **geo)), None))
    assert "ksize=5" in m or "no ROCm-capable device" in m
    assert "null" in err(lib.vgh_rasterize(None, None, 3, None, 3, None, 8, 8, 0, None, None))
    assert "null" in err(lib.vgh_letterbox(None, 4, 4, 3, 12, None, None, None, None, 4, 4, 0, 0, None, None, 8, None))
    assert "null" in err(lib.vgh_detector_create(None, None, None, None))
    assert "1..4" in err(lib.vgh_net_set_split(C.c_void_p(1), 9)) or "null" in lib.vgh_last_error().decode()
    with pytest.raises(_lib.VghError, match="libvgh error"):
        _lib.check(lib.vgh_net_forward(None, None, 0, 1, None))
    assert lib.vgh_conv_num_cfgs() > 70 and lib.vgh_conv_cfg_name(19).decode().startswith("p8x40")


def test_ping_pong_tiles_qualify_only_for_their_conv_class():
    """vgh_conv_cfg_ok (host logic, no GPU): the g / h tiles of csrc/conv_pp.hip take 3x3 / stride-1 bf16 convs with the 16-byte epilogue whose cout_pad is a
    multiple of their cout tile -- what the tuner and the table loader rely on when a key names one."""
    from head_detector_amd import _lib

    lib = _lib.load()
    names = [lib.vgh_conv_cfg_name(i).decode() for i in range(lib.vgh_conv_num_cfgs())]
    for fam in "ghs":  # s (r06): the g tiles with two 4 x 8 sub-patches per wave
        for bc in (128, 96, 64):
            c = names.index(f"{fam}8x8x{bc}_n8" if fam != "s" else f"s4x8x{bc}_n8")
            assert lib.vgh_conv_cfg_cout_tile(c) == bc
            assert lib.vgh_conv_cfg_ok(c, 3, 1, 3 * bc, 1, 0) == 1
            assert lib.vgh_conv_cfg_ok(c, 3, 1, 3 * bc + 32, 1, 0) == 0  # cout_pad not a multiple of the tile
            assert lib.vgh_conv_cfg_ok(c, 3, 2, 3 * bc, 1, 0) == 0  # stride 2
            assert lib.vgh_conv_cfg_ok(c, 1, 1, 3 * bc, 1, 0) == 0  # 1x1
            assert lib.vgh_conv_cfg_ok(c, 3, 1, 3 * bc, 0, 0) == 0  # fp32 / unaligned store
            assert lib.vgh_conv_cfg_ok(c, 3, 1, 3 * bc, 1, 1) == 0  # ConvTranspose pixel shuffle
    assert lib.vgh_conv_cfg_ok(names.index("s4x8x128_n8"), 3, 1, 1024, 1, 0) == 1 and lib.vgh_conv_cfg_ok(names.index("s4x8x128_n8"), 3, 1, 1152, 1, 0) == 0  # 4-KB bias vector


def test_save_meshes_matches_reference_obj_bytes(tmp_path):
    """PredictionResult.save_meshes writes byte-for-byte what the reference's MeshSaver / save_meshes wrote for the same vertices and
    triangles (tests/golden/mesh_obj.npz, produced by running head_detector/detection_result.py:22-35,73-78)."""
    import types

    from conftest import golden
    from head_detector_amd.detection_result import PredictionResult

    g = golden("mesh_obj.npz")
    heads = [types.SimpleNamespace(vertices_3d=v) for v in g["heads"]]
    pr = PredictionResult(np.zeros((4, 4, 3), dtype=np.uint8), heads, faces=g["faces"])
    pr.save_meshes(str(tmp_path / "meshes"))
    names = sorted(os.listdir(tmp_path / "meshes"))
    assert names == [str(n) for n in g["names"]]
    for i, n in enumerate(names):
        assert open(tmp_path / "meshes" / n, "rb").read() == g[f"obj{i}"].tobytes()
    with pytest.raises(ValueError):
        PredictionResult(np.zeros((4, 4, 3), dtype=np.uint8), heads).save_meshes(str(tmp_path / "none"))


def test_weight_manifest_diff_names_every_mismatch():
    """N1 / ADVICE: a released blob whose parameters are renamed or re-shaped must fail with the whole difference up front."""
    from head_detector_amd.detector import weight_manifest_diff

    sd = arch.random_state_dict("vgg_heads_m", 3)
    assert not any(weight_manifest_diff("vgg_heads_m", sd).values())
    sd["backbone.stem.conv.post_bn.num_batches_tracked"] = np.zeros(())  # bookkeeping buffers of a real archive are ignored
    assert not any(weight_manifest_diff("vgg_heads_m", sd).values())
    bad = dict(sd)
    bad["backbone.stage1.downsample.rbr_reparam.weight"] = np.zeros((96, 48, 3, 3), np.float32)  # fused twin kept by SG: ignored
    del bad["heads.head1.cls_pred.bias"]
    bad["heads.head2.reg_pred.weight"] = np.zeros((68, 5, 1, 1), np.float32)
    bad["neck.neck9.extra"] = np.zeros(3, np.float32)
    d = weight_manifest_diff("vgg_heads_m", bad)
    assert d["missing"] == ["heads.head1.cls_pred.bias"] and d["unexpected"] == ["neck.neck9.extra"] and len(d["shape"]) == 1 and "heads.head2.reg_pred.weight" in d["shape"][0]
    assert len(weight_manifest_diff("vgg_heads_l", sd)["shape"]) > 10  # an M archive offered as L


def test_pack_file_round_trip(tmp_path):
    """.vghpack (head_detector_amd/pack.py): header fields, section sizes and the per-op tile names survive a write/read cycle; the
    op / buffer records use the C layout of include/vgh.h (what csrc/ctx.hip freads)."""
    import ctypes as C

    from head_detector_amd import pack
    from head_detector_amd.synthetic import synthetic_flame_model

    P = arch.build_program("vgg_heads_m", arch.random_state_dict("vgg_heads_m", 7), 128)
    assert pack.tile_names_for(P, 32, 2) == {}  # the measured table holds 640 / 1280 geometries only
    P640 = arch.build_program("vgg_heads_m", arch.random_state_dict("vgg_heads_m", 7), 640)
    n640 = pack.tile_names_for(P640, 32, 2)
    assert len(n640) > 90 and all(P640.ops[i]["kind"] == 1 for i in n640)
    names = {i: ("256x128_w64x64_k1_r3" if i % 2 else "q16x16x64_n4x1") for i, op in enumerate(P.ops) if op["kind"] == 1 and i % 3}
    fm = synthetic_flame_model(seed=3)
    path = str(tmp_path / "m.vghpack")
    n = pack.write_pack(path, P, fm, names, 32)
    h = pack.read_header(path)
    assert (h["variant"], h["image_size"], h["n_ops"], h["n_bufs"], h["n_levels"], h["shape_c"], h["expr_c"], h["has_flame"], h["tune_batch"]) == \
        ("vgg_heads_m", 128, len(P.ops), len(P.bufs), 3, 64, 32, 1, 32)
    assert (h["V"], h["NB"], h["NJ"]) == (5023, 400, 5) and abs(h["flops_per_image"] - P.flops) < 1
    assert C.sizeof(_lib.OpDesc) == 104 and C.sizeof(_lib.BufDesc) == 20
    w, b = P.arrays()
    flame_bytes = 4 * (5023 * 3 + 5023 * 3 * 400 + 36 * 3 * 5023 + 5 * 5023 + 5 + 5023 * 5) + 4 * 3 * h["F"]
    assert n == 128 + 20 * len(P.bufs) + (104 + 32) * len(P.ops) + 20 * 3 + 4 * (w.size + b.size) + flame_bytes
    raw = open(path, "rb").read()
    off = 128 + 20 * len(P.bufs)
    ops = (_lib.OpDesc * len(P.ops)).from_buffer_copy(raw[off : off + 104 * len(P.ops)])
    assert [o.cout_pad for o in ops] == [op["cout_pad"] for op in P.ops] and [o.w_off for o in ops] == [op["w_off"] for op in P.ops]
    tn = np.frombuffer(raw[off + 104 * len(P.ops) : off + 136 * len(P.ops)], dtype="S32")
    assert {i: t.decode() for i, t in enumerate(tn) if t} == names
    woff = off + 136 * len(P.ops) + 60
    assert np.array_equal(np.frombuffer(raw[woff : woff + 4 * w.size], dtype=np.float32), w)
    with pytest.raises(ValueError):
        open(tmp_path / "junk", "wb").write(b"x" * 200)
        pack.read_header(str(tmp_path / "junk"))
    # CLI
    pack.main(["vgg_heads_m", "seed:7", "none", str(tmp_path / "cli.vghpack"), "--image-size", "128", "--batch", "32"])
    assert pack.read_header(str(tmp_path / "cli.vghpack"))["has_flame"] == 0


def test_tuning_lookup_precedence_and_kernel_name_list():
    """The measured tile table is keyed by batch bucket and lane count: an entry measured with this lane count wins over one measured without
    lanes, the b64 bucket falls back to b32 (which covered every large batch before r03), the parity modes carry a precision prefix -- and every
    tile the committed table names exists in the library.  arch.NET_KERNEL_MARKERS (what the PMC tools sum over) covers every kernel family that
    runs an op of the program."""
    import json

    from head_detector_amd import pack
    from head_detector_amd.engine import TUNING_DIR, tuning_key, tuning_lookup

    P = arch.build_program("vgg_heads_l", arch.random_state_dict("vgg_heads_l", 1), 640)
    op = next(o for o in P.ops if o["kind"] == 1 and o["ksize"] == 3 and o["stride"] == 1)
    k64x2, k32x2, k64, k32 = tuning_key(op, 64, 2), tuning_key(op, 64, 2, bucket=32), tuning_key(op, 64), tuning_key(op, 64, bucket=32)
    assert k64x2.startswith("b64x2_") and k32x2.startswith("b32x2_") and k64.startswith("b64_m") and k32.startswith("b32_m") and tuning_key(op, 32, 2) == k32x2
    assert tuning_key(op, 16, 2).startswith("b8x2_") and tuning_key(op, 1).startswith("b1_")
    t = {k32: "d"}
    assert tuning_lookup(t, op, 64, 2) == "d" and tuning_lookup(t, op, 64, 1) == "d" and tuning_lookup(t, op, 32, 2) == "d" and tuning_lookup(t, op, 16, 2) is None
    t[k64] = "c"
    assert tuning_lookup(t, op, 64, 2) == "c" and tuning_lookup(t, op, 32, 2) == "d"
    t[k32x2] = "b"
    assert tuning_lookup(t, op, 64, 2) == "b" and tuning_lookup(t, op, 64, 1) == "c" and tuning_lookup(t, op, 32, 2) == "b"
    t[k64x2] = "a"
    assert tuning_lookup(t, op, 64, 2) == "a" and tuning_lookup(t, op, 32, 2) == "b"
    assert tuning_lookup(t, op, 64, 2, "fp16x3:") is None and tuning_lookup({"fp16x3:" + k32x2: "s"}, op, 64, 2, "fp16x3:") == "s"
    table = json.load(open(os.path.join(TUNING_DIR, "conv_cfg.json")))
    lib = _lib.load()
    bf16 = {lib.vgh_conv_cfg_name(i).decode() for i in range(lib.vgh_conv_num_cfgs())}
    split = {lib.vgh_conv_split_cfg_name(i).decode() for i in range(lib.vgh_conv_split_num_cfgs())}
    for key, name in table.items():
        assert name in (split if ":" in key else bf16), (key, name)
    assert {n[0] for n in bf16} >= set("pqtghsrw") and "d" not in {n[0] for n in bf16} and any(n.startswith("t") for n in table.values())  # halo-patch v2 / v3, streaming 1x1, ping-pong g / h / s, register-resident r / w (csrc/ds_b2b.hip); the stride-2 "d" tiles: experiments build only (r06)
    # the parity mode's pack carries the tile names of ITS table
    P3 = arch.build_program("vgg_heads_l", arch.random_state_dict("vgg_heads_l", 1), 640, "fp16x3")
    n3 = pack.tile_names_for(P3, 32, 2)
    assert len(n3) > 40 and set(n3.values()) <= split
    for k in ("conv_igemm_kernel<256, 128, 64, 64, 1, 1, 3, 0>", "conv3x3_patch_kernel<16, 16, 64, 4, 1, 0, 0>", "conv3x3_patch3_kernel<16, 16, 96, 4, 1>", "conv1x1_stream_kernel<128, 96, 32, 96, 1, 3>",
              "stem_kernel<1, 1, 0>", "stem_ds_kernel<0>", "ds_b2b_kernel<6, 0>", "ds_conv_kernel", "w_conv_kernel<8, 4, 1>", "spp_pool_kernel", "spp_pool_split_kernel<3>", "conv_f32_kernel", "stem_f32_kernel", "spp_pool_f32_kernel"):
        assert arch.is_net_kernel("void (anonymous namespace)::" + k + "(ConvArgs, int)"), k
    assert not arch.is_net_kernel("void (anonymous namespace)::flame_mfma_lds_kernel<2, 64>(VertArgs)") and not arch.is_net_kernel("__amd_rocclr_fillBufferAligned")


def test_fused_archive_keys_fold_to_the_same_network():
    """an archive exported after QARepVGG fusion (`rbr_reparam` instead of the branch tensors; post_bn kept when only
    partially fused) passes the manifest check and folds to the same conv + bias as the unfused keys."""
    from head_detector_amd.detector import weight_manifest_diff

    variant = "vgg_heads_m"
    sd = arch.random_state_dict(variant, 9)
    F = arch.fold_state_dict(variant, sd)
    qarep = [sp for sp in arch.layer_specs(variant) if sp.kind == "qarep"]
    full = {k: v for k, v in sd.items() if not any(k.startswith(sp.name + ".") for sp in qarep)}
    part = dict(full)
    for sp in qarep:
        W, b = F[sp.name]
        full[f"{sp.name}.rbr_reparam.weight"], full[f"{sp.name}.rbr_reparam.bias"] = W.astype(np.float32), b.astype(np.float32)
        s_, t_ = arch._bn_affine(sd, f"{sp.name}.post_bn")
        part[f"{sp.name}.rbr_reparam.weight"], part[f"{sp.name}.rbr_reparam.bias"] = (W / s_[:, None, None, None]).astype(np.float32), ((b - t_) / s_).astype(np.float32)
        for k in sd:
            if k.startswith(sp.name + ".post_bn."):
                part[k] = sd[k]
    for what, d, tol in (("full", full, 0.0), ("partial", part, 1e-5)):
        diff = weight_manifest_diff(variant, d)
        assert not any(diff.values()), (what, {k: v[:3] for k, v in diff.items()})
        G = arch.fold_state_dict(variant, d)
        for sp in qarep:
            W, b = F[sp.name]
            assert np.abs(G[sp.name][0] - W.astype(np.float32)).max() <= tol * (np.abs(W).max() + 1e-30) + 1e-12 and np.abs(G[sp.name][1] - b.astype(np.float32)).max() <= tol * (np.abs(b).max() + 1) + 1e-12, (what, sp.name)
    broken = dict(full)
    del broken[f"{qarep[3].name}.rbr_reparam.bias"]
    assert weight_manifest_diff(variant, broken)["missing"] == [f"{qarep[3].name}.rbr_reparam.bias"]


def test_archives_with_unexpected_key_names_are_reported_in_full(tmp_path):
    """First contact with a released blob whose key names differ from the reconstruction must name EVERY difference before anything is folded:
    a DataParallel `module.` prefix, a block without its post_bn, BatchNorm bookkeeping counters (ignored), a tensor of another shape."""
    from head_detector_amd import arch
    from head_detector_amd.detector import load_weights, weight_manifest_diff

    variant = "vgg_heads_m"
    sd = arch.random_state_dict(variant, 3)
    qarep = next(sp.name for sp in arch.layer_specs(variant) if sp.kind == "qarep")
    bad = {}
    for k, v in sd.items():
        if k.startswith(qarep + ".post_bn."):
            continue  # this block lost its post_bn
        bad["module." + k if k.startswith("heads.") else k] = v  # the heads carry a DataParallel prefix
    bad[qarep + ".branch_3x3.bn.num_batches_tracked"] = np.zeros((), np.int64)  # bookkeeping: ignored
    some = next(k for k in sd if k.endswith("conv.weight") and k.startswith("neck."))
    bad[some] = np.zeros((3, 3, 1, 1), np.float32)
    p = str(tmp_path / "odd.pth")
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in bad.items()}, p)
    diff = weight_manifest_diff(variant, load_weights(p))
    n_heads = sum(k.startswith("heads.") for k in sd)
    assert len([k for k in diff["missing"] if k.startswith("heads.")]) == n_heads and len([k for k in diff["unexpected"] if k.startswith("module.heads.")]) == n_heads
    assert all(k in diff["missing"] for k in sd if k.startswith(qarep + ".post_bn.") and "num_batches_tracked" not in k)
    assert not any("num_batches_tracked" in k for k in diff["unexpected"])
    assert any(s.startswith(some + ":") for s in diff["shape"])
    with pytest.raises(Exception):
        arch.build_program(variant, load_weights(p), 256)


def _bn_folded_by_exporter(variant, sd):
    """The state dict as torch.onnx.export leaves a Conv + BatchNorm block whose BN it merged but whose conv kept its name: `<block>.conv.weight`, `<block>.conv.bias`, no BN keys."""
    out = dict(sd)
    for sp in arch.layer_specs(variant):
        if sp.kind in ("conv", "cbr"):
            pfx = sp.name if sp.kind == "conv" else f"{sp.name}.seq"
            s_, t_ = arch._bn_affine(sd, f"{pfx}.bn")
            out[f"{pfx}.conv.weight"] = (sd[f"{pfx}.conv.weight"].astype(np.float64) * s_[:, None, None, None]).astype(np.float32)
            out[f"{pfx}.conv.bias"] = t_.astype(np.float32)
            for k in list(out):
                if k.startswith(pfx + ".bn."):
                    del out[k]
    return out


def test_onnx_initializer_ingest_without_the_onnx_package(tmp_path):
    """N4 (ingest half; /root/reference/README.md:199 publishes ONNX weights): ModelProto.graph.initializer read by head_detector_amd/onnx_wire.py -- raw_data f32,
    packed float_data, fp16 / bf16 raw data, proto2-style unpacked dims -- through load_weights -> weight_manifest_diff -> fold_state_dict: the same folded network as
    the state dict the file was written from; an export whose Conv + BN blocks arrive merged folds to the same network too; malformed files fail with a reason."""
    from head_detector_amd import onnx_wire
    from head_detector_amd.detector import load_weights, weight_manifest_diff

    variant = "vgg_heads_m"
    sd = arch.random_state_dict(variant, 4)
    keys = list(sd)
    how = {keys[1]: "float_data", keys[2]: "dims_unpacked", keys[5]: "float_data"}
    extra = {"anchor_points": np.arange(6, dtype=np.int64).reshape(3, 2), "onnx::Reshape_991": np.array([1, -1], dtype=np.int64), "scalar_eps": np.array(1e-6, dtype=np.float64)}
    p = str(tmp_path / "vgg_heads_m.onnx")
    onnx_wire.write_model(p, {**sd, **extra}, how, prefix="model.")
    tensors, nodes = onnx_wire.load_initializers(p)
    assert nodes == [] and tensors["model.onnx::Reshape_991"].tolist() == [1, -1] and tensors["model.scalar_eps"].shape == () and tensors["model.anchor_points"].dtype == np.int64
    got = load_weights(p)
    assert set(sd) <= set(got)
    for k in sd:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], sd[k]), k
    diff = weight_manifest_diff(variant, got)
    # r06: initializers that are not parameters (shape constants, anchors, stray scalars) are set aside by the loader and listed, not reported as unexpected keys
    from head_detector_amd import detector as _det

    assert not any(diff.values()) and _det.LAST_LOAD_REPORT["how"] == "by name"
    assert sorted(_det.LAST_LOAD_REPORT["not_weights"]) == ["model.anchor_points", "model.onnx::Reshape_991", "model.scalar_eps"]
    F, G = arch.fold_state_dict(variant, sd), arch.fold_state_dict(variant, got)
    assert all(np.array_equal(F[k][0], G[k][0]) for k in F)
    # half-precision exports: values are the 16-bit roundings of the source
    p16 = str(tmp_path / "half.onnx")
    onnx_wire.write_model(p16, {k: (v.astype(np.float16) if i % 2 else v) for i, (k, v) in enumerate(sd.items())}, {k: "bf16" for i, k in enumerate(keys) if i % 2 == 0}, prefix="model.")
    g16 = load_weights(p16)
    for i, k in enumerate(keys):
        ref = sd[k].astype(np.float16).astype(np.float32) if i % 2 else torch.from_numpy(sd[k]).bfloat16().float().numpy()
        assert np.array_equal(g16[k], ref), k
    # Conv + BN merged by the exporter
    folded = _bn_folded_by_exporter(variant, sd)
    pf = str(tmp_path / "folded.onnx")
    onnx_wire.write_model(pf, folded, prefix="model.")
    gf = load_weights(pf)
    assert not any(weight_manifest_diff(variant, gf).values())
    H = arch.fold_state_dict(variant, gf)
    for k in F:
        if F[k][1] is not None:
            assert np.abs(H[k][0] - F[k][0]).max() <= 1e-6 * (np.abs(F[k][0]).max() + 1e-30) and np.abs(H[k][1] - F[k][1]).max() <= 1e-6 * (np.abs(F[k][1]).max() + 1), k
    # malformed input
    raw = open(p, "rb").read()
    (tmp_path / "cut.onnx").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(onnx_wire.OnnxWireError, match="runs past the end|truncated"):
        load_weights(str(tmp_path / "cut.onnx"))
    (tmp_path / "nograph.onnx").write_bytes(onnx_wire._vi(1, 8))
    with pytest.raises(onnx_wire.OnnxWireError, match="no ModelProto.graph"):
        load_weights(str(tmp_path / "nograph.onnx"))
    ext = onnx_wire._ld(7, onnx_wire._ld(5, onnx_wire._ld(1, onnx_wire._enc_varint(4)) + onnx_wire._vi(2, 1) + onnx_wire._ld(8, b"w") + onnx_wire._vi(14, 1)))
    (tmp_path / "ext.onnx").write_bytes(ext)
    with pytest.raises(onnx_wire.OnnxWireError, match="external file"):
        load_weights(str(tmp_path / "ext.onnx"))
    short = onnx_wire._ld(7, onnx_wire._ld(5, onnx_wire._ld(1, onnx_wire._enc_varint(4)) + onnx_wire._vi(2, 1) + onnx_wire._ld(9, b"\0" * 8) + onnx_wire._ld(8, b"w")))
    (tmp_path / "short.onnx").write_bytes(short)
    with pytest.raises(onnx_wire.OnnxWireError, match="want 4 elements"):
        load_weights(str(tmp_path / "short.onnx"))


def test_onnx_graph_ingest_binds_a_simplified_export_by_topology(tmp_path):
    """N4, the graph half (r06).  The reference's exporter fuses the RepVGG blocks, merges every BatchNorm and runs onnxsim.simplify
    (exportable_mesh_model.py:392-393,440-453,483-488): what reaches the user has `onnx::Conv_NNN` initializers and no parameter name.  tests/onnx_export_standin.py writes
    that shape of file (fused weights, anonymous names, shuffled initializers, Relu / Add(Mul(x, alpha)) / Concat / MaxPool nodes, the heads' tanh * 3 and exp / 0.05 tail, a
    DFL projection conv); head_detector_amd/onnx_graph.py must bind every Conv node to its module by topology + shape and return the same folded network -- for both
    variants, with the residual scale as a Constant node, with CSP conv2 written ahead of conv1's bottlenecks, in fp16 -- and refuse a graph with a missing or a
    wrong-stride node AT THAT MODULE."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from onnx_export_standin import write_simplified_export

    from head_detector_amd import detector as _det
    from head_detector_amd import onnx_graph
    from head_detector_amd.detector import load_weights, weight_manifest_diff

    def same_network(F, G, tol):
        for k in F:
            if F[k][1] is None:
                assert abs(F[k][0] - G[k][0]) <= tol * max(1.0, abs(F[k][0])), k
            else:
                assert np.abs(G[k][0] - F[k][0]).max() <= tol * (np.abs(F[k][0]).max() + 1e-30) and np.abs(G[k][1] - F[k][1]).max() <= tol * (np.abs(F[k][1]).max() + 1), k

    for variant in ("vgg_heads_m", "vgg_heads_l"):
        sd = arch.random_state_dict(variant, 9)
        F = arch.fold_state_dict(variant, sd)
        p = str(tmp_path / f"{variant}_sim.onnx")
        info = write_simplified_export(p, variant, sd, seed=2)
        got = load_weights(p)  # no variant given: the other architecture must fail to bind, this one must bind
        rep = dict(_det.LAST_LOAD_REPORT)
        assert rep["how"] == "by graph position" and rep["variant"] == variant and rep["conv_nodes_bound"] == info["n_conv"] == sum(1 for sp in arch.layer_specs(variant) if sp.kind != "alpha")
        assert len(rep["conv_nodes_unbound"]) == 3 and all("(1, 17, 1, 1)" in d for d in rep["conv_nodes_unbound"])  # the three DFL projection convs are left alone
        assert not any(weight_manifest_diff(variant, got).values())
        assert not any(k.startswith("onnx::") for k in got)
        same_network(F, arch.fold_state_dict(variant, got), 1e-6)
        arch.build_program(variant, got, 256)  # lowers
    variant = "vgg_heads_m"
    sd = arch.random_state_dict(variant, 10)
    F = arch.fold_state_dict(variant, sd)
    for kw, tol in ((dict(alpha_as_constant_node=True), 1e-6), (dict(swap_siblings=True), 1e-6), (dict(fp16=True), 2e-3)):
        p = str(tmp_path / "v.onnx")
        write_simplified_export(p, variant, sd, seed=3, **kw)
        same_network(F, arch.fold_state_dict(variant, load_weights(p, variant)), tol)
    p = str(tmp_path / "dropped.onnx")
    write_simplified_export(p, variant, sd, seed=4, drop_module="backbone.stage2.blocks.bottlenecks.1.cv2")
    with pytest.raises(onnx_graph.OnnxGraphError, match=r"no node for module 'backbone\.stage2\.blocks\.bottlenecks\.1\.cv[12]'"):
        load_weights(p, variant)
    p = str(tmp_path / "stride.onnx")
    write_simplified_export(p, variant, sd, seed=5, wrong_stride_module="neck.neck3.conv")
    with pytest.raises(onnx_graph.OnnxGraphError, match=r"no node for module 'neck\.neck3\.conv'.*strides \[2, 2\]"):
        load_weights(p, variant)
    with pytest.raises(onnx_graph.OnnxGraphError, match="does not bind to any known architecture"):
        load_weights(p)


@pytest.mark.parametrize("variant,prec", [("vgg_heads_l", "bf16"), ("vgg_heads_m", "bf16"), ("vgg_heads_m", "fp32"), ("vgg_heads_l", "fp16x3")])
def test_latency_schedule_is_a_valid_order_with_acyclic_lane_waits(variant, prec):
    """arch.schedule_latency (r06): (i) the reordered program is a valid SERIAL order -- every op's input / residual tensor has been written by an earlier op --
    (ii) every cross-lane reader of a tensor is ordered behind its writer by the one-op waits plus stream order, and (iii) the wait-for relation among the SIDE lanes
    is acyclic: ROCm 7.0's hipStreamEndCapture recurses without end on two side streams that waited for each other's events, and csrc/net.hip refuses such a capture."""
    sd = arch.random_state_dict(variant, seed=3)
    P0 = arch.build_program(variant, sd, 320, prec)
    names0 = sorted(op["name"] for op in P0.ops)
    P = arch.schedule_latency(arch.build_program(variant, sd, 320, prec))
    assert sorted(op["name"] for op in P.ops) == names0
    lane = [op.get("lane", 0) & 255 for op in P.ops]
    dep = [(op.get("lane", 0) >> 8) - 1 for op in P.ops]
    assert all(-1 <= d < i for i, d in enumerate(dep)) and all(lane[d] != lane[i] for i, d in enumerate(dep) if d >= 0)
    # happens-before: op j precedes op i if same lane and j < i, or i waits for j -- transitively
    n = len(P.ops)
    before = [set() for _ in range(n)]
    last_on = {}
    for i in range(n):
        for j in ([last_on[lane[i]]] if lane[i] in last_on else []) + ([dep[i]] if dep[i] >= 0 else []):
            before[i] |= before[j] | {j}
        last_on[lane[i]] = i
    def reads(op):  # (buffer, first channel, end channel) windows an op reads -- tests/program_ref.py's run_op
        out = []
        if op["kind"] in (1, 2) and op["in_buf"] >= 0:
            g = op["cout_pad"] // op["grp_cout"] if op.get("grp_cout") else 1
            out.append((op["in_buf"], op["in_coff"], op["in_coff"] + (g - 1) * op.get("grp_in_stride", 0) + op["cin"]))
        if op.get("res_buf", -1) >= 0:
            out.append((op["res_buf"], op["res_coff"], op["res_coff"] + op["cout_store"]))
        return out

    def writes(op):
        if op["kind"] == 2:
            return [(op["out_buf"], op["out_coff"], P.bufs[op["out_buf"]]["pitch"])]
        store = op["cout_store"] if not op["shuffle"] else op["cout_pad"] // 4
        split = min(op["out_split"], store)
        return [(op["out_buf"], op["out_coff"], op["out_coff"] + split)] + ([(op["out_buf"], op["out_coff2"], op["out_coff2"] + store - split)] if store > split else [])

    written, was_read = [], []
    for i, op in enumerate(P.ops):
        if op["kind"] not in (0, 1, 2):
            continue
        for b, lo, hi in reads(op):
            for w, (wb, wlo, whi) in written:  # read after write
                if wb == b and wlo < hi and lo < whi:
                    assert w in before[i], (P.ops[w]["name"], "->", op["name"])
        for b, lo, hi in writes(op):
            for j, (ob, olo, ohi) in written + was_read:  # write after write / write after read: a window is written again only behind everything that touched it
                if ob == b and olo < hi and lo < ohi and j != i:
                    assert j in before[i], (P.ops[j]["name"], "touched before", op["name"], "rewrites it")
        written += [(i, w) for w in writes(op)]
        was_read += [(i, r) for r in reads(op)]
    waits = {(lane[i], lane[d]) for i, d in enumerate(dep) if d >= 0 and lane[i] and lane[d]}
    reach = set(waits)
    for _ in range(4):
        reach |= {(a, d) for a, b in reach for c, d in reach if b == c}
    assert waits and not any(a == b for a, b in reach), sorted(waits)


def test_module_graph_is_the_oracles_forward_order():
    """arch.module_graph (what the ONNX binder and the export stand-in walk) against oracle/net_oracle.py run under forward hooks: the same conv modules in the same
    execution order, with the same weight shapes -- the oracle is the checker here (its module tree restates super_gradients' forward bodies and
    yolo_head_dfl_head.py:143-164)."""
    from oracle import net_oracle

    for variant, okey in (("vgg_heads_m", "m"), ("vgg_heads_l", "l")):
        net = net_oracle.YoloHeadsOracle(okey).eval()
        order = []
        leaf = (net_oracle.Conv, net_oracle.ConvBNReLU, net_oracle.QARepVGGBlock, torch.nn.ConvTranspose2d)
        hooks = []
        for name, mod in net.named_modules():
            is_pred = isinstance(mod, torch.nn.Conv2d) and (name.endswith(("reg_pred", "cls_pred")) or (".flame_" in name and name.rsplit(".", 1)[-1].isdigit()))
            if isinstance(mod, leaf) or is_pred:
                hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: order.append(name)))
        with torch.no_grad():
            net.dense(torch.rand(1, 3, 64, 64))
        for h in hooks:
            h.remove()
        mine = [m["name"] for m in arch.module_graph(variant) if m["op"] in ("conv", "convT")]
        assert order == mine, [(a, b) for a, b in zip(order, mine) if a != b][:5]


def test_fp8_and_fp16_programs_are_the_bf16_program_with_other_storage():
    """r05 (CPU): precision="fp8" is the bf16 op program with OCP-e4m3 LINKS -- a buffer written by one 3x3 / stride-1 conv and read by one: every bottleneck's cv1 -> cv2
    tensor on maps of at least fp8_min_px pixels a side, and the shape | expression part of the FLAME branches' middle layer (a buffer of its own beside the bf16 transform
    channels) -- with 64-byte K blocks (a 96-channel link sits in a 128-byte pixel over zero weight columns); precision="fp16" is the same program with every 16-bit buffer
    as ONE fp16 plane.  Same ops, same FLOPs, same weights; the algorithmic byte count follows the storage."""
    for variant, n_links, min20 in (("vgg_heads_l", 24, 31), ("vgg_heads_m", 17, 22)):
        sd = arch.random_state_dict(variant, 2)
        Pb = arch.build_program(variant, sd, 640, "bf16")
        P8 = arch.build_program(variant, sd, 640, "fp8", fp8_scales={"backbone.stage2.blocks.mid0": 10.0})
        Ph = arch.build_program(variant, sd, 640, "fp16")
        assert [o["name"] for o in P8.ops] == [o["name"] for o in Pb.ops] == [o["name"] for o in Ph.ops] and P8.flops == Pb.flops == Ph.flops
        links = [i for i, bf in enumerate(P8.bufs) if bf["is_f32"] == arch.FMT_FP8]
        assert len(links) == n_links and set(arch.fp8_link_names(variant)) == {P8.bufs[i]["name"] for i in links}
        assert len([bf for bf in arch.build_program(variant, sd, 640, "fp8", fp8_min_px=20).bufs if bf["is_f32"] == arch.FMT_FP8]) == min20
        for i in links:
            bf = P8.bufs[i]
            wr = [o for o in P8.ops if o["kind"] == 1 and o["out_buf"] == i]
            rd = [o for o in P8.ops if o["kind"] == 1 and o["in_buf"] == i]
            assert bf["pitch"] % 64 == 0 and bf["scale"] > 0 and min(bf["h"], bf["w"]) >= 40 and not any(o["res_buf"] == i for o in P8.ops)
            assert wr and rd and all(o["ksize"] == 3 and o["stride"] == 1 and not o.get("grp_cout") for o in wr + rd) and all(o["cin"] % 64 == 0 and o["in_coff"] % 16 == 0 for o in rd)
            assert len(wr) == len(rd) and all(o["res_buf"] < 0 and o["cout_store"] == o["cout_pad"] for o in wr)  # an e4m3 output takes whole cout tiles and no residual
            if bf["name"].split(".")[-1].startswith("mid"):
                assert len(wr) == 1 and bf["live"] == wr[0]["cout_store"] and rd[0]["cin"] == bf["pitch"]
        sc = {bf["name"]: bf["scale"] for bf in P8.bufs if bf["is_f32"] == arch.FMT_FP8}
        assert abs(sc["backbone.stage2.blocks.mid0"] - 10.0 * arch.FP8_HEADROOM / arch.FP8_MAX) < 1e-9 and abs(sc["backbone.stage2.blocks.mid1"] - 8.0 * arch.FP8_HEADROOM / arch.FP8_MAX) < 1e-9
        if variant == "vgg_heads_l":  # the 96-channel links of stage 1: zero weight columns over the 32 bytes nobody writes
            op = next(o for o in P8.ops if o["name"] == "backbone.stage1.blocks.bottlenecks.0.cv2")
            w = P8.arrays()[0][op["w_off"] : op["w_off"] + op["cout_pad"] * 9 * op["cin"]].reshape(op["cout_pad"], 9, op["cin"])
            assert op["cin"] == 128 and float(np.abs(w[..., 96:]).max()) == 0.0 and float(np.abs(w[..., :96]).max()) > 0
        assert all(bf["is_f32"] in (arch.FMT_F16, arch.FMT_F32) for bf in Ph.bufs) and sum(bf["is_f32"] == arch.FMT_F32 for bf in Ph.bufs) == 3
        ab, a8, ah = (arch.program_algorithmic_bytes(P, 64) for P in (Pb, P8, Ph))
        assert a8["write"] < ab["write"] and a8["read"] < ab["read"] and ah["read"] > 0.98 * ab["read"]  # fp16 keeps the padded 64-channel stem pitch: a little more than bf16
        assert arch.op_touches_fp8(P8, next(o for o in P8.ops if o["name"].endswith("bottlenecks.0.cv1") and "stage2" in o["name"])) and not any(arch.op_touches_fp8(Pb, o) for o in Pb.ops)


def test_int8_program_is_the_fp8_program_with_signed_byte_links():
    """r05 (CPU): precision="int8" (the reference exporter's QuantizationMode.INT8: exportable_mesh_model.py:175-178,398-411) places VGH_FMT_I8 buffers exactly where "fp8"
    places e4m3 ones -- same ops, same pitches, same byte counts -- with scale = calibrated max * 1.25 / 127; the pack header keeps the bf16 precision code."""
    from head_detector_amd import _lib

    assert arch.FMT_I8 == _lib.VGH_FMT_I8 == 6 and arch.Q8_PRECISIONS == {"fp8": arch.FMT_FP8, "int8": arch.FMT_I8}
    for variant in ("vgg_heads_l", "vgg_heads_m"):
        sd = arch.random_state_dict(variant, 2)
        sc = {"backbone.stage2.blocks.mid0": 10.0}
        P8, Pi = arch.build_program(variant, sd, 640, "fp8", fp8_scales=sc), arch.build_program(variant, sd, 640, "int8", fp8_scales=sc)
        assert [o["name"] for o in P8.ops] == [o["name"] for o in Pi.ops] and len(P8.bufs) == len(Pi.bufs)
        for b8, bi in zip(P8.bufs, Pi.bufs):
            assert (b8["is_f32"] == arch.FMT_FP8) == (bi["is_f32"] == arch.FMT_I8) and {k: b8[k] for k in ("name", "h", "w", "pitch")} == {k: bi[k] for k in ("name", "h", "w", "pitch")}
            if bi["is_f32"] == arch.FMT_I8:
                amax = sc.get(bi["name"], 8.0)
                assert abs(bi["scale"] - amax * arch.I8_HEADROOM / 127.0) < 1e-9 and abs(b8["scale"] - amax * arch.FP8_HEADROOM / 448.0) < 1e-9
        assert arch.program_algorithmic_bytes(P8, 64) == arch.program_algorithmic_bytes(Pi, 64)
        assert arch.PRECISION_FMT["int8"] == arch.FMT_BF16 and sum(arch.op_touches_fp8(Pi, o) for o in Pi.ops) == sum(arch.op_touches_fp8(P8, o) for o in P8.ops) > 0


def test_flame_prologue_joint_reduction_tree_is_the_xor_butterfly():
    """CPU spec of the cross-lane reduction in csrc/flame.hip::prep_head (J = J0 + JS beta: 24 sums over 64 lanes): on the first three levels the xor partners split
    the outputs between them (keep half, send half), then three outputs per lane go through a plain butterfly -- 30 cross-lane moves instead of 144.  Every output's
    addition tree must be the butterfly's (level by level own + partner's partial sum), i.e. the same fp32 bits, or a head's vertices would change with the kernel
    revision.  Also specifies where the sums land: outputs jbase .. jbase + 2 in the lanes with lane % 8 == 0, jbase = 12 b5 + 6 b4 + 3 b3."""
    import numpy as np

    f32 = np.float32
    lanes = np.arange(64)
    for seed in range(4):
        s = (np.random.default_rng(seed).standard_normal((24, 64)) * 10.0 ** (seed - 2)).astype(f32)
        ref = np.zeros(24, f32)
        for o in range(24):
            v = s[o].copy()
            for off in (32, 16, 8, 4, 2, 1):
                v = (v + v[lanes ^ off]).astype(f32)
            ref[o] = v[o]
        t, n = s, 24
        for bit in (32, 16, 8):  # keep half / send half
            n //= 2
            up = (lanes & bit) != 0
            t = np.stack([(np.where(up, t[q + n], t[q]) + np.where(up, t[q], t[q + n])[lanes ^ bit]).astype(f32) for q in range(n)])
        for off in (4, 2, 1):
            t = np.stack([(t[q] + t[q][lanes ^ off]).astype(f32) for q in range(3)])
        out = np.zeros(24, f32)
        for lane in range(0, 64, 8):
            jbase = (12 if lane & 32 else 0) + (6 if lane & 16 else 0) + (3 if lane & 8 else 0)
            out[jbase : jbase + 3] = t[:, lane]
        assert np.array_equal(out, ref)


def test_flame_c3_live_groups_tile_rows_and_interleaved_basis_layout():
    """CPU spec of the index arithmetic of csrc/flame.hip::flame_c3_kernel: (1) the live k-groups (8 consecutive k) of the shape / expression / pose ranges in ascending
    order and gi -> g; (2) walking them group by group, pair by pair, with the kernel's `live` test visits exactly the even k of the three ranges in ascending order --
    the fmaf chain every FLAME vertex kernel must reproduce -- and the rows it does NOT visit inside a live group are the ones `zero_dead_rows` clears; (3) the
    coefficient tile row of a k (gi * 8 + k % 8) agrees with the fused variant's staging map; (4) the k-interleaved basis copy: the 16 bytes a lane (j, half) loads for
    group g hold k = 8g + half + {0, 2, 4, 6} of its vertex, and the layout is a bijection."""
    import numpy as np

    NB, K, Kp, Vp = 400, 436, 440, 64

    def groups(r0e, r1b, r1e, r2b, r2e):
        g0e = (r0e + 7) >> 3
        g1b, g1e = (max(r1b >> 3, g0e), max((r1e + 7) >> 3, max(r1b >> 3, g0e))) if r1e > r1b else (g0e, g0e)
        g2b = max(r2b >> 3, g1e)
        g2e = max((r2e + 7) >> 3, g2b)
        c0, c01 = g0e, g0e + (g1e - g1b)
        ng = c01 + (g2e - g2b)
        gof = lambda gi: gi if gi < c0 else g1b + (gi - c0) if gi < c01 else g2b + (gi - c01)  # noqa: E731
        return g0e, g1b, c0, c01, ng, gof

    for shape_live, expr_live in ((64, 32), (128, 64), (300, 100), (300, 0), (0, 100), (0, 0), (296, 4), (4, 36)):
        r0e, r1b, r1e, r2b, r2e = shape_live, 300, 300 + expr_live, NB, K
        live = lambda k: k < r0e or r1b <= k < r1e or r2b <= k < r2e  # noqa: E731
        g0e, g1b, c0, c01, ng, gof = groups(r0e, r1b, r1e, r2b, r2e)
        gs = [gof(gi) for gi in range(ng)]
        assert gs == sorted(set(gs)) and all(0 <= g < Kp // 8 for g in gs)  # ascending, no group twice, inside the padded basis
        want = [k for k in range(0, K, 2) if live(k)]
        visited = [gof(gi) * 8 + 2 * i for gi in range(ng) for i in range(4) if live(gof(gi) * 8 + 2 * i)]
        assert visited == want  # the chain order
        assert all(any(live(g * 8 + r) for r in range(8)) for g in gs)  # no dead group is staged
        assert all(live(k) == live(k + 1) for k in range(0, Kp, 2))  # a pair is in or out as a whole (even bounds)
        for k in want:  # tile row of k: the fused staging map == the enumeration
            g = k >> 3
            gi_stage = g if g < g0e else c0 + (g - g1b)
            assert k >= r2b or gof(gi_stage) == g
        assert c01 * 8 + (K - NB) <= ng * 8 and gof(c01) * 8 == NB  # the pose rows start a group right behind the betas' groups
    # (4) basis8[((g * 3 + c) * 2 + (k & 1)) * Vp + v][(k >> 1) & 3]
    idx = np.full(Kp * 3 * Vp, -1, np.int64)
    for k in range(Kp):
        for c in range(3):
            v = np.arange(Vp)
            flat = ((((k >> 3) * 3 + c) * 2 + (k & 1)) * Vp + v) * 4 + ((k >> 1) & 3)
            assert (idx[flat] == -1).all()
            idx[flat] = (k * 3 + c) * Vp + v
    assert (idx >= 0).all()  # a bijection onto the padded basis
    for g, c, half, j in ((0, 0, 0, 0), (5, 2, 1, 17), (54, 1, 0, 63)):
        base = (((g * 3 + c) * 2 + half) * Vp + j) * 4
        assert [int(idx[base + i]) for i in range(4)] == [((8 * g + half + 2 * i) * 3 + c) * Vp + j for i in range(4)]
