"""ONNX initializers without the ``onnx`` package (ingest half).

The reference advertises ONNX weights next to the TorchScript archive (/root/reference/README.md:23,199; the exporter is
yolo_head_training/yolo_head/exportable_mesh_model.py:398-411).  All the engine needs from such a file is the tensors of
``ModelProto.graph.initializer`` -- (name, dims, data) -- which then go through the same ``weight_manifest_diff`` /
``arch.fold_state_dict`` path as a .trcd state_dict.  This module reads exactly that much of the protobuf wire format
(varint / 64-bit / length-delimited / 32-bit records; https://protobuf.dev/programming-guides/encoding/), and writes it (the
tests build their files with ``write_model``: there is no ONNX file of the released model in this image, so ingest of a REAL
export is unpinned -- see ``load_initializers`` on what an exporter may have folded away).

Field numbers (onnx/onnx.proto3):  ModelProto.graph = 7;  GraphProto.node = 1, .name = 2, .initializer = 5, .input = 11, .output = 12;
TensorProto.dims = 1, .data_type = 2, .float_data = 4, .int32_data = 5, .int64_data = 7, .name = 8, .raw_data = 9, .double_data = 10,
.external_data = 13, .data_location = 14;  NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5;
AttributeProto.name = 1, .f = 2, .i = 3, .s = 4, .t = 5, .floats = 7, .ints = 8;  ValueInfoProto.name = 1.

r06: ``load_graph`` reads the NODES too (op type, inputs, outputs, the integer / float / tensor attributes), which is what a file that went through
``onnxsim.simplify`` needs (exportable_mesh_model.py:483-488): its Conv weights are ``onnx::Conv_NNN`` initializers with the BatchNorm already merged, so they can
only be bound to the op program by graph position -- head_detector_amd/onnx_graph.py does that.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np

# TensorProto.DataType -> numpy dtype of raw_data (little-endian, as the spec fixes it)
_RAW_DTYPE = {1: "<f4", 2: "u1", 3: "i1", 5: "<i2", 6: "<i4", 7: "<i8", 9: "?", 10: "<f2", 11: "<f8", 12: "<u4", 13: "<u8", 16: "bf16"}
FLOAT, FLOAT16, BFLOAT16, DOUBLE, INT64 = 1, 10, 16, 11, 7


class OnnxWireError(ValueError):
    pass


def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    out, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise OnnxWireError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise OnnxWireError("varint longer than 10 bytes")


def _fields(buf: memoryview) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message: value = int (varint, fixed), or a memoryview (length-delimited)."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > len(buf):
                raise OnnxWireError(f"field {fno}: truncated 64-bit record")
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 5:
            if pos + 4 > len(buf):
                raise OnnxWireError(f"field {fno}: truncated 32-bit record")
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > len(buf):
                raise OnnxWireError(f"field {fno}: length {n} runs past the end of its message")
            v, pos = buf[pos : pos + n], pos + n
        else:
            raise OnnxWireError(f"field {fno}: wire type {wt} (groups) is not part of ONNX")
        yield fno, wt, v


def _packed_varints(v, wt) -> List[int]:
    if wt == 0:
        return [v]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _signed64(x: int) -> int:
    return x - (1 << 64) if x >= (1 << 63) else x


def _bf16_to_f32(raw: bytes) -> np.ndarray:
    return (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)


def _tensor(buf: memoryview) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype, name, raw, external = 0, "", None, False
    f32: List[np.ndarray] = []
    f64: List[np.ndarray] = []
    i32: List[int] = []
    i64: List[int] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += [_signed64(x) for x in _packed_varints(v, wt)]
        elif fno == 2:
            dtype = v
        elif fno == 4:  # float_data: packed (wire type 2) or one 32-bit record per element
            f32.append(np.frombuffer(bytes(v), dtype="<f4") if wt == 2 else np.array([struct.unpack("<f", struct.pack("<I", v))[0]], dtype=np.float32))
        elif fno == 5:
            i32 += _packed_varints(v, wt)
        elif fno == 7:
            i64 += [_signed64(x) for x in _packed_varints(v, wt)]
        elif fno == 8:
            name = bytes(v).decode("utf-8")
        elif fno == 9:
            raw = bytes(v)
        elif fno == 10:
            f64.append(np.frombuffer(bytes(v), dtype="<f8") if wt == 2 else np.array([struct.unpack("<d", struct.pack("<Q", v))[0]]))
        elif fno == 13 or (fno == 14 and v == 1):
            external = True
    if external:
        raise OnnxWireError(f"initializer {name!r} keeps its data in an external file (data_location = EXTERNAL): re-export with the weights embedded")
    n = int(np.prod(dims)) if dims else 1
    if raw is not None:
        if dtype not in _RAW_DTYPE:
            raise OnnxWireError(f"initializer {name!r}: data_type {dtype} is not supported")
        a = _bf16_to_f32(raw) if dtype == BFLOAT16 else np.frombuffer(raw, dtype=_RAW_DTYPE[dtype])
    elif f32:
        a = np.concatenate(f32)
    elif f64:
        a = np.concatenate(f64)
    elif i64:
        a = np.array(i64, dtype=np.int64)
    elif i32:  # also the carrier of FLOAT16 / BFLOAT16 bit patterns and of the small integer types
        a = np.array(i32, dtype=np.int64)
        if dtype == FLOAT16:
            a = a.astype(np.uint16).view(np.float16)
        elif dtype == BFLOAT16:
            a = _bf16_to_f32(a.astype("<u2").tobytes())
        else:
            a = a.astype(np.int32)
    else:
        a = np.zeros(0, dtype=np.float32)
    if a.size != n:
        raise OnnxWireError(f"initializer {name!r}: dims {dims} want {n} elements, the data holds {a.size}")
    return name, a.reshape(dims)


def load_initializers(path: str) -> Tuple[Dict[str, np.ndarray], List[Tuple[str, str, List[str]]]]:
    """({initializer name: array}, [(op_type, node name, input names)] in graph order) of an ONNX file.

    What an exporter does to the names is outside this reader: ``torch.onnx.export`` keeps a parameter's qualified name for an initializer
    it does not touch and renames what it constant-folds (a Conv whose BatchNorm it merged arrives as ``onnx::Conv_123`` with a bias and no
    BN tensors).  ``detector.load_weights`` strips the ``model.`` prefix; ``weight_manifest_diff`` then reports the whole difference, and
    ``arch.fold_state_dict`` accepts a conv that arrives pre-folded under its own name (``<block>.conv.weight`` + ``<block>.conv.bias``, no BN keys)."""
    with open(path, "rb") as f:
        data = memoryview(f.read())
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise OnnxWireError(f"{path}: no ModelProto.graph (field 7): not an ONNX model")
    tensors: Dict[str, np.ndarray] = {}
    nodes: List[Tuple[str, str, List[str]]] = []
    for fno, wt, v in _fields(graph):
        if fno == 5 and wt == 2:
            name, arr = _tensor(v)
            if name in tensors:
                raise OnnxWireError(f"{path}: initializer {name!r} appears twice")
            tensors[name] = arr
        elif fno == 1 and wt == 2:
            ins, op, nm = [], "", ""
            for nf, nw, nv in _fields(v):
                if nf == 1:
                    ins.append(bytes(nv).decode("utf-8"))
                elif nf == 3:
                    nm = bytes(nv).decode("utf-8")
                elif nf == 4:
                    op = bytes(nv).decode("utf-8")
            nodes.append((op, nm, ins))
    return tensors, nodes


def _attribute(buf: memoryview) -> Tuple[str, object]:
    name, val = "", None
    ints: List[int] = []
    floats: List[float] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode("utf-8")
        elif fno == 2:
            val = struct.unpack("<f", struct.pack("<I", v))[0]
        elif fno == 3:
            val = _signed64(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            floats += list(np.frombuffer(bytes(v), dtype="<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif fno == 8:
            ints += [_signed64(x) for x in _packed_varints(v, wt)]
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def load_graph(path: str) -> dict:
    """The whole GraphProto the binder needs: ``{"tensors": {name: array}, "nodes": [{"op", "name", "inputs", "outputs", "attrs"}] in file order (= a topological
    order: the ONNX spec requires it), "inputs": [graph input names that are not initializers], "outputs": [...]}``.  ``Constant`` nodes are folded into
    ``tensors`` under their output name (an exporter is free to keep a scalar as either)."""
    with open(path, "rb") as f:
        data = memoryview(f.read())
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise OnnxWireError(f"{path}: no ModelProto.graph (field 7): not an ONNX model")
    tensors: Dict[str, np.ndarray] = {}
    nodes: List[dict] = []
    gin: List[str] = []
    gout: List[str] = []
    for fno, wt, v in _fields(graph):
        if fno == 5 and wt == 2:
            name, arr = _tensor(v)
            if name in tensors:
                raise OnnxWireError(f"{path}: initializer {name!r} appears twice")
            tensors[name] = arr
        elif fno == 1 and wt == 2:
            nd = {"op": "", "name": "", "inputs": [], "outputs": [], "attrs": {}}
            for nf, nw, nv in _fields(v):
                if nf == 1:
                    nd["inputs"].append(bytes(nv).decode("utf-8"))
                elif nf == 2:
                    nd["outputs"].append(bytes(nv).decode("utf-8"))
                elif nf == 3:
                    nd["name"] = bytes(nv).decode("utf-8")
                elif nf == 4:
                    nd["op"] = bytes(nv).decode("utf-8")
                elif nf == 5:
                    k, a = _attribute(nv)
                    nd["attrs"][k] = a
            if nd["op"] == "Constant" and isinstance(nd["attrs"].get("value"), np.ndarray) and nd["outputs"]:
                tensors[nd["outputs"][0]] = nd["attrs"]["value"]
            else:
                nodes.append(nd)
        elif fno in (11, 12) and wt == 2:
            for vf, vw, vv in _fields(v):
                if vf == 1:
                    (gin if fno == 11 else gout).append(bytes(vv).decode("utf-8"))
    return {"tensors": tensors, "nodes": nodes, "inputs": [n for n in gin if n not in tensors], "outputs": gout}


# ---------------------------------------------------------------------------------------------------------------------
# writer (tests, and `python -m head_detector_amd.onnx_wire out.onnx <variant> <seed>`): the same wire format, initializers only
def _enc_varint(x: int) -> bytes:
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(fno: int, x: int) -> bytes:
    return _enc_varint(fno << 3) + _enc_varint(x)


def encode_tensor(name: str, a: np.ndarray, how: str = "raw") -> bytes:
    """how: 'raw' (raw_data in the array's dtype: f32 / f16 / f64 / i64), 'float_data' (packed float_data), 'bf16' (raw bfloat16, round to nearest even),
    'dims_unpacked' (raw f32 with one varint record per dim, as proto2 writers emit)."""
    shape = np.asarray(a).shape  # (np.ascontiguousarray promotes a 0-d array to 1-d: a scalar initializer has NO dims record)
    a = np.ascontiguousarray(a)
    body = b""
    if how == "dims_unpacked":
        for d in shape:
            body += _vi(1, int(d))
    elif shape:
        body += _ld(1, b"".join(_enc_varint(int(d)) for d in shape))
    if how == "float_data":
        body += _vi(2, FLOAT) + _ld(4, a.astype("<f4").tobytes())
    elif how == "bf16":
        u = a.astype(np.float32).view(np.uint32)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype("<u2")
        body += _vi(2, BFLOAT16) + _ld(9, r.tobytes())
    else:
        code = {np.dtype("float32"): FLOAT, np.dtype("float16"): FLOAT16, np.dtype("float64"): DOUBLE, np.dtype("int64"): INT64}[a.dtype]
        body += _vi(2, code) + _ld(9, a.astype(a.dtype.newbyteorder("<")).tobytes())
    return body + _ld(8, name.encode("utf-8"))


def encode_node(op: str, inputs: List[str], outputs: List[str], name: str = "", attrs: Dict[str, object] = None, packed_ints: bool = True) -> bytes:
    """NodeProto; attrs: int -> i, float -> f, list of int -> ints (packed, or one varint record each as proto2 writers emit), ndarray -> t."""
    body = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    if name:
        body += _ld(3, name.encode())
    body += _ld(4, op.encode())
    for k, v in (attrs or {}).items():
        a = _ld(1, k.encode())
        if isinstance(v, np.ndarray):
            a += _ld(5, encode_tensor("", v)) + _vi(20, 4)
        elif isinstance(v, float):
            a += _enc_varint((2 << 3) | 5) + struct.pack("<f", v) + _vi(20, 1)
        elif isinstance(v, int):
            a += _vi(3, v) + _vi(20, 2)
        else:
            a += (_ld(8, b"".join(_enc_varint(int(x)) for x in v)) if packed_ints else b"".join(_vi(8, int(x)) for x in v)) + _vi(20, 7)
        body += _ld(5, a)
    return body


def write_model(path: str, tensors: Dict[str, np.ndarray], how: Dict[str, str] = None, graph_name: str = "vgg_heads", prefix: str = "", nodes: List[bytes] = None,
                inputs: List[str] = None, outputs: List[str] = None) -> None:
    """``nodes``: encoded NodeProto records (``encode_node``) written ahead of the initializers, ``inputs`` / ``outputs``: graph value names."""
    how = how or {}
    parts = [_ld(1, n) for n in (nodes or [])] + [_ld(2, graph_name.encode())]  # (one join at the end: appending to a growing bytes object copies it every time)
    for k, v in tensors.items():
        parts.append(_ld(5, encode_tensor(prefix + k, np.asarray(v), how.get(k, "raw"))))
    for n in inputs or []:
        parts.append(_ld(11, _ld(1, n.encode())))
    for n in outputs or []:
        parts.append(_ld(12, _ld(1, n.encode())))
    graph = b"".join(parts)
    model = _vi(1, 8) + _ld(2, b"head_detector_amd.onnx_wire") + _ld(7, graph) + _ld(8, _vi(2, 17))  # ir_version 8, producer, graph, opset 17
    with open(path, "wb") as f:
        f.write(model)


if __name__ == "__main__":
    import sys

    from . import arch

    out, variant, seed = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "vgg_heads_m", int(sys.argv[3]) if len(sys.argv) > 3 else 1
    write_model(out, arch.random_state_dict(variant, seed), prefix="model.")
    print(f"wrote {out}: {len(arch.random_state_dict(variant, seed))} initializers of {variant} (seed {seed})")
