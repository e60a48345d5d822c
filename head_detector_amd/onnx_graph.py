"""Graph-level ONNX ingest: bind the Conv nodes of a SIMPLIFIED export to the architecture by graph position.

The reference's exporter fuses the RepVGG blocks (``prep_model_for_conversion``), lets ``torch.onnx.export`` merge every eval-mode BatchNorm into its conv, and
then runs ``onnxsim.simplify`` (yolo_head_training/yolo_head/exportable_mesh_model.py:392-393, 440-453, 483-488; README.md:23,199 announces the files).  What is
left of the parameter names is ``onnx::Conv_1234``: a name-keyed reader (onnx_wire.load_initializers + weight_manifest_diff) cannot load such a file.  This module
reads it by STRUCTURE instead:

  * ``arch.module_graph(variant)`` is the network as a module-level dataflow graph in forward order;
  * every Conv / ConvTranspose node of the file gets its set of nearest upstream Conv nodes (through Relu / Add / Mul / Concat / MaxPool / Cast / ...);
  * the modules are walked in forward order and each takes the FIRST unbound node (file order = a topological order) whose operator, weight shape, strides and
    group match its Spec, whose upstream set is exactly the nodes its producers were bound to, and whose nearest downstream convs include the ones the module feeds
    (that separates CSP conv1 from conv2 and the cls tower from the reg tower whatever their order in the file).  Only the rotation / jaw / translation / scale
    branches of a head -- same shapes from the stem to the last 1x1, jaw and translation even in their 3 outputs -- are told apart by file order alone, i.e. by the
    exporter's trace order = the forward order of yolo_head_dfl_head.py:155-160 -- stated here because nothing in the file can confirm it;
  * a bottleneck's residual scale is read off the graph: the Add behind cv2 takes ``Mul(x, alpha)`` (a one-element tensor) or ``x`` itself (alpha = 1);
  * the first module that finds no node stops the ingest with what it expected and which candidates it rejected -- not with a list of names.

The result is a state dict in the FUSED naming ``arch.fold_state_dict`` already accepts (``<block>.rbr_reparam.{weight,bias}``, ``<block>.conv.{weight,bias}`` without
BatchNorm tensors), so everything downstream (weight_manifest_diff, build_program, pack) is the path every other archive takes.  No ``onnx`` package involved.
Unpinned: no released ``.onnx`` exists in this image; the tests write their files with ``tests/onnx_export_standin.py`` (the module graph with fused weights under
anonymous names), and tools/first_contact.py is the one command to run on a real one."""
from __future__ import annotations

from typing import Dict, FrozenSet, List, Optional, Tuple

import numpy as np

from . import arch

CONV_OPS = ("Conv", "ConvTranspose")
_PASS_THROUGH = ("Relu", "Cast", "Identity", "Dropout")


class OnnxGraphError(ValueError):
    pass


def _ints(v, default):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) and len(v) else list(default)


def bind_graph(variant: str, graph: dict) -> Tuple[Dict[str, np.ndarray], dict]:
    """(state dict in the fused naming, report) for ``graph`` = onnx_wire.load_graph(path).  Raises OnnxGraphError at the first module without a node."""
    tensors, nodes = graph["tensors"], graph["nodes"]
    producer: Dict[str, int] = {}
    consumers: Dict[str, List[int]] = {}
    for i, nd in enumerate(nodes):
        for o in nd["outputs"]:
            producer[o] = i
        for t in nd["inputs"]:
            consumers.setdefault(t, []).append(i)
    memo: Dict[str, FrozenSet] = {}

    def upstream(t: str) -> FrozenSet:
        """Nearest Conv / ConvTranspose nodes (or "image") the value ``t`` is computed from."""
        if t in memo:
            return memo[t]
        stack, seen, out = [t], set(), set()
        while stack:
            x = stack.pop()
            if x in seen or x == "":
                continue
            seen.add(x)
            if x in memo:
                out |= memo[x]
            elif x in tensors:
                continue
            elif x not in producer:
                out.add("image")
            elif nodes[producer[x]]["op"] in CONV_OPS:
                out.add(producer[x])
            else:
                stack.extend(nodes[producer[x]]["inputs"])
        memo[t] = frozenset(out)
        return memo[t]

    mg = arch.module_graph(variant)
    kind_of = {m["name"]: m["op"] for m in mg}
    ins_of = {m["name"]: m["inputs"] for m in mg}

    def anchors(t: str) -> FrozenSet[str]:  # the same walk on the module graph
        if t == "image" or kind_of.get(t) in ("conv", "convT"):
            return frozenset([t])
        out = set()
        for x in ins_of[t]:
            out |= anchors(x)
        return frozenset(out)

    # what consumes a conv: the weight shapes of its nearest DOWNSTREAM convs, in the file and in the architecture.  Siblings with one producer and one signature differ
    # there (CSP conv1 feeds the bottlenecks' 3x3 convs, conv2 only conv3; the cls tower ends in a 1-channel prediction, the reg tower in 68), so their order in the
    # file does not matter; what the file adds behind the network (a DFL projection conv behind reg_pred) may come on top: expected must be CONTAINED in found
    conv_ids = [i for i, nd in enumerate(nodes) if nd["op"] in CONV_OPS and len(nd["inputs"]) > 1 and nd["inputs"][1] in tensors]
    down_found: Dict[int, List[tuple]] = {i: [] for i in conv_ids}
    for j in conv_ids:
        for a in upstream(nodes[j]["inputs"][0]):
            if a != "image" and a in down_found:
                down_found[a].append(tuple(tensors[nodes[j]["inputs"][1]].shape))

    def wshape(m):
        sp_ = m["spec"]
        return (sp_.cin, sp_.cout, 2, 2) if m["op"] == "convT" else (sp_.cout, sp_.cin, sp_.k, sp_.k)

    down_want: Dict[str, List[tuple]] = {m["name"]: [] for m in mg if m["op"] in ("conv", "convT")}
    for m in mg:
        if m["op"] in ("conv", "convT"):
            for a in anchors(m["inputs"][0]):
                if a != "image":
                    down_want[a].append(wshape(m))

    def contains(found: List[tuple], want: List[tuple]) -> bool:
        left = list(found)
        for w_ in want:
            if w_ not in left:
                return False
            left.remove(w_)
        return True

    bound: Dict[str, int] = {}
    used = set()
    sd: Dict[str, np.ndarray] = {}

    def describe(i: int) -> str:
        nd = nodes[i]
        w = tensors.get(nd["inputs"][1]) if len(nd["inputs"]) > 1 else None
        return f"node {i} {nd['op']} {nd['name'] or nd['outputs'][0]!r} W{tuple(w.shape) if w is not None else '?'} strides={_ints(nd['attrs'].get('strides'), [1, 1])} fed by {sorted(map(str, upstream(nd['inputs'][0])))}"

    for m in mg:
        if m["op"] not in ("conv", "convT"):
            continue
        sp = m["spec"]
        want_op = "ConvTranspose" if m["op"] == "convT" else "Conv"
        want_w = (sp.cin, sp.cout, 2, 2) if m["op"] == "convT" else (sp.cout, sp.cin, sp.k, sp.k)
        want_up = frozenset("image" if a == "image" else bound[a] for a in anchors(m["inputs"][0]))
        pick, rejected = None, []
        for i, nd in enumerate(nodes):
            if i in used or nd["op"] != want_op or len(nd["inputs"]) < 2 or nd["inputs"][1] not in tensors:
                continue
            w = tensors[nd["inputs"][1]]
            ok_shape = tuple(w.shape) == want_w
            ok_attr = _ints(nd["attrs"].get("strides"), [1, 1]) == [sp.stride, sp.stride] and int(nd["attrs"].get("group", 1) or 1) == 1
            ok_up = upstream(nd["inputs"][0]) == want_up and contains(down_found[i], down_want[sp.name])
            if ok_shape and ok_attr and ok_up:
                pick = i
                break
            if ok_up or (ok_shape and ok_attr and len(rejected) < 4):
                rejected.append(i)
        if pick is None:
            fed = sorted("image" if a == "image" else f"{a} (node {bound[a]})" for a in anchors(m["inputs"][0]))
            raise OnnxGraphError(f"{variant}: no node for module {sp.name!r}: expected {want_op} with W{want_w}, strides [{sp.stride}, {sp.stride}], group 1, fed by {fed}; "
                                 f"{len(bound)} modules bound before it; nearest rejected: {[describe(i) for i in rejected[:4]] or 'none'}")
        nd = nodes[pick]
        used.add(pick)
        bound[sp.name] = pick
        w = np.ascontiguousarray(tensors[nd["inputs"][1]], dtype=np.float32)
        if len(nd["inputs"]) < 3 or nd["inputs"][2] not in tensors:
            raise OnnxGraphError(f"{variant}: {describe(pick)} (module {sp.name!r}) has no constant bias: its BatchNorm was not merged -- an unsimplified export is read by name, "
                                 "not by graph position")
        b = np.ascontiguousarray(tensors[nd["inputs"][2]], dtype=np.float32).reshape(-1)
        if m["relu"]:  # Conv -> (Cast) -> Relu
            outs = [c for c in consumers.get(nd["outputs"][0], [])]
            hops = 0
            while outs and all(nodes[c]["op"] in ("Cast", "Identity") for c in outs) and hops < 3:
                outs = [c2 for c in outs for c2 in consumers.get(nodes[c]["outputs"][0], [])]
                hops += 1
            if not any(nodes[c]["op"] == "Relu" for c in outs):
                raise OnnxGraphError(f"{variant}: {describe(pick)} was bound to {sp.name!r}, which ends in a ReLU, but no Relu consumes its output "
                                     f"(consumers: {[nodes[c]['op'] for c in outs]})")
        key = {"qarep": f"{sp.name}.rbr_reparam", "conv": f"{sp.name}.conv", "cbr": f"{sp.name}.seq.conv", "plain": sp.name, "convT": sp.name}[sp.kind]
        sd[f"{key}.weight"], sd[f"{key}.bias"] = w, b

    # bottleneck alphas: Add(<x or Mul(x, alpha)>, relu(cv2))
    for m in mg:
        if m["op"] != "add":
            continue
        cv2 = nodes[bound[m["inputs"][1]]]
        t = cv2["outputs"][0]
        add = None
        for _ in range(4):  # through Relu / Cast to the Add
            nxt = [c for c in consumers.get(t, [])]
            a = [c for c in nxt if nodes[c]["op"] == "Add"]
            if a:
                add = nodes[a[0]]
                break
            p = [c for c in nxt if nodes[c]["op"] in _PASS_THROUGH]
            if not p:
                break
            t = nodes[p[0]]["outputs"][0]
        if add is None:
            raise OnnxGraphError(f"{variant}: no Add behind the node bound to {m['inputs'][1]!r} (the bottleneck's residual sum)")
        other = [x for x in add["inputs"] if x != t]
        alpha = 1.0
        if len(other) == 1 and other[0] in producer and nodes[producer[other[0]]]["op"] == "Mul":
            mul = nodes[producer[other[0]]]
            sc = [tensors[x] for x in mul["inputs"] if x in tensors and np.asarray(tensors[x]).size == 1]
            if len(sc) != 1:
                raise OnnxGraphError(f"{variant}: the residual of {m['name']!r} passes a Mul without a one-element constant (inputs {mul['inputs']})")
            alpha = float(np.asarray(sc[0], dtype=np.float64).reshape(-1)[0])
        sd[m["alpha"]] = np.array([alpha], dtype=np.float32)
    report = {"conv_nodes_bound": len(bound), "conv_nodes_unbound": [describe(i) for i, nd in enumerate(nodes) if nd["op"] in CONV_OPS and i not in used][:8],
              "alphas": sum(1 for m in mg if m["op"] == "add")}
    return sd, report


def load_by_graph(path: str, variant: Optional[str] = None) -> Tuple[str, Dict[str, np.ndarray], dict]:
    """(variant, fused-name state dict, report) of a simplified ONNX export; ``variant=None`` tries every known architecture (their channel counts differ, so at most
    one binds) and reports each failure if none does."""
    from . import onnx_wire

    graph = onnx_wire.load_graph(path)
    if not any(nd["op"] in CONV_OPS for nd in graph["nodes"]):
        raise OnnxGraphError(f"{path}: the graph holds no Conv node (initializers only): nothing to bind by position")
    errs = []
    for v in ([variant] if variant else sorted(arch.VARIANTS)):
        try:
            sd, rep = bind_graph(v, graph)
            return v, sd, rep
        except OnnxGraphError as e:
            errs.append(str(e))
    raise OnnxGraphError(f"{path}: the graph does not bind to any known architecture:\n  " + "\n  ".join(errs))
