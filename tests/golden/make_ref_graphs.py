"""Generates tests/golden/ref_graphs.json from what the REFERENCE ships under /root/reference/models (run in the build
container, where the reference tree exists; the GPU box only sees the committed JSON):

* every ``*.ckpt.index`` (TF V2 tensor bundle index): inference variable names and shapes -- this is all the reference
  holds for the L12 / L8 models (the models the headline metric is quoted on; their .data files are not shipped);
* every ``*.ckpt.meta`` (MetaGraphDef written by tf.train.Saver next to the L7 / L2 checkpoints): the inference
  subgraph of ``y_`` -- which conv reads which tensor and which variable, bias adds, the PReLU subgraph, the input
  ORDER of every ConcatV2, DepthToSpace block sizes, the final add with x2.

Both are parsed without TensorFlow: the index with dcscn-super-resolution_amd/ckpt.py, the meta graph with the
protobuf wire-format walker below (GraphDef = repeated NodeDef{name=1, op=2, input=3, attr=5}).  tests/test_ref_graphs.py
checks oracle/dcscn_oracle.py (variable_shapes, build_topology) and the library's tensor list against this file.

    python tests/golden/make_ref_graphs.py [/root/reference/models]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dcscn_amd import ckpt  # noqa: E402


def _varint(b, p):
    r = s = 0
    while True:
        x = b[p]
        p += 1
        r |= (x & 0x7F) << s
        if not x & 0x80:
            return r, p
        s += 7


def _fields(b):
    p, n = 0, len(b)
    while p < n:
        key, p = _varint(b, p)
        f, w = key >> 3, key & 7
        if w == 0:
            v, p = _varint(b, p)
        elif w == 1:
            v = b[p:p + 8]
            p += 8
        elif w == 2:
            ln, p = _varint(b, p)
            v = b[p:p + ln]
            p += ln
        elif w == 5:
            v = b[p:p + 4]
            p += 4
        else:
            raise ValueError("wire type %d" % w)
        yield f, w, v


def _attr_value(buf):
    """AttrValue: s=2 bytes, i=3 int64, b=5 bool, list=1 {i=3 repeated (packed or not)}."""
    out = {}
    for f, w, v in _fields(buf):
        if f == 2:
            out["s"] = v.decode("latin1")
        elif f == 3:
            out["i"] = v
        elif f == 5:
            out["b"] = bool(v)
        elif f == 8:                      # TensorProto: float_val = 5 (fixed32, possibly packed), tensor_content = 4
            import struct
            vals = []
            for f2, w2, v2 in _fields(v):
                if f2 == 5 and w2 == 5:
                    vals.append(struct.unpack("<f", v2)[0])
                elif f2 == 5 and w2 == 2:
                    vals.extend(struct.unpack("<%df" % (len(v2) // 4), v2))
                elif f2 == 4 and len(v2) % 4 == 0 and len(v2) <= 64:
                    vals.extend(struct.unpack("<%df" % (len(v2) // 4), v2))
            out["tensor_f"] = vals
        elif f == 1:
            ints = []
            for f2, w2, v2 in _fields(v):
                if f2 == 3 and w2 == 0:
                    ints.append(v2)
                elif f2 == 3 and w2 == 2:
                    p = 0
                    while p < len(v2):
                        x, p = _varint(v2, p)
                        ints.append(x)
            out["list_i"] = ints
    return out


def read_graph(meta_path):
    with open(meta_path, "rb") as f:
        buf = f.read()
    graph = None
    for f_, w, v in _fields(buf):
        if f_ == 2:
            graph = v                         # MetaGraphDef.graph_def
    nodes = {}
    order = []
    for f_, w, v in _fields(graph):
        if f_ != 1:
            continue
        nd = {"input": [], "attr": {}}
        for f2, w2, v2 in _fields(v):
            if f2 == 1:
                nd["name"] = v2.decode()
            elif f2 == 2:
                nd["op"] = v2.decode()
            elif f2 == 3:
                nd["input"].append(v2.decode())
            elif f2 == 5:
                k = val = None
                for f3, w3, v3 in _fields(v2):
                    if f3 == 1:
                        k = v3.decode()
                    elif f3 == 2:
                        val = v3
                nd["attr"][k] = _attr_value(val) if val is not None else {}
        nodes[nd["name"]] = nd
        order.append(nd["name"])
    return nodes, order


CONV_OPS = ("Conv2D", "DepthwiseConv2dNative", "Conv2DBackpropInput")


def inference_graph(meta_path):
    nodes, order = read_graph(meta_path)

    def clean(name):
        name = name.lstrip("^")
        return name.split(":")[0]

    # the output: the Add that consumes placeholder x2 (DCSCN.py:325)
    outs = [n for n in order if nodes[n]["op"] in ("Add", "AddV2") and "x2" in [clean(i) for i in nodes[n]["input"]]]
    assert len(outs) == 1, outs
    y = outs[0]
    # backward closure of y_
    seen, stack = set(), [y]
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        stack.extend(clean(i) for i in nodes[n]["input"] if not i.startswith("^"))
    sub = [n for n in order if n in seen]

    def var_of(name):                          # "<var>/read" -> "<var>"
        nd = nodes[clean(name)]
        if nd["op"] == "Identity" and nodes[clean(nd["input"][0])]["op"] in ("VariableV2", "Variable"):
            return clean(nd["input"][0])
        return None

    conv_nodes = [n for n in sub if nodes[n]["op"] in CONV_OPS]
    scopes = sorted({n.rsplit("/", 1)[0] for n in conv_nodes}, key=len, reverse=True)
    # separable_conv2d puts its two convs one level deeper ("<scope>/<name>_conv/depthwise"): use the layer scope
    scopes = sorted({s.rsplit("/", 1)[0] if s.endswith("_conv") else s for s in scopes}, key=len, reverse=True)

    def producer(name):
        """Layer scope / placeholder / concat / depth_to_space node that produced tensor `name`."""
        name = clean(name)
        nd = nodes[name]
        if nd["op"] == "Placeholder" or nd["op"] in ("ConcatV2", "DepthToSpace"):
            return name
        for s in scopes:
            if name.startswith(s + "/"):
                return s
        raise ValueError("unattributed tensor " + name)

    layers = []
    done = set()
    for n in sub:
        nd = nodes[n]
        if nd["op"] not in CONV_OPS:
            continue
        scope = n.rsplit("/", 1)[0]
        if scope.endswith("_conv"):
            scope = scope.rsplit("/", 1)[0]
        if scope in done:
            continue
        done.add(scope)
        members = [m for m in sub if m.startswith(scope + "/")]
        convs = [m for m in members if nodes[m]["op"] in CONV_OPS]
        entry = {"scope": scope, "ops": [nodes[m]["op"] for m in convs]}
        first = nodes[convs[0]]
        data_in = first["input"][2] if first["op"] == "Conv2DBackpropInput" else first["input"][0]
        entry["src"] = producer(data_in)
        entry["filters"] = []
        for m in convs:
            filt = nodes[m]["input"][1]
            entry["filters"].append(var_of(filt))
        entry["strides"] = first["attr"].get("strides", {}).get("list_i")
        entry["padding"] = first["attr"].get("padding", {}).get("s")
        entry["data_format"] = first["attr"].get("data_format", {}).get("s")
        # bias: an Add whose second input is a variable read
        entry["bias"] = None
        for m in members:
            if nodes[m]["op"] in ("Add", "AddV2", "BiasAdd") and len(nodes[m]["input"]) == 2:
                v = var_of(nodes[m]["input"][1])
                if v and v.endswith("conv_B"):
                    entry["bias"] = v
        # activator subgraph (build_activator, tf_graph.py:77-102)
        ops_in = sorted({nodes[m]["op"] for m in members if "/prelu/" in m or m.rsplit("/", 1)[-1].startswith(("relu", "Relu", "leaky", "sigmoid", "tanh", "selu"))})
        prelu = [m for m in members if m.startswith(scope + "/prelu/")]
        if prelu:
            alpha = [var_of(nodes[m]["input"][0]) for m in prelu if nodes[m]["op"] == "Mul" and var_of(nodes[m]["input"][0])]
            # the second multiply, by the constant 0.5 (named mul_1 or mul depending on the TF version that wrote the graph)
            half = None
            for m in prelu:
                if nodes[m]["op"] == "Mul" and m in seen and not var_of(nodes[m]["input"][0]):
                    c = nodes[clean(nodes[m]["input"][1])]
                    if c["op"] == "Const":
                        vals = c["attr"].get("value", {}).get("tensor_f")
                        half = vals[0] if vals else "Const"
            entry["activator"] = {"kind": "prelu", "alpha": alpha[0] if alpha else None,
                                  "ops": sorted(nodes[m]["op"] for m in prelu if m in seen and nodes[m]["op"] in ("Relu", "Abs", "Sub", "Mul", "Add", "AddV2")),
                                  "half_const": half}
        else:
            entry["activator"] = None if not ops_in else {"kind": "other", "ops": ops_in}
        layers.append(entry)

    concats = []
    for n in sub:
        if nodes[n]["op"] == "ConcatV2":
            concats.append({"name": n, "srcs": [producer(i) for i in nodes[n]["input"][:-1]], "N": nodes[n]["attr"].get("N", {}).get("i")})
    d2s = []
    for n in sub:
        if nodes[n]["op"] == "DepthToSpace":
            d2s.append({"name": n, "src": producer(nodes[n]["input"][0]), "block_size": nodes[n]["attr"]["block_size"]["i"]})
    out = {"name": y, "srcs": [producer(i) for i in nodes[y]["input"]]}
    return {"layers": layers, "concats": concats, "depth_to_space": d2s, "output": out}


def main():
    models = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/models"
    doc = {"source": "jiny2001/dcscn-super-resolution models/ (parsed by tests/golden/make_ref_graphs.py, no TensorFlow)", "models": {}}
    for fn in sorted(os.listdir(models)):
        if not fn.endswith(".ckpt.index"):
            continue
        name = fn[:-len(".ckpt.index")]
        prefix = os.path.join(models, name + ".ckpt")
        variables = {k: list(v) for k, v in ckpt.list_variables(prefix) if not ckpt.is_optimizer_slot(k)}
        entry = {"variables": variables, "has_data": ckpt.has_data(prefix)}
        meta = prefix + ".meta"
        if os.path.isfile(meta):
            entry["graph"] = inference_graph(meta)
        doc["models"][name] = entry
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_graphs.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", out, "with", len(doc["models"]), "models")


if __name__ == "__main__":
    main()
